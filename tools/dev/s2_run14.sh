cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_prefill_gpu.py tests/test_paged_kv_gpu.py tests/test_kvfp8_gpu.py tests/test_prefill_fullsize_gpu.py tests/test_qkv_rope_fusion_gpu.py -x -q > gpurun_out/s2_t14.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/s2_t14.log
tail -3 gpurun_out/s2_t14.log | cut -c1-600
[ $rc -ne 0 ] && exit 1
for i in 1 2; do for c in prefill:f16:1:2048 prefill:f16:8:512 prefill:f16:4:1024 prefill:f16:2:1024 prefill:f16:16:256; do timeout -k 10 200 python3 bench.py --only $c 2>/dev/null | tail -1 | cut -c1-330; done; done

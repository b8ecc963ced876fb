cd $GRAFT_REPO_ROOT
bash tools/dev/s2_lx.sh | grep -c "ok"
timeout -k 10 1000 python -m pytest tests/test_decoder_gpu.py tests/test_kernels_gpu.py tests/test_ragged_gpu.py tests/test_paged_kv_gpu.py tests/test_kvfp8_gpu.py tests/test_fullsize_gpu.py tests/test_prefill_gpu.py tests/test_prefill_fullsize_gpu.py -x -q > gpurun_out/s2_t7.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/s2_t7.log
tail -4 gpurun_out/s2_t7.log | cut -c1-600
[ $rc -ne 0 ] && exit 1
bash tools/dev/kt_quick.sh decode:f16:1:2048 2>&1 | cut -c1-170
for i in 1 2; do timeout -k 10 200 python3 bench.py --only decode:f16:1:2048 2>/dev/null | tail -1 | cut -c1-120; timeout -k 10 200 python3 bench.py --only prefill:f16:1:2048 2>/dev/null | tail -1 | cut -c1-330; done

cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_prefill_gpu.py tests/test_paged_kv_gpu.py tests/test_kvfp8_gpu.py tests/test_prefill_fullsize_gpu.py tests/test_qkv_rope_fusion_gpu.py -x -q > gpurun_out/s2_t9.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/s2_t9.log
tail -3 gpurun_out/s2_t9.log | cut -c1-600
[ $rc -ne 0 ] && exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I llm-inference-engine_amd/csrc -I include tools/micro/flash_probe.hip llm-inference-engine_amd/csrc/runtime.cpp -o /tmp/flash_probe_x > gpurun_out/s2_probe_build.log 2>&1
timeout -k 10 120 /tmp/flash_probe_x
for i in 1 2; do for c in prefill:f16:1:2048 prefill:f16:8:512; do timeout -k 10 200 python3 bench.py --only $c 2>/dev/null | tail -1 | cut -c1-330; done; done

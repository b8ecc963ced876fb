cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_quant_gpu.py tests/test_prefill_gpu.py tests/test_workspace_gpu.py -x -q > gpurun_out/s2_t28.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/s2_t28.log
tail -3 gpurun_out/s2_t28.log | cut -c1-600
[ $rc -ne 0 ] && exit 1
for T in 256 384 512 768; do timeout -k 10 200 python3 bench.py --only prefill:fp8:1:$T 2>/dev/null | tail -1 | cut -c1-330; done

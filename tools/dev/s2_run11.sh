cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_lane_exchange_gpu.py tests/test_prefill_gpu.py -x -q > gpurun_out/s2_t11.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/s2_t11.log
tail -5 gpurun_out/s2_t11.log | cut -c1-600

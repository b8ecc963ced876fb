cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_kernels_gpu.py tests/test_prefill_gpu.py tests/test_workspace_gpu.py tests/test_cpp_api_gpu.py tests/test_packed_only_gpu.py -x -q > gpurun_out/s2_t19.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/s2_t19.log
tail -3 gpurun_out/s2_t19.log | cut -c1-600
[ $rc -ne 0 ] && exit 1
bash tools/dev/s2_run18.sh

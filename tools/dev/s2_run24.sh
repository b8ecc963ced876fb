cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_decoder_gpu.py tests/test_packed_only_gpu.py tests/test_packed_gpu.py tests/test_chain_gpu.py tests/test_path_switches_gpu.py tests/test_ragged_gpu.py tests/test_paged_kv_gpu.py tests/test_quant_gpu.py tests/test_fullsize_gpu.py tests/test_cpp_api_gpu.py -x -q > gpurun_out/s2_t24.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/s2_t24.log
tail -3 gpurun_out/s2_t24.log | cut -c1-600

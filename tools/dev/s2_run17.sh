cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_kernels_gpu.py tests/test_quant_gpu.py tests/test_prefill_gpu.py tests/test_prefill_fullsize_gpu.py tests/test_qkv_rope_fusion_gpu.py tests/test_packed_only_gpu.py tests/test_workspace_gpu.py -x -q > gpurun_out/s2_t17.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/s2_t17.log
tail -3 gpurun_out/s2_t17.log | cut -c1-600

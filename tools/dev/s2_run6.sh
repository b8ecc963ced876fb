cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_qkv_rope_fusion_gpu.py tests/test_packed_only_gpu.py -x -q > gpurun_out/s2_t6.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/s2_t6.log
tail -5 gpurun_out/s2_t6.log | cut -c1-800

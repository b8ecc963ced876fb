cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_qkv_rope_fusion_gpu.py tests/test_path_switches_gpu.py::test_no_other_switches_are_read -x -q > gpurun_out/s2_t1.log 2>&1; echo "rc=$?" >> gpurun_out/s2_t1.log
tail -5 gpurun_out/s2_t1.log

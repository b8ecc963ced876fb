cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python3 tools/gemm8p_soak.py 120 > gpurun_out/s2_soak.log 2>&1; echo "rc=$?" >> gpurun_out/s2_soak.log
grep -v amdgpu.ids gpurun_out/s2_soak.log | tail -40

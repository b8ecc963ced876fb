cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 tools/dev/attn_step_ab.py > gpurun_out/s2_ab13.log 2>&1; echo "rc=$?" >> gpurun_out/s2_ab13.log; cat gpurun_out/s2_ab13.log | grep -v amdgpu.ids

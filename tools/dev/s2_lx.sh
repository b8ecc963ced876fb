cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I llm-inference-engine_amd/csrc -I include tools/micro/lane_xor_check.hip -o /tmp/lane_xor_check > gpurun_out/s2_lx_build.log 2>&1
timeout -k 10 60 /tmp/lane_xor_check > gpurun_out/s2_lx.log 2>&1; echo "rc=$?" >> gpurun_out/s2_lx.log; cat gpurun_out/s2_lx.log

cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -DFLASH_STAMPS -I llm-inference-engine_amd/csrc -I include tools/micro/flash_probe.hip llm-inference-engine_amd/csrc/runtime.cpp -o /tmp/flash_probe > gpurun_out/s2_probe_build.log 2>&1
timeout -k 10 120 /tmp/flash_probe > gpurun_out/s2_probe.log 2>&1; echo "rc=$?" >> gpurun_out/s2_probe.log
cat gpurun_out/s2_probe.log

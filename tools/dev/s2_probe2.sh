cd $GRAFT_REPO_ROOT
for v in "" "-DFLASH_SKIP_STAGE" "-DFLASH_SKIP_SOFTMAX" "-DFLASH_SKIP_STAGE -DFLASH_SKIP_SOFTMAX"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value $v -I llm-inference-engine_amd/csrc -I include tools/micro/flash_probe.hip llm-inference-engine_amd/csrc/runtime.cpp -o /tmp/flash_probe_x > gpurun_out/s2_probe_build.log 2>&1
  echo "== variant [$v]"; timeout -k 10 120 /tmp/flash_probe_x
done > gpurun_out/s2_probe2.log 2>&1
cat gpurun_out/s2_probe2.log

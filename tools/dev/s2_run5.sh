cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_qkv_rope_fusion_gpu.py  -x -q > gpurun_out/s2_t5.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/s2_t5.log
tail -5 gpurun_out/s2_t5.log | cut -c1-800
[ $rc -ne 0 ] && exit 1
for i in 1 2 3; do
for c in prefill:f16:1:128 prefill:int8:1:128; do
  echo "== $c fused"; timeout -k 10 200 python3 bench.py --only $c 2>/dev/null | tail -1
  echo "== $c plain"; LLMIE_NO_QKV_ROPE_FUSION=1 timeout -k 10 200 python3 bench.py --only $c 2>/dev/null | tail -1
done
done > gpurun_out/s2_ab5.log 2>&1
cat gpurun_out/s2_ab5.log

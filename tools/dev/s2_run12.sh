cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_prefill_gpu.py -q -k "peaked or single_40" > gpurun_out/s2_t12.log 2>&1; echo "rc=$?" >> gpurun_out/s2_t12.log
grep -E "passed|failed|Error|rc=" gpurun_out/s2_t12.log | cut -c1-300 | tail -6

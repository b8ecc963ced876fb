cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/dev/qkv_rope_diff.py int8_b2 > gpurun_out/s2_diff.log 2>&1; echo "rc=$?" >> gpurun_out/s2_diff.log
tail -40 gpurun_out/s2_diff.log | cut -c1-400

cd $GRAFT_REPO_ROOT
timeout -k 10 1150 python -m pytest tests -m gpu -x -q > gpurun_out/s2_full.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/s2_full.log
tail -6 gpurun_out/s2_full.log | cut -c1-600

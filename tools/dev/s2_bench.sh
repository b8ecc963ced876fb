cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python3 bench.py > gpurun_out/s2_bench.json 2> gpurun_out/s2_bench.err; echo "rc=$?"
tail -c 1500 gpurun_out/s2_bench.json

set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2pk3
mkdir -p $O
timeout -k 10 300 python3 tools/pksweep.py 32 > $O/sweep32.log 2>&1; cat $O/sweep32.log
timeout -k 10 300 python3 tools/pksweep.py 16 > $O/sweep16.log 2>&1; cat $O/sweep16.log

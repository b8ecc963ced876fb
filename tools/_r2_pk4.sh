set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2pk4
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_packed_gpu.py -q > $O/test.log 2>&1 || { tail -30 $O/test.log; }
tail -3 $O/test.log
timeout -k 10 300 python3 tools/pksweep.py 32 > $O/sweep32.log 2>&1; cat $O/sweep32.log
timeout -k 10 300 python3 tools/pkbench.py int8 M=32 > $O/pkbench.log 2>&1; cat $O/pkbench.log

set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2pk1
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_packed_gpu.py -q > $O/test.log 2>&1 || { tail -40 $O/test.log; }
tail -3 $O/test.log
timeout -k 10 300 python3 tools/pkbench.py int8 M=32 M=16 > $O/pkbench.log 2>&1
cat $O/pkbench.log
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tools/pkbench.py int8 M=32 > $O/kt.log 2>&1
python3 $R/tools/summarize_prof.py stats $O/kt $O/kt_stats.csv "pkbench int8 M=32" && grep -v quantize $O/kt_stats.csv | cut -c1-150

# kernel trace of the 128-token weight-only int8 prefill
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; P=$R/gpurun_out/r3prof; S=$R/gpurun_out/r3sum; mkdir -p $P $S; cd $R
rm -rf $P/kt_pf128i8
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt_pf128i8 -- python3 bench.py --only prefill:int8:1:128 > $P/kt_pf128i8.log 2>&1 || { tail -5 $P/kt_pf128i8.log; exit 1; }
grep only $P/kt_pf128i8.log | cut -c1-400
python3 tools/summarize_prof.py stats $P/kt_pf128i8 $S/r03_prefill_int8_s128_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --only prefill:int8:1:128"
cut -c1-200 $S/r03_prefill_int8_s128_kernel_stats.csv

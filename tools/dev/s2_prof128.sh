# refresh of profiles/r03_prefill_f16_s128_kernel_stats.csv alone (same command as tools/profile_r03.sh)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; P=$R/gpurun_out/r3prof; S=$R/gpurun_out/r3sum; mkdir -p $P $S; cd $R
rm -rf $P/kt_pf128
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt_pf128 -- python3 bench.py --only prefill:f16:1:128 > $P/kt_pf128.log 2>&1 || { tail -5 $P/kt_pf128.log; exit 1; }
tail -1 $P/kt_pf128.log | cut -c1-300
python3 tools/summarize_prof.py stats $P/kt_pf128 $S/r03_prefill_f16_s128_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --only prefill:f16:1:128"

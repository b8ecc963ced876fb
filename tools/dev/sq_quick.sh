export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for f in int8 fp8; do
  timeout -k 10 300 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d gpurun_out/sqx_$f -- python3 bench.py --only prefill:$f:1:2048 > gpurun_out/sqx_$f.log 2>&1
  python3 tools/summarize_prof.py sq gpurun_out/sqx_$f gpurun_out/sqx_$f.csv "x" gemm8p
done

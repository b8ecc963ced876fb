#!/bin/bash
# The rocprofv3 runs behind profiles/r03_* (run on the GPU box: gpurun -- 'bash tools/profile_r03.sh'; raw output lands in
# gpurun_out/r3prof, the summaries in gpurun_out/r3sum, from where they are copied to profiles/).  Counter passes are separate
# runs (kernel trace only beside them), the profiled program stands directly behind `--`.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
P=$R/gpurun_out/r3prof
S=$R/gpurun_out/r3sum
mkdir -p $P $S
cd $R
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run() { name=$1; shift; echo "== $name"; timeout -k 10 420 rocprofv3 "$@" > $P/$name.log 2>&1 || { tail -5 $P/$name.log; return 1; }; tail -1 $P/$name.log; }
trio() {   # kernel trace + FETCH_SIZE pass + WRITE_SIZE pass of one bench configuration
  tag=$1; shift
  run kt_$tag --kernel-trace --stats --output-format csv -d $P/kt_$tag -- python3 bench.py --only "$@" --no-graph --steps 8 --warmup 2
  run pf_$tag --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/pf_$tag -- python3 bench.py --only "$@" --no-graph --steps 2 --warmup 1
  run pw_$tag --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/pw_$tag -- python3 bench.py --only "$@" --no-graph --steps 2 --warmup 1
}
SUM="python3 tools/summarize_prof.py"
# ---- BASELINE configs[2] (headline): fp16, batch 1, ctx 2048
trio f16 decode:f16:1:2048
$SUM stats $P/kt_f16 $S/r03_decode_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --only decode:f16:1:2048 --no-graph --steps 8 --warmup 2"
$SUM pmc $P/pf_f16 $P/pw_f16 $S/r03_decode_pmc_hbm_traffic.csv $S/_tmp.json gemv_ksplit 0
J=$S/r03_rocprof_roofline.json
$SUM roofline $J decode_f16_b1_ctx2048 $P/kt_f16 "gemv_ksplit_kernel<1, 4, 2, 16" 0 profiles/r03_decode_kernel_stats.csv $P/pf_f16 $P/pw_f16
$SUM roofline $J decode_f16_b1_ctx2048_attention $P/kt_f16 "decode_attn_split_kernel" 0 profiles/r03_decode_kernel_stats.csv $P/pf_f16 $P/pw_f16
$SUM roofline $J decode_f16_b1_ctx2048_merge $P/kt_f16 "decode_attn_combine_kernel" 0 profiles/r03_decode_kernel_stats.csv $P/pf_f16 $P/pw_f16
# ---- BASELINE configs[3]: int8 weight-only, batch 32, ctx 128
trio i8 decode:int8:32:128
run sq_i8 --pmc $SQ1 --kernel-trace --output-format csv -d $P/sq_i8 -- python3 bench.py --only decode:int8:32:128 --no-graph --steps 2 --warmup 1
$SUM stats $P/kt_i8 $S/r03_int8_b32_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --only decode:int8:32:128 --no-graph --steps 8 --warmup 2"
$SUM pmc $P/pf_i8 $P/pw_i8 $S/r03_int8_b32_pmc_hbm_traffic.csv $S/_tmp.json pk_mfma 0
$SUM sq $P/sq_i8 $S/r03_int8_b32_sq_pmc.csv "rocprofv3 --pmc $SQ1 -- python3 bench.py --only decode:int8:32:128 --no-graph" pk_mfma decode_attn
$SUM roofline $J decode_int8_b32_ctx128 $P/kt_i8 "pk_mfma_kernel<2, 8, 1, 1>" 0 profiles/r03_int8_b32_kernel_stats.csv $P/pf_i8 $P/pw_i8
# ---- prefill 1 x 2048 tokens: fp16 and weight-only int8 (kernel trace + SQ counters: MFMA busy, waits, LDS conflicts)
for f in f16 int8; do
  run kt_pf_$f --kernel-trace --stats --output-format csv -d $P/kt_pf_$f -- python3 bench.py --only prefill:$f:1:2048
  run sq_pf_$f --pmc $SQ1 --kernel-trace --output-format csv -d $P/sq_pf_$f -- python3 bench.py --only prefill:$f:1:2048
  $SUM stats $P/kt_pf_$f $S/r03_prefill_${f}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --only prefill:$f:1:2048"
  $SUM sq $P/sq_pf_$f $S/r03_prefill_${f}_sq_pmc.csv "rocprofv3 --pmc $SQ1 -- python3 bench.py --only prefill:$f:1:2048" gemm8p prefill_flash
done
$SUM roofline $J prefill_f16_b1_s2048 $P/kt_pf_f16 "gemm8p_kernelILb0ELb0ELb1ELi0E" 0 profiles/r03_prefill_f16_kernel_stats.csv
$SUM roofline $J prefill_int8_b1_s2048 $P/kt_pf_int8 "gemm8p_kernelILb0ELb0ELb1ELi8E" 0 profiles/r03_prefill_int8_kernel_stats.csv
# ---- BASELINE configs[1] shape: 128-token prefill (the 128-row split-K kernel): kernel trace + HBM traffic
run kt_pf128 --kernel-trace --stats --output-format csv -d $P/kt_pf128 -- python3 bench.py --only prefill:f16:1:128
run pf_pf128 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/pf_pf128 -- python3 bench.py --only prefill:f16:1:128
run pw_pf128 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/pw_pf128 -- python3 bench.py --only prefill:f16:1:128
$SUM stats $P/kt_pf128 $S/r03_prefill_f16_s128_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --only prefill:f16:1:128"
$SUM pmc $P/pf_pf128 $P/pw_pf128 $S/r03_prefill_f16_s128_pmc_hbm_traffic.csv $S/_tmp.json mid_splitk 0
rm -f $S/_tmp.json
ls -la $S
echo done

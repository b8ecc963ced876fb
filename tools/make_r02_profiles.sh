#!/bin/bash
# Turns the raw rocprofv3 output of tools/profile_r02.sh (gpurun_out/r2prof) into the tracked summaries under profiles/.
set -e
cd "$(dirname "$0")/.."
P=gpurun_out/r2prof
S=tools/summarize_prof.py
SQ="rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace"
python $S stats $P/kt_i8 profiles/r02_int8_b32_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --only decode:int8:32:128 --no-graph --steps 8 --warmup 2  (BASELINE configs[3]: int8 weight-only, batch 32, ctx 128; eager launches; tools/profile_r02.sh)"
python $S pmc $P/pf_i8 $P/pw_i8 profiles/r02_int8_b32_pmc_hbm_traffic.csv /tmp/r02_i8_gu.json "pk_mfma_kernel<2, 8, 1, 1>" 131072
python $S sq $P/sq_i8 profiles/r02_int8_b32_sq_pmc.csv "$SQ -- python3 bench.py --only decode:int8:32:128 --no-graph --steps 2 --warmup 1  (tools/profile_r02.sh)" pk_mfma decode_attn
python $S stats $P/kt_f16 profiles/r02_decode_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --only decode:f16:1:2048 --no-graph --steps 8 --warmup 2  (BASELINE configs[2], the headline: fp16, batch 1, ctx 2048; eager launches; tools/profile_r02.sh)"
python $S stats $P/kt_pf profiles/r02_prefill_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --only prefill:f16:1:2048  (fp16 prefill of 1 x 2048 tokens, 32 layers; tools/profile_r02.sh)"
python $S sq $P/sq_pf profiles/r02_gemm256_pmc.csv "$SQ -- python3 bench.py --only prefill:f16:1:2048  (tools/profile_r02.sh)" gemm8p gemm256
python $S sq $P/sq_pf profiles/r02_flash_pmc.csv "$SQ -- python3 bench.py --only prefill:f16:1:2048  (tools/profile_r02.sh)" prefill_flash
python $S roofline profiles/r02_rocprof_roofline.json decode_int8_b32_ctx128 $P/kt_i8 "pk_mfma_kernel<2, 8, 1, 1>" 131072 profiles/r02_int8_b32_kernel_stats.csv /tmp/r02_i8_gu.json
python $S roofline profiles/r02_rocprof_roofline.json decode_f16_b1_ctx2048 $P/kt_f16 "gemv_ksplit_kernel<1, 4, 2, 16" 128256 profiles/r02_decode_kernel_stats.csv profiles/r01_pmc_traffic.json
python $S roofline profiles/r02_rocprof_roofline.json prefill_f16_b1_s2048 $P/kt_pf "gemm8p_kernelILb0ELb0ELb1E" 352256 profiles/r02_prefill_kernel_stats.csv

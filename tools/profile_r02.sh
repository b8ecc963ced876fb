#!/bin/bash
# The rocprofv3 runs behind profiles/r02_* (run on the GPU box: gpurun -- 'bash tools/profile_r02.sh'; raw output lands in
# gpurun_out/r2prof, tools/summarize_prof.py turns it into the tracked summaries).  Counter passes are separate runs
# (kernel trace only beside them), the profiled program stands directly behind `--`.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
P=$R/gpurun_out/r2prof
mkdir -p $P
cd $R
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run() { name=$1; shift; echo "== $name"; timeout -k 10 420 rocprofv3 "$@" > $P/$name.log 2>&1 || { tail -5 $P/$name.log; return 1; }; tail -1 $P/$name.log; }
# BASELINE configs[3]: int8 weight-only, batch 32, ctx 128 (eager launches: one kernel record per launch)
run kt_i8   --kernel-trace --stats --output-format csv -d $P/kt_i8   -- python3 bench.py --only decode:int8:32:128 --no-graph --steps 8 --warmup 2
run pf_i8   --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/pf_i8 -- python3 bench.py --only decode:int8:32:128 --no-graph --steps 2 --warmup 1
run pw_i8   --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/pw_i8 -- python3 bench.py --only decode:int8:32:128 --no-graph --steps 2 --warmup 1
run sq_i8   --pmc $SQ1 --kernel-trace --output-format csv -d $P/sq_i8 -- python3 bench.py --only decode:int8:32:128 --no-graph --steps 2 --warmup 1
# BASELINE configs[2] (headline): fp16, batch 1, ctx 2048
run kt_f16  --kernel-trace --stats --output-format csv -d $P/kt_f16  -- python3 bench.py --only decode:f16:1:2048 --no-graph --steps 8 --warmup 2
# prefill, 1 x 2048 tokens, fp16: kernel trace + SQ counters (MFMA busy, waits, LDS conflicts)
run kt_pf   --kernel-trace --stats --output-format csv -d $P/kt_pf   -- python3 bench.py --only prefill:f16:1:2048
run sq_pf   --pmc $SQ1 --kernel-trace --output-format csv -d $P/sq_pf -- python3 bench.py --only prefill:f16:1:2048
echo done

set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2pk2
mkdir -p $O
cd /tmp
rocprofv3 -L > $O/counters.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc1 -- python3 $R/tools/pkbench.py int8 M=32 > $O/pmc1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc2 -- python3 $R/tools/pkbench.py int8 M=32 > $O/pmc2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAIT_IFETCH GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc3 -- python3 $R/tools/pkbench.py int8 M=32 > $O/pmc3.log 2>&1 || true
echo done

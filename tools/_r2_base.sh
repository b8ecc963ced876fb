set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2base
mkdir -p $O
python3 tools/opprofile.py int8 32 128 > $O/op_i8_b32.log 2>&1
python3 tools/opprofile.py f16 1 2048 > $O/op_f16_b1.log 2>&1
python3 tools/opprofile.py int4 32 128 > $O/op_i4_b32.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_i8 -- python3 $R/tools/opprofile.py int8 32 128 8 > $O/kt_i8.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_i8 -- python3 $R/tools/opprofile.py int8 32 128 4 > $O/pmc_f_i8.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_i8 -- python3 $R/tools/opprofile.py int8 32 128 4 > $O/pmc_w_i8.log 2>&1
echo done

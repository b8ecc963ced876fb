set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2e1
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_decoder_gpu.py tests/test_quant_gpu.py tests/test_packed_gpu.py -q -x > $O/test.log 2>&1 || { tail -40 $O/test.log; exit 1; }
tail -3 $O/test.log
timeout -k 10 300 python3 tools/opprofile.py int8 32 128 2>&1 | grep -v amdgpu.ids > $O/op_i8_b32.log; cat $O/op_i8_b32.log
timeout -k 10 300 python3 tools/opprofile.py f16 32 512 2>&1 | grep -v amdgpu.ids > $O/op_f16_b32.log; cat $O/op_f16_b32.log

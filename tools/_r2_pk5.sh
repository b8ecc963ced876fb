set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2pk5
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_packed_gpu.py -q > $O/test.log 2>&1 || { tail -30 $O/test.log; }
tail -3 $O/test.log
LLMIE_STAMPS_LIB=$R/tools/libllmie_stamps.so timeout -k 10 300 python3 tools/pkstamps.py 2>&1 | grep -v amdgpu.ids > $O/stamps.log; head -12 $O/stamps.log
timeout -k 10 300 python3 tools/pksweep.py 32 2>&1 | grep -E "plain|norm" > $O/sweep32.log; cat $O/sweep32.log
timeout -k 10 300 python3 tools/pkbench.py int8 M=32 2>&1 | grep -v amdgpu.ids > $O/pkbench.log; cat $O/pkbench.log

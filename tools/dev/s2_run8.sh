cd $GRAFT_REPO_ROOT
bash tools/dev/s2_lx.sh | grep -c "ok"
timeout -k 10 1100 python -m pytest tests/test_decoder_gpu.py tests/test_kernels_gpu.py tests/test_packed_gpu.py tests/test_chain_gpu.py tests/test_quant_gpu.py tests/test_prefill_gpu.py tests/test_fullsize_gpu.py -x -q > gpurun_out/s2_t8.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/s2_t8.log
tail -4 gpurun_out/s2_t8.log | cut -c1-600
[ $rc -ne 0 ] && exit 1
bash tools/dev/kt_quick.sh decode:f16:1:2048 decode:int8:32:128 2>&1 | grep -i "topk\|tail\|pk_mfma\|attn_split\|==" | cut -c1-60,150-230

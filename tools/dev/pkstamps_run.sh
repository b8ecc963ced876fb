# dev: PK_STAMPS build of the library on the GPU box + tools/pkstamps.py (x32 input)
cd $GRAFT_REPO_ROOT
C=llm-inference-engine_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -DPK_STAMPS -x hip -c $C/pk_linear.hip -o /tmp/pk_stamps.o 2>/dev/null || exit 1
objs=$(ls $C/_obj/*.o | grep -v pk_linear)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libllmie_stamps.so /tmp/pk_stamps.o $objs || exit 1
LLMIE_STAMPS_LIB=/tmp/libllmie_stamps.so PK_X32=1 timeout -k 5 200 python tools/pkstamps.py

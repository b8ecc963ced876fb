cd $GRAFT_REPO_ROOT
for T in 192 256 512 768 1024 2048; do timeout -k 10 200 python3 bench.py --only prefill:fp8:1:$T 2>/dev/null | tail -1 | cut -c1-330; done
for T in 256 512 1024; do timeout -k 10 200 python3 bench.py --only prefill:int4:1:$T 2>/dev/null | tail -1 | cut -c1-330; done

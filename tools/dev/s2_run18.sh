cd $GRAFT_REPO_ROOT
for T in 160 192 256 384 512 768 1024 1536 2048 3072; do timeout -k 10 200 python3 bench.py --only prefill:f16:1:$T 2>/dev/null | tail -1 | cut -c1-330; done

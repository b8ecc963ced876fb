cd $GRAFT_REPO_ROOT
for i in 1 2; do for c in prefill:f16:1:1024 prefill:f16:2:512 prefill:int8:1:1024 prefill:f16:1:2048; do timeout -k 10 200 python3 bench.py --only $c 2>/dev/null | tail -1 | cut -c1-330; done; done

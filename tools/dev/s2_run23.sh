cd $GRAFT_REPO_ROOT
for c in decode:f16:4:512 decode:f16:5:512 decode:f16:6:512 decode:int8:3:512 decode:int8:4:512 decode:fp8:3:512 decode:fp8:4:512 decode:int4:2:512; do timeout -k 10 200 python3 bench.py --only $c 2>/dev/null | tail -1 | cut -c1-100; done

cd $GRAFT_REPO_ROOT
for B in 2 3 4 8 16 32 33 48 64 128; do timeout -k 10 200 python3 bench.py --only decode:int8:$B:512 2>/dev/null | tail -1 | cut -c1-100; done
for S in 128 512 1024 2048 4000; do timeout -k 10 200 python3 bench.py --only decode:f16:1:$S 2>/dev/null | tail -1 | cut -c1-100; done

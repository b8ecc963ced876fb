cd $GRAFT_REPO_ROOT
for B in 1 2 3 4 6 8 12 16 24 32 48 64 96 128; do timeout -k 10 200 python3 bench.py --only decode:f16:$B:512 2>/dev/null | tail -1 | cut -c1-260; done

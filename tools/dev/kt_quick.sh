export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in "$@"; do
  tag=$(echo $c | tr ':' '_')
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ktq_$tag -- python3 bench.py --only $c > gpurun_out/ktq_$tag.log 2>&1
  python3 tools/summarize_prof.py stats gpurun_out/ktq_$tag gpurun_out/ktq_$tag.csv "$c" > /dev/null
  echo "== $c"; grep -v "^#" gpurun_out/ktq_$tag.csv | cut -c1-160 | head -24
done

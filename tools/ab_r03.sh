cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for tree in . _r03tree; do
  echo "== tree $tree"
  (cd $tree && for n in 8192 12288 16384 20480; do timeout 100 python tools/prof_chol.py $n 5 | sort -t' ' -k6 -n | head -1; done)
done
done
for tree in . _r03tree . _r03tree; do
  echo "== c3 tree $tree"
  (cd $tree && timeout 200 python bench.py --config c3 --steps 5 --warmup 1 --no-cpu-baseline --no-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])")
  (cd $tree && timeout 200 python tools/sweep_sizes.py 2>/dev/null | tail -3)
done

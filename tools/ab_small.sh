cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for tree in . _r03tree; do
  echo "== tree $tree"
  (cd $tree && for n in 2048 4096 8192; do timeout 100 python tools/prof_chol.py $n 6 | sort -t' ' -k6 -n | head -1; done)
done
done

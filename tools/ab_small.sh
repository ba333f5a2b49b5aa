# Same-box comparison of this tree with the tree of the previous round (boxes differ by 2-3 percent, so only side-by-side numbers count).
# Set-up, once, in the repository root (the directory is git-ignored but travels with a gpurun snapshot; remove it afterwards):
#   git worktree add _r03tree 2be82c6 && (cd _r03tree && python -c "import __graft_entry__ as g; g.build()")
#   gpurun -- 'bash tools/ab_small.sh'
#   git worktree remove --force _r03tree
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for tree in . _r03tree; do
  echo "== tree $tree"
  (cd $tree && for n in 2048 4096 8192; do timeout 100 python tools/prof_chol.py $n 6 | sort -t' ' -k6 -n | head -1; done)
done
done

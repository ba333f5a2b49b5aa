cd $GRAFT_REPO_ROOT
for v in 0 1 2; do
  export LGP_CHAIN_FAR_RECURSIVE=$v
  echo "== chain far recursive $v"
  for n in 4096 8192 12288; do timeout 100 python tools/prof_chol.py $n 4 | tail -1; done
done

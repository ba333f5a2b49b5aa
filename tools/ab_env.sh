# Same-box A/B of one environment switch:  bash tools/ab_env.sh "VAR=a" "VAR=b" [sizes...]   (Cholesky alone, best of 5, alternating)
cd $GRAFT_REPO_ROOT
A="$1"; B="$2"; shift 2
SIZES="${@:-2048 4096 8192 12288 16384}"
for rep in 1 2; do
  for setting in "$A" "$B"; do
    echo "== $setting"
    for n in $SIZES; do env $setting timeout 100 python tools/prof_chol.py $n 6 | sort -t' ' -k6 -n | head -1; done
  done
done

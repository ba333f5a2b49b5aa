cd $GRAFT_REPO_ROOT
for v in none far far4096 far16384; do
  unset LGP_U2_T128_MIN_ROWS LGP_U2_STREAM
  case $v in far) export LGP_U2_STREAM=far;; far4096) export LGP_U2_STREAM=far LGP_U2_T128_MIN_ROWS=4096;; far16384) export LGP_U2_STREAM=far LGP_U2_T128_MIN_ROWS=16384;; esac
  echo "== $v"
  timeout 200 python bench.py --config c3 --steps 4 --warmup 1 --no-cpu-baseline --no-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['log_marginal_likelihood'])"
done

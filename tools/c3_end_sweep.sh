# the switches that shape the END of a large factorisation, swept on the C3 step of this tree (one box):  bash tools/c3_end_sweep.sh
cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 200 python bench.py --config c3 --steps 4 --warmup 1 --no-cpu-baseline --no-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), round(d['roofline']['frac'],4), d['roofline']['launches']//4)"; }
for rep in 1 2; do
for t64 in 2500 5000 10000 20000; do
  for tail in 8192 12288; do
    echo -n "64-tile updates up to $t64 tiles, chain tail $tail: "
    run LGP_GEMM_T64_TRAILING_MAX=$t64 LGP_SPLIT_CHAIN_TAIL=$tail
  done
done
done

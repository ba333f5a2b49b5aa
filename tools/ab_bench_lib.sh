# Same-box A/B of two BUILDS of the library on the C3 step:  bash tools/ab_bench_lib.sh <variant>     (python -m lsqfitgp_amd._build <variant> DEFINE... first)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in "" "$GRAFT_REPO_ROOT/lsqfitgp_amd/liblsqfitgp_hip_$1.so"; do
    echo -n "library ${lib:-default}: "
    LGP_LIBRARY=$lib timeout 200 python bench.py --config c3 --steps 5 --warmup 1 --no-cpu-baseline --no-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'ms, trailing update', round(d['roofline']['frac'],4), 'of peak')"
  done
done

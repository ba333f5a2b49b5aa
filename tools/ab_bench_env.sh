# Same-box A/B of one environment switch on the C3 step:  bash tools/ab_bench_env.sh "VAR=a" "VAR=b"   (bench.py c3, 5 steps, alternating twice)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for setting in "$1" "$2"; do
    echo -n "$setting: "
    env $setting timeout 200 python bench.py --config c3 --steps 5 --warmup 1 --no-cpu-baseline --no-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'ms, trailing update', round(d['roofline']['frac'],4), 'of peak')"
  done
done

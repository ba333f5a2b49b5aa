#!/bin/bash
# L2 counters of one trailing update in the in-situ layout: tools/pmc_gemm_l2.sh <outdir> <rows> <pad> (run on the GPU box)
set -e
out=${1:-gpurun_out/pmc_l2}
rows=${2:-16384}
pad=${3:-0}
mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 tools/bench_gemm_insitu_one.py $rows 2 32768 $pad > $out/p$i.log 2>&1 || echo "pass $i ($set) failed"
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(f'{k:28s} n={len(v)} last={v[-1]:.6g} mean={sum(v)/len(v):.6g}')
PY

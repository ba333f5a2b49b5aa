#!/bin/bash
# how much of the L2's fabric read traffic of one trailing update goes to DRAM (the rest: Infinity Cache): tools/pmc_dram.sh <rows>
export TMPDIR=/tmp
rows=${1:-16384}
out=gpurun_out/pmc_dram$rows
mkdir -p $out
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum" "TCC_BUBBLE_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 tools/bench_gemm_insitu_one.py $rows 2 > $out/p$i.log 2>&1 || echo "pass $i ($set) failed"
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

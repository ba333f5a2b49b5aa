#!/bin/bash
# kernel timeline of ONE checks-off prediction of the reference's small benchmark at N (default 1000):  bash tools/trace_small_n.sh [N]
N=${1:-1000}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/trace_small_$N
rm -rf $out && mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -- python3 tools/profile_small_n.py $N > $out/run.txt 2>&1
f=$(find $out -name '*kernel_trace.csv' | sort | tail -1)
python3 - $f <<'PY' > $out/timeline.txt
import csv, sys
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1])))
# the last prediction: from the last-but-one fused Gram kernel on
starts = [i for i, r in enumerate(rows) if 'gram_iso_kernel' in r[2] or 'predict_small_kernel' in r[2]]
i0 = starts[-2] if len(starts) >= 2 and 'gram_iso' in rows[starts[-1]][2] else starts[-1]
sel = rows[i0:]
t0 = sel[0][0]
prev = t0
for s, e, n in sel:
    n = n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    print(f'{(s - t0) / 1e3:9.1f} us  +{(s - prev) / 1e3:7.1f} gap  {(e - s) / 1e3:7.1f} us  {n[:80]}')
    prev = e
print(f'span {(sel[-1][1] - t0) / 1e3:.1f} us, kernels {len(sel)}, busy {sum(e - s for s, e, _ in sel) / 1e3:.1f} us')
PY
head -4 $out/run.txt; cat $out/timeline.txt

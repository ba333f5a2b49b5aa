#!/bin/bash
# kernel trace of C3 steps: for the last step, the large updates (128-tile), what runs between them on their queue and how long the queue idles:  bash tools/trace_c3_gaps.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/trace_c3
rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --no-sweep > $out/bench.json 2> $out/run.log
f=$(find $out -name '*kernel_trace.csv' | sort | tail -1)
python3 - $f <<'PY' > $out/gaps.txt
import csv, sys
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in csv.DictReader(open(sys.argv[1])))
starts = [i for i, r in enumerate(rows) if 'gram_reg_kernel' in r[2]]
sel = rows[starts[-1]:]
t0 = sel[0][0]
big = [r for r in sel if '128, 256, 1' in r[2]]
# (bench.py times the stand-alone Gram kernel after its steps: the step ends where those launches begin)
after = [r[0] for r in sel if 'gram_iso_kernel' in r[2] and r[0] > big[-1][1]]
if after:
    sel = [r for r in sel if r[1] <= after[0]]
q = big[0][3]
onq = [r for r in sel if r[3] == q]
print(f'step span {(max(r[1] for r in sel) - t0) / 1e3:.0f} us; queue {q} carries the large updates')
prev_end = None
tot_gap = tot_other = 0.
for k, b in enumerate(big):
    if prev_end is not None:
        between = [r for r in onq if r[0] >= prev_end and r[1] <= b[0]]
        busy = sum(r[1] - r[0] for r in between) / 1e3
        gap = (b[0] - prev_end) / 1e3
        tot_gap += gap - busy; tot_other += busy
        print(f'update {k:2d}: starts {(b[0] - t0) / 1e3:9.0f} us, runs {(b[1] - b[0]) / 1e3:8.0f} us; since the previous one {gap:7.0f} us of which other kernels of the queue {busy:7.0f} us, idle {gap - busy:7.0f} us')
    else:
        print(f'update {k:2d}: starts {(b[0] - t0) / 1e3:9.0f} us, runs {(b[1] - b[0]) / 1e3:8.0f} us')
    prev_end = b[1]
print(f'between the large updates: {tot_other:.0f} us of other kernels on their queue, {tot_gap:.0f} us idle; after the last: {(max(r[1] for r in sel) - prev_end) / 1e3:.0f} us; before the first: {(big[0][0] - t0) / 1e3:.0f} us')
names = {}
for s, e, n, qq in sel:
    n = n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:60]
    a = names.setdefault(n, [0, 0., 0.]); a[0] += 1; a[1] += (e - s) / 1e3; a[2] = max(a[2], (e - s) / 1e3)
for n, a in sorted(names.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f'{n:62s} n={a[0]:4d} total {a[1]:9.0f} us avg {a[1] / a[0]:8.1f} max {a[2]:8.0f}')
PY
cat $out/gaps.txt

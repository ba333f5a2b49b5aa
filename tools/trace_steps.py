"""per-step view of a rocprofv3 kernel trace of bench.py: python tools/trace_steps.py <kernel_trace.csv> [steps...]"""
import csv, sys
f = sys.argv[1]
steps = [int(v) for v in sys.argv[2:]] or [0, 10, 20]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if '128, 256, 1>' in r[2]]
last = idx[-29:]
def short(n):
    return n.replace('void (anonymous namespace)::', '')[:58]
prev_end = None
tot_gap = 0
for k, i in enumerate(last):
    s, e = rows[i][0], rows[i][1]
    if prev_end is not None:
        tot_gap += s - prev_end
    prev_end = e
print(f'sum of trailing launches {sum(rows[i][1]-rows[i][0] for i in last)/1e6:.1f} ms, sum of gaps between them {tot_gap/1e6:.1f} ms')
for step in steps:
    i = last[step]; s, e = rows[i][0], rows[i][1]
    nxt = rows[last[step + 1]][0] if step + 1 < len(last) else e
    print(f'--- step {step}: trailing {(e-s)/1e6:.3f} ms, gap to next {(nxt-e)/1e3:.0f} us; kernels overlapping it:')
    agg = {}
    for r in rows:
        if r[0] < e and r[1] > s and r is not rows[i]:
            k = short(r[2]); a = agg.setdefault(k, [0, 0.0, 0]); a[0] += 1; a[1] += (r[1]-r[0])/1e3; a[2] = max(a[2], (r[1]-r[0])/1e3)
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'   {k:60s} n={a[0]:3d} total {a[1]:8.1f} us  max {a[2]:7.1f} us')

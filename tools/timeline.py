"""timeline of the LAST factorisation in a rocprofv3 kernel trace:  python tools/timeline.py <kernel_trace.csv> [max_lines]
One line per kernel in start order: start (us since the first kernel of the factorisation), duration, queue, name, and per
queue the idle time since that queue's previous kernel.  Ends with per-kernel-name totals and the busy time of each queue."""
import csv, sys
f = sys.argv[1]
maxl = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')))
rows.sort()
def short(n):
    n = n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    return n[:70]
# the last factorisation starts at the last kernel whose name contains one of these (prep of Chol.__init__)
starts = [i for i, r in enumerate(rows) if 'diag_scale_kernel' in r[2] or 'gram_reg_kernel' in r[2]]
i0 = starts[-1] if starts else 0
sel = rows[i0:]
t0 = sel[0][0]
last_end = {}
tot = {}
busy = {}
for k, (s, e, n, q) in enumerate(sel):
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    a = tot.setdefault(short(n), [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    busy[q] = busy.get(q, 0.0) + (e - s) / 1e3
    if k < maxl:
        print(f'{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} q{q:>3s} idle {gap:7.1f}  {short(n)}')
print(f'--- total span {(max(r[1] for r in sel) - t0) / 1e3:.1f} us')
for q, b in sorted(busy.items()):
    print(f'queue {q}: busy {b:.1f} us')
for n, a in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f'{n:72s} n={a[0]:4d} total {a[1]:9.1f} us avg {a[1] / a[0]:7.1f}')

"""two-workgroup chain (walker.inc, wk_tail) from LGP_CHAIN_TRACE=<file>:  python tools/tail_trace.py <file>
Per block (us): factor workgroup: start, load, potf2 (to flags out), outputs; tail workgroup: start (loads), block steps done, handed over;
and the factor's period."""
import sys
rows = [[int(x) for x in l.split()] for l in open(sys.argv[1]) if not l.startswith('#')]
A = {(r[2], r[3]): r[6:10] for r in rows if r[1] == 3 and r[6] > 0}
B = {(r[2], r[3]): r[6:10] for r in rows if r[1] == 4 and r[6] > 0}
t0 = min(v[0] for v in A.values())
prev = None
print('block     f.start  f.load f.potf2  f.out |  t.start(rel f.start)  first rows in  steps 0-3  steps 4-7  (end rel f.flags) | period')
for key in sorted(A):
    a = A[key]; b = B.get(key)
    per = '' if prev is None else f'{(a[0] - prev) / 100:7.1f}'
    line = f'k{key[0]} j{key[1]}: {(a[0] - t0) / 100:9.1f} {(a[1] - a[0]) / 100:7.1f} {(a[2] - a[1]) / 100:7.1f} {(a[3] - a[2]) / 100:6.1f} |'
    if b:
        line += f' {(b[0] - a[0]) / 100:9.1f} {(b[1] - b[0]) / 100:12.1f} {(b[2] - b[1]) / 100:10.1f} {(b[3] - b[2]) / 100:10.1f} {(b[3] - a[2]) / 100:10.1f} |'
    print(line + ' ' + per)
    prev = a[0]

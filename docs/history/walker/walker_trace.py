"""phases of the walker (csrc/walker_body.inc) from LGP_CHAIN_TRACE=<file>:  python tools/walker_trace.py <file>
Per block of each panel's square (us): tile load (first block of a square only), potf2, outputs + wait for the next block's rows,
their load, solve, store + park + flags, load of the next block, its update; and the period."""
import sys
rows = [[int(x) for x in l.split()] for l in open(sys.argv[1]) if not l.startswith('#')]
A = {(r[2], r[3]): r[6:10] for r in rows if r[1] == 3 and r[9] > 0}
B = {(r[2], r[3]): r[6:10] for r in rows if r[1] == 4 and r[9] > 0}
t0 = min(v[0] for v in A.values())
prev = None
print('block      start   load  potf2 out+wait  xload  solve  store  dload update | period')
for key in sorted(A):
    a = A[key]; b = B.get(key, [a[3]] * 4)
    st = [a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]]
    d = [(st[i + 1] - st[i]) / 100 for i in range(7)]
    per = '' if prev is None else f'{(st[0] - prev) / 100:7.1f}'
    print(f'k{key[0]} j{key[1]}: {(st[0] - t0) / 100:9.1f} ' + ' '.join(f'{x:6.1f}' for x in d) + ' | ' + per)
    prev = st[0]

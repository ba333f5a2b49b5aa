"""timeline of the chain kernel from LGP_CHAIN_TRACE=<file>:  python tools/chain_trace.py <file> [panel]
Per task of one panel: when it was taken, how long it waited for its inputs, how long it ran (us), and which worker ran it;
then the dependent chain P(j) -> S(2j+2|2j+3, j) -> G(.., j) -> P(j+1) with the hand-over gaps."""
import sys
rows = [l.split() for l in open(sys.argv[1]) if not l.startswith('#')]
panel = int(sys.argv[2]) if len(sys.argv) > 2 else 1
T = {0: 'P', 1: 'S', 2: 'G'}
tasks = []
for t, typ, k, j, r, c, wk, taken, ready, done in rows:
    if int(k) == panel and int(done) > 0:
        tasks.append((int(taken), T[int(typ)], int(j), int(r), int(c), int(wk), int(ready), int(done)))
if not tasks:
    sys.exit('no tasks of that panel in the trace')
t0 = min(t[0] for t in tasks)
us = lambda x: (x - t0) / 100.0
print('   taken   waited      ran  worker task')
for taken, typ, j, r, c, wk, ready, done in sorted(tasks)[:120]:
    print(f'{us(taken):8.1f} {(ready - taken) / 100:8.1f} {(done - ready) / 100:8.1f}   w{wk:<3d} {typ}(j={j}, r={r}, c={c})')
by = {(typ, j, r, c): (taken, ready, done) for taken, typ, j, r, c, wk, ready, done in tasks}
print('--- chain: end of P(j) -> S ready .. done -> G ready .. done -> next P ready .. done')
for j in range(8):
    if ('P', j, 0, 0) not in by:
        break
    p = by[('P', j, 0, 0)]
    line = f'P({j}) ran {(p[2] - p[1]) / 100:6.1f} (ready at {us(p[1]):8.1f})'
    s = [by.get(('S', j, 2 * j + 2 + h, 0)) for h in (0, 1)]
    if all(s):
        line += f' | S starts +{(min(x[1] for x in s) - p[2]) / 100:5.1f}, runs {(max(x[2] for x in s) - min(x[1] for x in s)) / 100:5.1f}'
        g = [by.get(('G', j, 2 * j + 2 + a, 2 * j + 2 + b)) for a, b in ((0, 0), (1, 0), (1, 1))]
        if all(g):
            line += f' | G starts +{(min(x[1] for x in g) - max(x[2] for x in s)) / 100:5.1f}, runs {(max(x[2] for x in g) - min(x[1] for x in g)) / 100:5.1f}'
            nxt = by.get(('P', j + 1, 0, 0))
            if nxt:
                line += f' | next P starts +{(nxt[1] - max(x[2] for x in g)) / 100:5.1f}'
    print(line)
print(f'panel span {(max(t[7] for t in tasks) - t0) / 100:.1f} us, tasks {len(tasks)}')
# ---- every panel: when its first P became ready after the previous panel's last P finished, and how long its chain took
allp = {}
for t, typ, k, j, r, c, wk, taken, ready, done in rows:
    if int(typ) == 0 and int(done) > 0:
        allp[(int(k), int(j))] = (int(ready), int(done))
ks = sorted({k for k, j in allp})
first = min(v[0] for v in allp.values())
print('--- panels: start of P(k,0) since the first task (us), hand-over from the previous panel, chain duration')
for k in ks:
    js = sorted(j for kk, j in allp if kk == k)
    a, b = allp[(k, js[0])], allp[(k, js[-1])]
    prev = allp.get((k - 1, max([j for kk, j in allp if kk == k - 1], default=0)))
    hand = (a[0] - prev[1]) / 100 if prev else 0.0
    print(f'panel {k:3d}: P(0) ready at {(a[0] - first) / 100:9.1f}, hand-over {hand:7.1f}, chain {(b[1] - a[0]) / 100:7.1f}')

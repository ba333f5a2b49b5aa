"""python tools/small_n.py -- the reference's own published benchmark (docs/userguide/optim.rst:88-159): `doinference` = GP(ExpQuad())
.addx(x, 'data').addx(xplot, 'plot').predfromdata({'data': data}, 'plot', raw=True) + sqrt(diag(cov)), N = 10 and 1000 data
points, 100 plot points, through the host classes; wall clock, best of 20 (the reference: best of 5 timeit repeats)."""
import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp


def doinference(data, **options):
    x = np.linspace(0, 10, len(data))
    xplot = np.linspace(0, 10, 100)
    mean, cov = lgp.GP(lgp.ExpQuad(), **options).addx(x, 'data').addx(xplot, 'plot').predfromdata({'data': data}, 'plot', raw=True)
    return mean, np.sqrt(np.diag(cov))


def best(f, *a, reps=20, **k):
    f(*a, **k)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(*a, **k); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3


if __name__ == '__main__':
    lgp.Context.default()
    for n in (10, 1000):
        data = np.zeros(n)
        print(f'N = {n}: default options best %.3f ms (median %.3f)' % best(doinference, data))
        print(f'N = {n}: checkpos=False, checksym=False best %.3f ms (median %.3f)' % best(doinference, data, checkpos=False, checksym=False))
    if len(sys.argv) > 1:
        data = np.zeros(1000)
        pr = cProfile.Profile(); pr.enable()
        for _ in range(50):
            doinference(data, **({} if len(sys.argv) > 2 else dict(checkpos=False, checksym=False)))
        pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(35)

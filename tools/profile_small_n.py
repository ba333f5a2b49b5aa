"""python tools/profile_small_n.py [N] -- where the time of the reference's small-N benchmark goes (docs/userguide/optim.rst:88-159), checks off"""
import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ctx = lgp.Context.default()
x = np.linspace(0, 10, n); xplot = np.linspace(0, 10, 100); data = np.zeros(n)
def doinference():
    mean, cov = lgp.GP(lgp.ExpQuad(), ctx=ctx, checkpos=False, checksym=False).addx(x, 'data').addx(xplot, 'plot').predfromdata({'data': data}, 'plot', raw=True)
    return mean, np.sqrt(np.diag(cov))
for _ in range(5):
    doinference()
ts = []
for _ in range(30):
    t0 = time.perf_counter(); doinference(); ts.append((time.perf_counter() - t0) * 1e3)
print('N = %d: best %.3f ms, median %.3f ms' % (n, min(ts), sorted(ts)[15]))
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    doinference()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)

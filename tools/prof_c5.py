"""host-side profile of the C5 step (new GP -> predfromdata + marginal_likelihood through the host classes): python tools/prof_c5.py"""
import sys, cProfile, pstats, time
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
from workloads import load_abalone, c5_prepare, c5_gp
ctx = lgp.Context.default()
X, y = load_abalone()
n = 4096
prep = c5_prepare(lgp, X, y, n)
given = prep['given']
def step():
    gp = c5_gp(lgp, prep, diag=lgp.diag)
    mean, cov = gp.predfromdata(given, 'testmean', raw=True)
    return mean, cov, gp.marginal_likelihood(given, epsrel=0)
for _ in range(3):
    step()
ctx.sync()
t0 = time.perf_counter()
for _ in range(20):
    step()
ctx.sync()
print('ms per step', (time.perf_counter() - t0) / 20 * 1e3)
# the same with the GPU made to wait at the end only: host time alone = time until the last call returns
t0 = time.perf_counter()
gp = c5_gp(lgp, prep, diag=lgp.diag)
t1 = time.perf_counter()
mean, cov = gp.predfromdata(given, 'testmean', raw=True)
t2 = time.perf_counter()
ml = gp.marginal_likelihood(given, epsrel=0)
t3 = time.perf_counter()
print('new GP %.2f ms, predfromdata %.2f ms, marginal_likelihood %.2f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(30)

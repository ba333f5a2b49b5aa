"""python tools/profile_default_path.py [config] -- where the host time of the reference call sequence goes (cProfile of bench.default_path's step)"""
import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np
import bench, lsqfitgp_amd as lgp
cfgname = sys.argv[1] if len(sys.argv) > 1 else 'c2'
ctx = lgp.Context.default()
cfg = bench.CONFIGS[cfgname]
n, m = cfg['n'], cfg['m']
x, xs, y = bench.synth(cfg)
kern = lgp.ExpQuad() if cfg['kernel'] == 'expquad' else lgp.Maternp(p=2, scale=1.)
a, b = x[0], xs[0]
gc = {('data', 'data'): lgp.diag(np.full(n, 0.01))}
def step():
    gp = lgp.GP(kern, checkpos=False, checksym=False, ctx=ctx).addx(a, 'data').addx(b, 'test')
    mean, cov = gp.predfromdata({'data': y}, 'test', gc, raw=True)
    return gp.marginal_likelihood({'data': y}, gc)
for _ in range(3):
    step()
ctx.sync()
ts = []
for _ in range(10):
    t0 = time.perf_counter(); step(); ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3)
print('step ms: min %.3f median %.3f' % (min(ts), sorted(ts)[5]))
# the same with the two calls timed apart
tp, tm = [], []
for _ in range(10):
    gp = lgp.GP(kern, checkpos=False, checksym=False, ctx=ctx).addx(a, 'data').addx(b, 'test')
    t0 = time.perf_counter(); gp.predfromdata({'data': y}, 'test', gc, raw=True); t1 = time.perf_counter()
    gp.marginal_likelihood({'data': y}, gc); t2 = time.perf_counter()
    tp.append((t1 - t0) * 1e3); tm.append((t2 - t1) * 1e3)
print('predfromdata ms: min %.3f   marginal_likelihood ms: min %.3f' % (min(tp), min(tm)))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
ctx.sync()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)

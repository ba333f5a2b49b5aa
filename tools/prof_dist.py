""" phases of the sharded path with one rank on a BASELINE configuration (gram / regularisation / factorisation), next to the
single-GPU fused step: python tools/prof_dist.py [c3|c2|c4] """
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['LGP_DIST_TIMING'] = '1'
import numpy as np
import bench
import lsqfitgp_amd as lgp
from lsqfitgp_amd._dist import DistributedChol
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'c3']
ctx = lgp.Context(0)
lgp.Context._default = ctx
x, xs, y = bench.synth(cfg)
kern = lgp.ExpQuad() if cfg['kernel'] == 'expquad' else lgp.Maternp(p=2, scale=1.)
if cfg['d'] == 1:
    pts, tpts = x[0], xs[0]
else:
    dt = [(f'f{i}', float) for i in range(cfg['d'])]
    pts = np.empty(cfg['n'], dt); tpts = np.empty(cfg['m'], dt)
    for i, f in enumerate(pts.dtype.names):
        pts[f] = x[i]; tpts[f] = xs[i]
for rep in range(4):
    t0 = time.perf_counter()
    d = DistributedChol(kern, pts, noise=0.01, nb=1024, ctx=ctx, carry=(tpts, y))
    t1 = time.perf_counter()
    mean, cov, mlnd = d.posterior(kern, pts, tpts, y)
    ctx.sync()
    t2 = time.perf_counter()
    print(f'rep {rep}: total {(t2 - t0) * 1e3:.1f} ms  ctor {(t1 - t0) * 1e3:.1f}  posterior {(t2 - t1) * 1e3:.1f}  phases', {k: round(v, 2) for k, v in d.phases.items()}, 'mll', -mlnd, flush=True)
    del d

import os, sys, time, numpy as np
sys.path.insert(0, '/root/repo')
os.environ['LGP_DIST_TIMING']='1'
import bench, lsqfitgp_amd as lgp
from lsqfitgp_amd._dist import DistributedChol
ctx = lgp.Context(0); lgp.Context._default = ctx
cfg = bench.CONFIGS['c3']
x, xs, y = bench.synth(cfg)
dt=[(f'f{i}', float) for i in range(3)]
pts=np.empty(cfg['n'],dt); tp=np.empty(cfg['m'],dt)
for i,f in enumerate(pts.dtype.names): pts[f]=x[i]; tp[f]=xs[i]
k=lgp.Maternp(p=2,scale=1.)
for rep in range(3):
    t0=time.perf_counter()
    d=DistributedChol(k, pts, noise=0.01, nb=1024, ctx=ctx, carry=(tp,y))
    t1=time.perf_counter()
    m,c,l=d.posterior(k, pts, tp, y); ctx.sync()
    t2=time.perf_counter()
    print({a:round(b,2) for a,b in d.phases.items()}, 'ctor', round((t1-t0)*1e3,1), 'posterior', round((t2-t1)*1e3,1))
    del d

import sys, os, numpy as np
sys.path.insert(0, '.')
import bench
import lsqfitgp_amd as lgp
ctx = lgp.Context(0)
for cfgname, n in (('c2', 3000), ('c3', 5000), ('c2', 8192)):
    cfg = dict(bench.CONFIGS[cfgname]); cfg['n'] = n
    os.environ.pop('LGP_BENCH_SEPARATE_SOLVE', None)
    a = bench.Workload(cfg, ctx); a.step(); ctx.sync(); ma, ca = a.posterior_host(); ra = a.results
    os.environ['LGP_BENCH_SEPARATE_SOLVE'] = '1'
    b = bench.Workload(cfg, ctx); b.step(); ctx.sync(); mb, cb = b.posterior_host(); rb = b.results
    print(cfgname, n, 'mean rel', np.linalg.norm(ma - mb) / np.linalg.norm(mb), 'cov', np.linalg.norm(ca - cb, 2) / np.linalg.norm(cb, 2), 'mll', ra['mll'], rb['mll'], 'eps', ra['eps'] == rb['eps'])

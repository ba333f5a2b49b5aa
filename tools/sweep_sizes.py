"""python tools/sweep_sizes.py -- bench.py's smaller-N steps (C3's kernel, carried rows) with more repetitions than the sweep takes"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import bench, lsqfitgp_amd as lgp
ctx = lgp.Context.default()
for n in (4096, 8192, 16384):
    c = dict(bench.CONFIGS['c3']); c['n'] = n
    wl = bench.Workload(c, ctx)
    wl.step(); ctx.sync()
    ts, chol = [], []
    for k in range(8):
        ev = [ctx.event() for _ in range(4)]
        t0 = time.perf_counter(); wl.step(ev); ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3)
        chol.append(ctx.elapsed_ms(ev[1], ev[2]))
    print(f'N={n}: step min {min(ts):.3f} median {sorted(ts)[4]:.3f} ms; cholesky phase min {min(chol):.3f} median {sorted(chol)[4]:.3f} ms', flush=True)

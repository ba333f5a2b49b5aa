"""Maternp(2) 3-D Gram of a point set with itself, N = 32768: python tools/bench_gram_sym.py   (default: one triangle + mirror; LGP_GRAM_FULL=1: the full grid)"""
import sys
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ctx = lgp.Context.default()
rng = np.random.default_rng(0)
e0, e1 = ctx.event(), ctx.event()
k = lgp.Maternp(p=2)
x = rng.uniform(0, 10, (3, n))
xs = np.empty(n, [(f'f{i}', float) for i in range(3)])
for i in range(3):
    xs[f'f{i}'] = x[i]
dp = k.prepare(xs, ctx=ctx)
out = lgp.DeviceArray.empty((n, n))
ts = []
for _ in range(12):
    ctx.record(e0); k.gram(dp, dp, ctx=ctx, out=out); ctx.record(e1)
    ts.append(ctx.elapsed_ms(e0, e1))
print(f'N={n}: best {min(ts):.3f} ms = {8. * n * n / min(ts) / 1e6:.0f} GB/s written; median {sorted(ts)[6]:.3f} ms', flush=True)

"""stand-alone isotropic Gram matrix, one triangle (xs is ys) against the full grid: python tools/bench_gram_iso.py [N]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ctx = lgp.Context.default()
rng = np.random.default_rng(0)
e0, e1 = ctx.event(), ctx.event()
for name, k, d in (('Maternp(2) 3-D', lgp.Maternp(p=2), 3), ('ExpQuad 1-D', lgp.ExpQuad(), 1), ('ExpQuad 3-D', lgp.ExpQuad(), 3)):
    x = rng.uniform(0, 10, (d, n))
    xs = np.empty(n, [(f'f{i}', float) for i in range(d)]) if d > 1 else x[0]
    if d > 1:
        for i in range(d):
            xs[f'f{i}'] = x[i]
    dp = k.prepare(xs, ctx=ctx)
    dq = k.prepare(xs, ctx=ctx)
    out = lgp.DeviceArray.empty((n, n))
    for label, a, b in (('one triangle', dp, dp), ('full grid', dp, dq)):
        best = 1e9
        for _ in range(6):
            ctx.record(e0); k.gram(a, b, ctx=ctx, out=out); ctx.record(e1)
            best = min(best, ctx.elapsed_ms(e0, e1))
        print(f'{name:16s} N={n} {label:12s}: {best:.3f} ms = {8. * n * n / best / 1e6:.0f} GB/s written', flush=True)

"""the BART Gram kernel of configuration C5 alone (p = 10, maxd = 10, reset = [2, 4, 6, 8]: one depth-2 closed form iterated 5 times,
fp32, N x N on the first 4096 rows of bart-data.txt), bin indices uploaded once: python tools/bench_bart_gram.py [reps]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
from workloads import load_abalone, c5_prepare
ctx = lgp.Context(0); lgp.Context._default = ctx
X, y = load_abalone()
n = 4096
prep = c5_prepare(lgp, X, y, n)
kern, pts = prep['kern'], prep['pts']
dp = kern.prepare(pts[:n], ctx=ctx)
out = lgp.DeviceArray.empty((n, n), np.float32, ctx)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
e0, e1 = ctx.event(), ctx.event()
kern.gram(dp, dp, ctx=ctx, out=out)
best = 1e9
for _ in range(reps):
    ctx.record(e0); kern.gram(dp, dp, ctx=ctx, out=out); ctx.record(e1)
    best = min(best, ctx.elapsed_ms(e0, e1))
print(f'bart gram {n} x {n} p=10 fp32: {best:.3f} ms  {n * n / best / 1e6:.1f} G pairs/s')

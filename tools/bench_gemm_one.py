"""one gemm shape, for PMC runs: python tools/bench_gemm_one.py m n k lower reps"""
import sys
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
from lsqfitgp_amd import _lib
m, n, k, lower, reps = [int(v) for v in sys.argv[1:6]]
ctx = lgp.Context.default(); lib = _lib.load()
rng = np.random.default_rng(0)
A = lgp.DeviceArray.empty((m, k)); C = lgp.DeviceArray.zeros((m, n))
blk = rng.standard_normal((1024, k))
for r0 in range(0, m, 1024):
    r1 = min(r0 + 1024, m); A.rows(r0, r1).assign(lgp.DeviceArray.from_host(blk[:r1 - r0]))
e0, e1 = ctx.event(), ctx.event()
for r in range(reps):
    ctx.record(e0)
    _lib.check(lib.lgp_update_nt_f64(ctx.handle, m, n, k, A.ptr, A.ld, A.ptr, A.ld, C.ptr, C.ld, lower))
    ctx.record(e1)
    ms = ctx.elapsed_ms(e0, e1)
    print(f'{ms:.3f} ms {2.0 * m * n * k * (0.5 if lower else 1) / ms / 1e9:.1f} TFLOP/s', flush=True)

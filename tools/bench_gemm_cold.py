"""does the trailing update depend on what ran before it?  Alternates two separate matrices so that every launch
finds its operands cold: python tools/bench_gemm_cold.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
from lsqfitgp_amd import _lib
ctx = lgp.Context.default(); lib = _lib.load()
N = 32768
mats = []
rng = np.random.default_rng(0)
blk = lgp.DeviceArray.from_host(rng.standard_normal((1024, N)))
for _ in range(2):
    L = lgp.DeviceArray.empty((N, N))
    for r0 in range(0, N, 1024):
        L.rows(r0, r0 + 1024).assign(blk)
    mats.append(L)
e0, e1 = ctx.event(), ctx.event()
rows = 30720
for mode in ('same matrix', 'alternating matrices'):
    ts = []
    for rep in range(6):
        L = mats[rep % 2] if mode.startswith('alt') else mats[0]
        P = L.block(N - rows, N, 0, 1024); C = L.block(N - rows, N, N - rows, N)
        ctx.record(e0)
        _lib.check(lib.lgp_update_nt_f64(ctx.handle, rows, rows, 1024, P.ptr, P.ld, P.ptr, P.ld, C.ptr, C.ld, 1))
        ctx.record(e1)
        ts.append(ctx.elapsed_ms(e0, e1))
    print(mode, ' '.join(f'{t:.2f}' for t in ts), 'ms ->', ' '.join(f'{rows * (rows + 1) * 1024 / t / 1e9:.1f}' for t in ts), 'TFLOP/s', flush=True)

"""trailing update with and without 128 carried rows: python tools/bench_gemm_rect.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
from lsqfitgp_amd import _lib
ctx = lgp.Context.default(); lib = _lib.load()
N = 32768
L = lgp.DeviceArray.empty((N + 128, N))
rng = np.random.default_rng(0)
blk = lgp.DeviceArray.from_host(rng.standard_normal((1024, N)))
for r0 in range(0, N, 1024):
    L.rows(r0, r0 + 1024).assign(blk)
L.rows(N, N + 128).assign(blk.rows(0, 128))
e0, e1 = ctx.event(), ctx.event()
for rows in (30720, 16384, 8192):
    for extra in (0, 128):
        P = L.block(N - rows, N + extra, 0, 1024)
        C = L.block(N - rows, N + extra, N - rows, N)
        best = 1e9
        for rep in range(4):
            ctx.record(e0)
            _lib.check(lib.lgp_update_nt_f64(ctx.handle, rows + extra, rows, 1024, P.ptr, P.ld, P.ptr, P.ld, C.ptr, C.ld, 1))
            ctx.record(e1)
            best = min(best, ctx.elapsed_ms(e0, e1))
        fl = 1024 * (rows * (rows + 1) + 2 * extra * rows)
        print(f'rows={rows} extra={extra}: {best:.3f} ms {fl / best / 1e9:.1f} TFLOP/s', flush=True)

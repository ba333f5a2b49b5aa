"""trailing-update gemm with the in-situ operand layout (panel = column slice of an N x N matrix): python tools/bench_gemm_ld.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
from lsqfitgp_amd import _lib
ctx = lgp.Context.default(); lib = _lib.load()
N = 32768
L = lgp.DeviceArray.empty((N, N))
rng = np.random.default_rng(0)
blk = lgp.DeviceArray.from_host(rng.standard_normal((1024, N)))
for r0 in range(0, N, 1024):
    L.rows(r0, r0 + 1024).assign(blk)
e0, e1 = ctx.event(), ctx.event()
for rows in (30720, 16384, 8192):
    P = L.block(N - rows, N, 0, 1024)            # rows x 1024 panel, ld = N
    C = L.block(N - rows, N, N - rows, N)        # trailing matrix, ld = N
    for rep in range(3):
        ctx.record(e0)
        _lib.check(lib.lgp_update_nt_f64(ctx.handle, rows, rows, 1024, P.ptr, P.ld, P.ptr, P.ld, C.ptr, C.ld, 1))
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1)
    print(f'rows={rows} ld={P.ld}: {ms:.3f} ms {rows * (rows + 1) * 1024 / ms / 1e9:.1f} TFLOP/s', flush=True)

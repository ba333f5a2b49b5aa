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

# the same update with the panel in a buffer of its own (the panel slots of the sharded factorisation): padded leading dimensions
for ldp in (1024, 1040, 1056, 1088, 1152, 1280, 2048 + 16):
    for rows in (30720, 16384):
        Pp = lgp.DeviceArray.empty((rows, ldp))
        Pp.block(0, rows, 0, 1024).assign(L.block(N - rows, N, 0, 1024))
        C = L.block(N - rows, N, N - rows, N)
        for rep in range(3):
            ctx.record(e0)
            _lib.check(lib.lgp_update_nt_f64(ctx.handle, rows, rows, 1024, Pp.ptr, Pp.ld, Pp.ptr, Pp.ld, C.ptr, C.ld, 1))
            ctx.record(e1)
            ms = ctx.elapsed_ms(e0, e1)
        print(f'rows={rows} panel apart, ldp={ldp}: {ms:.3f} ms {rows * (rows + 1) * 1024 / ms / 1e9:.1f} TFLOP/s', flush=True)
        del Pp

"""Cholesky with and without carried right-hand sides: python tools/prof_chol_rhs.py N [nrhs]"""
import sys, time, ctypes
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
from lsqfitgp_amd import _lib
n = int(sys.argv[1]); q = int(sys.argv[2]) if len(sys.argv) > 2 else 101
ctx = lgp.Context.default(); lib = _lib.load()
rng = np.random.default_rng(0)
x = rng.uniform(0, n / 16, n)
K = lgp.ExpQuad().gram(x, x)
_lib.check(lib.lgp_add_diag_f64(ctx.handle, n, K.ptr, K.ld, None, 0.01))
nx = -(-q // 128) * 128
L = lgp.DeviceArray.empty((n + nx, n)); W = lgp.DeviceArray.empty((lib.lgp_chol_dblk_size(n),))
L.block(n, n + nx, 0, n).fill_zero()
eps = ctypes.c_double()
e0, e1 = ctx.event(), ctx.event()
for label, fn in (('plain', lambda: lib.lgp_chol_factor_f64(ctx.handle, n, K.ptr, K.ld, 0., 0., 1, L.ptr, L.ld, W.ptr, ctypes.byref(eps))),
                  (f'carrying {nx} rows', lambda: lib.lgp_chol_factor_rhs_f64(ctx.handle, n, K.ptr, K.ld, 0., 0., 1, L.ptr, L.ld, nx, W.ptr, ctypes.byref(eps)))):
    for r in range(3):
        ctx.sync(); ctx.record(e0)
        info = fn()
        ctx.record(e1); ms = ctx.elapsed_ms(e0, e1)
        print(f'n {n} {label}: info {info} chol {ms:.2f} ms -> {n**3 / 3 / ms / 1e9:.1f} TFLOP/s', flush=True)

"""python tools/prof_chol_ld.py N PAD [reps] -- Cholesky as tools/prof_chol.py, the factor in a buffer with leading dimension N + PAD
(is a power-of-two row stride what the walker's block loads pay for?)"""
import sys, time, ctypes
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
from lsqfitgp_amd import _lib
n = int(sys.argv[1]); pad = int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = lgp.Context.default(); lib = _lib.load()
rng = np.random.default_rng(0)
x = rng.uniform(0, n / 16, n)
K = lgp.ExpQuad().gram(x, x)
_lib.check(lib.lgp_add_diag_f64(ctx.handle, n, K.ptr, K.ld, None, 0.01))
Lw = lgp.DeviceArray.empty((n, n + pad)); L = Lw.block(0, n, 0, n)
W = lgp.DeviceArray.empty((lib.lgp_chol_dblk_size(n),))
eps = ctypes.c_double()
e0, e1 = ctx.event(), ctx.event()
for r in range(reps):
    ctx.sync(); ctx.record(e0)
    info = lib.lgp_chol_factor_f64(ctx.handle, n, K.ptr, K.ld, 0., 0., 1, L.ptr, L.ld, W.ptr, ctypes.byref(eps))
    ctx.record(e1); ms = ctx.elapsed_ms(e0, e1)
    print('n', n, 'ld', L.ld, 'info', info, 'chol %.2f ms -> %.1f TFLOP/s' % (ms, n**3 / 3 / ms / 1e9), flush=True)

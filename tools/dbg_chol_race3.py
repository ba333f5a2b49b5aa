import sys, time, ctypes
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
from lsqfitgp_amd import _lib
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
A = rng.standard_normal((n, n))
K = A @ A.T / n + np.eye(n)
ctx = lgp.Context(0)
dec = lgp.Chol(K, ctx=ctx)
L, dblk = dec._L, dec._dblk
DA = lgp.DeviceArray
eye = np.eye(n)
F = ctypes.c_double
Xs, IKs, Gs = [], [], []
B = rng.standard_normal((n, n))
Bd = DA.from_host(B, ctx=ctx)
for rep in range(6):
    X = DA.from_host(eye, ctx=ctx)
    _lib.check(lib.lgp_trsm_f64(ctx.handle, n, L.ptr, L.ld, dblk.ptr, 0, X.ptr, X.ld, n), 'trsm')
    Xh = X.to_host()
    IK = DA.empty((n, n), ctx=ctx)
    _lib.check(lib.lgp_gemm_f64(ctx.handle, 1, 0, n, n, n, F(1), X.ptr, X.ld, X.ptr, X.ld, F(0), IK.ptr, IK.ld), 'gemm')
    IKh = IK.to_host()
    G = DA.empty((n, n), ctx=ctx)
    _lib.check(lib.lgp_gemm_f64(ctx.handle, 0, 0, n, n, n, F(1), Bd.ptr, Bd.ld, IK.ptr, IK.ld, F(0), G.ptr, G.ld), 'gemm')
    Gh = G.to_host()
    Xs.append(Xh); IKs.append(IKh); Gs.append(Gh)
    print(rep, 'X same', np.array_equal(Xh, Xs[0]), 'IK same', np.array_equal(IKh, IKs[0]), 'G same', np.array_equal(Gh, Gs[0]), flush=True)
Lh = L.to_host()
Xref = np.linalg.solve(Lh, eye)
print('X err', np.abs(Xs[0] - Xref).max(), 'IK err', np.abs(IKs[0] - Xs[0].T @ Xs[0]).max(), 'G err', np.abs(Gs[0] - B @ IKs[0]).max())

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
Lh = L.to_host()
DA = lgp.DeviceArray
F = ctypes.c_double
r = rng.standard_normal(n)
import scipy.linalg as sl
for q in (1, 2, 5):
    R = rng.standard_normal((n, q))
    yref = sl.solve_triangular(Lh, R, lower=True)
    zref = sl.solve_triangular(Lh.T, yref, lower=False)
    outs = []
    for rep in range(8):
        X = DA.from_host(R, ctx=ctx)
        _lib.check(lib.lgp_trsm_f64(ctx.handle, n, L.ptr, L.ld, dblk.ptr, 0, X.ptr, X.ld, q), 'trsm')
        y = X.to_host()
        _lib.check(lib.lgp_trsm_f64(ctx.handle, n, L.ptr, L.ld, dblk.ptr, 1, X.ptr, X.ld, q), 'trsm')
        z = X.to_host()
        outs.append((y, z))
        print(q, rep, 'y err %.2e' % (np.abs(y - yref).max() / np.abs(yref).max()), 'z err %.2e' % (np.abs(z - zref).max() / np.abs(zref).max()),
              'same', np.array_equal(y, outs[0][0]), np.array_equal(z, outs[0][1]), flush=True)
# the whole gradient call on synthetic derivative matrices, repeated
dK = [DA.from_host(rng.standard_normal((n, n)), ctx=ctx) for _ in range(2)]
dKh = [d.to_host() for d in dK]
for d in dKh:
    d += d.T
dK = [DA.from_host(d, ctx=ctx) for d in dKh]
Kinv = np.linalg.inv(Lh @ Lh.T)
z = Kinv @ r
gref = np.array([0.5 * (np.sum(Kinv * d) - z @ d @ z) for d in dKh])
for rep in range(8):
    v, g, _ = dec.normal_density_derivs(r, dK, None)
    print('grad', rep, np.asarray(g), 'ref', gref, flush=True)

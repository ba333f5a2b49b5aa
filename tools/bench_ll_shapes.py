"""the left-looking block-column updates of the chain driver at N = 8192, each alone:  [LGP_GEMM_FORCE64=1] python tools/bench_ll_shapes.py
C (rows x 1024) -= A (rows x K) B (1024 x K)^T with K = 1024 (k + 1): the products U3(k) of potrf_chain (csrc/potrf.hip)"""
import sys, ctypes
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
from lsqfitgp_amd import _lib
ctx = lgp.Context.default(); lib = _lib.load()
N = 8192
rng = np.random.default_rng(0)
L = lgp.DeviceArray.from_host(rng.standard_normal((N, N)))
e0, e1 = ctx.event(), ctx.event()
for k in range(6):
    rows, K = N - 1024 * (k + 2), 1024 * (k + 1)
    A = L.block(N - rows, N, 0, K)
    B = L.block(1024 * (k + 1), 1024 * (k + 2), 0, K)
    C = L.block(N - rows, N, 1024 * (k + 1), 1024 * (k + 2))
    best = 1e9
    for rep in range(5):
        ctx.record(e0)
        _lib.check(lib.lgp_gemm_f64(ctx.handle, 0, 1, rows, 1024, K, -1.0, A.ptr, A.ld, B.ptr, B.ld, 1.0, C.ptr, C.ld))
        ctx.record(e1)
        best = min(best, ctx.elapsed_ms(e0, e1))
    fl = 2.0 * rows * 1024 * K
    print(f'rows={rows} K={K}: {best * 1e3:.0f} us {fl / best / 1e9:.1f} TFLOP/s', flush=True)

"""one trailing update with the in-situ operand layout (panel = column slice of the N x N matrix), for PMC runs:
python tools/bench_gemm_insitu_one.py rows reps [N [pad]]   (pad: extra columns of the leading dimension)"""
import os
import sys
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
from lsqfitgp_amd import _lib
rows, reps = int(sys.argv[1]), int(sys.argv[2])
N = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
pad = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ctx = lgp.Context.default(); lib = _lib.load()
L = lgp.DeviceArray.empty((N, N + pad))
rng = np.random.default_rng(0)
blk = lgp.DeviceArray.from_host(rng.standard_normal((1024, N + pad)))
for r0 in range(0, N, 1024):
    L.rows(r0, r0 + 1024).assign(blk)
e0, e1 = ctx.event(), ctx.event()
P = L.block(N - rows, N, 0, 1024)            # rows x 1024 panel, ld = N
C = L.block(N - rows, N, N - rows, N)        # trailing matrix, ld = N
rect = int(os.environ.get('RECT', 0))          # RECT=n: the rectangular product rows x n instead (B = the first n rows of the panel)
for rep in range(reps):
    ctx.record(e0)
    if rect:
        _lib.check(lib.lgp_update_nt_f64(ctx.handle, rows, rect, 1024, P.ptr, P.ld, P.ptr, P.ld, C.ptr, C.ld, 0))
    else:
        _lib.check(lib.lgp_update_nt_f64(ctx.handle, rows, rows, 1024, P.ptr, P.ld, P.ptr, P.ld, C.ptr, C.ld, 1))
    ctx.record(e1)
    ms = ctx.elapsed_ms(e0, e1)
    print(f'rows={rows} ld={P.ld}: {ms:.3f} ms {rows * (rows + 1) * 1024 / ms / 1e9:.1f} TFLOP/s', flush=True)

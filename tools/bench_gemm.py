"""micro-benchmark of the MFMA gemm: python tools/bench_gemm.py"""
import sys, ctypes
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
from lsqfitgp_amd import _lib
ctx = lgp.Context.default()
lib = _lib.load()
def bench(m, n, k, lower, reps=5, ta=0, tb=1):
    A = lgp.DeviceArray.empty((k, m) if ta else (m, k)); B = lgp.DeviceArray.empty((n, k) if tb else (k, n)); C = lgp.DeviceArray.zeros((m, n))
    A.fill_zero(); B.fill_zero()
    # random-ish content: upload a small random block pattern
    rng = np.random.default_rng(0)
    blk = rng.standard_normal((min(A.shape[0], 1024), A.shape[1]))
    for r0 in range(0, A.shape[0], blk.shape[0]):
        r1 = min(r0 + blk.shape[0], A.shape[0]); A.rows(r0, r1).assign(lgp.DeviceArray.from_host(blk[:r1 - r0]))
    blk = rng.standard_normal((min(B.shape[0], 1024), B.shape[1]))
    for r0 in range(0, B.shape[0], blk.shape[0]):
        r1 = min(r0 + blk.shape[0], B.shape[0]); B.rows(r0, r1).assign(lgp.DeviceArray.from_host(blk[:r1 - r0]))
    e0, e1 = ctx.event(), ctx.event()
    best = 1e9
    for r in range(reps):
        ctx.record(e0)
        if lower:
            _lib.check(lib.lgp_update_nt_f64(ctx.handle, m, n, k, A.ptr, A.ld, B.ptr, B.ld, C.ptr, C.ld, 1))
        else:
            _lib.check(lib.lgp_gemm_f64(ctx.handle, ta, tb, m, n, k, -1.0, A.ptr, A.ld, B.ptr, B.ld, 1.0, C.ptr, C.ld))
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1); best = min(best, ms)
    flops = 2.0 * m * n * k * (0.5 if lower else 1.0)
    print(f'm={m} n={n} k={k} lower={lower} ta={ta} tb={tb}: {best:.3f} ms  {flops / best / 1e9:.1f} TFLOP/s', flush=True)
import os
if os.environ.get('QUICK'):
  for r in (4096, 8192, 16384, 30720):
    bench(r, r, 1024, 1, reps=3)
  bench(16384, 16384, 2048, 1, reps=3)
  bench(16384, 16384, 4096, 1, reps=3)
  bench(30720, 1024, 1024, 0, reps=3)
  bench(30720, 128, 1024, 0, reps=3)
  sys.exit(0)
if not os.environ.get('SKINNY'):
  for r in (2048, 4096, 6144, 8192, 12288, 16384, 20480, 24576, 30720):
    bench(r, r, 1024, 1, reps=3)
for r in (4096, 16384, 30720):
    bench(r, 1024, 1024, 1, reps=3)
for r in (4096, 16384, 30720):
    bench(r, 1024, 1024, 0, reps=3)
    bench(r, 128, 1024, 0, reps=3)
    bench(r, 2048, 1024, 0, reps=3)

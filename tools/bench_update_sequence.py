"""the 18 large trailing updates of a C3 factorisation (N = 32768, nb = 1024) alone, back to back, in the factorisation's own operand
layout -- nothing else on the GPU: what the update kernel reaches per step when no look-ahead kernel shares the CUs with it.
python tools/bench_update_sequence.py [reps]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import lsqfitgp_amd as lgp
from lsqfitgp_amd import _lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N, nb = 32768, 1024
ctx = lgp.Context.default(); lib = _lib.load()
L = lgp.DeviceArray.empty((N, N))
rng = np.random.default_rng(0)
blk = lgp.DeviceArray.from_host(rng.standard_normal((1024, N)) * 1e-3)
for r0 in range(0, N, 1024):
    L.rows(r0, r0 + 1024).assign(blk)
nsteps = 18
ev = [ctx.event() for _ in range(nsteps + 1)]
for rep in range(reps):
    ctx.sync()
    ctx.record(ev[0])
    for b in range(nsteps):
        rows = N - (b + 2) * nb
        P = L.block(N - rows, N, b * nb, (b + 1) * nb)
        C = L.block(N - rows, N, N - rows, N)
        _lib.check(lib.lgp_update_nt_f64(ctx.handle, rows, rows, nb, P.ptr, P.ld, P.ptr, P.ld, C.ptr, C.ld, 1))
        ctx.record(ev[b + 1])
    ctx.sync()
    tot = 0.
    line = []
    for b in range(nsteps):
        rows = N - (b + 2) * nb
        ms = ctx.elapsed_ms(ev[b], ev[b + 1])
        tot += ms
        line.append(f'{rows * (rows + 1) * nb / ms / 1e9:.1f}')
    print(f'rep {rep}: total {tot:.2f} ms; TFLOP/s per step: ' + ' '.join(line), flush=True)

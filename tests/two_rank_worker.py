"""
One rank of the two-rank on-device test of tests/test_gpu_dist.py (run as a script, one process per rank, both on
the single GPU of the box): DistributedChol + posterior with panels exchanged through the host-callback transport
(standard-library sockets, lsqfitgp_amd/_rendezvous.py), results written to <tmpdir>/rank<r>.npz.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(rank, world, port, n, nb, m, tmpdir):
    import os
    from lsqfitgp_amd._rendezvous import Rendezvous
    dist = Rendezvous(rank, world, port=port)
    import lsqfitgp_amd as lgp
    from lsqfitgp_amd._dist import DistributedChol
    ctx = lgp.Context(0)                      # both ranks on the one GPU of the box
    lgp.Context._default = ctx
    ctx.comm_init_host(rank, world, dist.collective, sendrecv=os.environ.get('LGP_TEST_SENDRECV', '1' if world >= 3 else '0') == '1')
    assert ctx.comm_info() == (rank, world, 'host')
    rng = np.random.default_rng(7)
    x = rng.uniform(0, n / 16, n)
    xs = rng.uniform(0, n / 16, m)
    y = np.sin(x / 4) + 0.1 * rng.standard_normal(n)
    carry = (xs, y) if os.environ.get('LGP_TEST_CARRY') == '1' else None
    amp = float(os.environ.get('LGP_TEST_AMP', '1'))          # kernel * amp: exercises the power-of-two diagonal scaling across ranks
    kern = lgp.ExpQuad() if amp == 1 else lgp.ExpQuad() * amp
    d = DistributedChol(kern, x, noise=0.01 * amp, nb=nb, ctx=ctx, carry=carry)
    mean, cov, mlnd = d.posterior(kern, x, xs, y)
    if os.environ.get('LGP_TEST_BIG') == '1':
        # large problem: instead of the factor itself, its action on fixed random vectors, summed / stacked by the parent:
        # L Z = sum over owned blocks L[:, blk] Z[blk];  (L^T Z)[blk] = L[:, blk]^T Z.  The rows below a diagonal block are
        # contracted on the device (MFMA gemms on the resident block columns), the diagonal block itself (whose strict upper
        # triangle is not part of the factor) on the host
        import ctypes
        from lsqfitgp_amd import _lib
        lib = _lib.load()
        F = ctypes.c_double
        Z = np.random.default_rng(11).standard_normal((n, 2))
        Zd = lgp.DeviceArray.from_host(Z, ctx=ctx)
        LZd = lgp.DeviceArray.zeros((n, 2), ctx=ctx)
        L = d.layout
        LZ_diag = np.zeros((n, 2))
        LtZ = {}
        diag = {}
        for b in L.owned():
            lb, w, c0 = L.local_index(b), L.width(b), L.col0(b)
            D = np.tril(d.Lloc.block(c0, c0 + w, lb * nb, lb * nb + w).to_host())
            diag[b] = np.diag(D).copy()
            LZ_diag[c0:c0 + w] += D @ Z[c0:c0 + w]
            lt = D.T @ Z[c0:c0 + w]
            below = n - (c0 + w)
            if below:
                A = d.Lloc.block(c0 + w, n, lb * nb, lb * nb + w)
                Zb = Zd.block(c0, c0 + w, 0, 2)
                out = LZd.block(c0 + w, n, 0, 2)
                _lib.check(lib.lgp_gemm_f64(ctx.handle, 0, 0, below, 2, w, F(1), A.ptr, A.ld, Zb.ptr, Zb.ld, F(1), out.ptr, out.ld), 'gemm')
                tmp = lgp.DeviceArray.empty((w, 2), ctx=ctx)
                Zbel = Zd.block(c0 + w, n, 0, 2)
                _lib.check(lib.lgp_gemm_f64(ctx.handle, 1, 0, w, 2, below, F(1), A.ptr, A.ld, Zbel.ptr, Zbel.ld, F(0), tmp.ptr, tmp.ld), 'gemm')
                lt = lt + tmp.to_host()
            LtZ[b] = lt
        LZ = LZd.to_host() + LZ_diag
        np.savez(os.path.join(tmpdir, f'rank{rank}.npz'), mean=mean, cov=cov, mlnd=mlnd, eps=d.eps, LZ=LZ, slab=d.slab,
                 owned=np.array(sorted(LtZ)), **{f'lt{b}': v for b, v in LtZ.items()}, **{f'dg{b}': v for b, v in diag.items()})
        dist.barrier()
        dist.close()
        return
    blocks = {b: blk for b, blk in d.local_factor().items()}
    np.savez(os.path.join(tmpdir, f'rank{rank}.npz'), mean=mean, cov=cov, mlnd=mlnd, eps=d.eps,
             owned=np.array(sorted(blocks)), **{f'b{b}': blk for b, blk in blocks.items()})
    dist.barrier()
    dist.close()




if __name__ == '__main__':
    a = sys.argv[1:]
    main(int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]), a[6])

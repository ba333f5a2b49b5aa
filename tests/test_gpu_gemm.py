"""
GPU tests of the MFMA gemm entry points (lgp_gemm_f64/f32, lgp_update_nt_f64) against numpy on the same seeded
inputs: every transpose combination, accumulate-in-place (alpha = +-1, beta = 1) and general alpha/beta epilogues,
64- and 128-tile paths, edge tiles, unaligned leading dimensions (general load path), split-K shapes, k = 0,
and the lower-trapezoid update used by the factorisation (reference: the `@` products of
src/lsqfitgp/_linalg/_decomp.py:409,420,431,435 and the trailing update inside jlinalg.cholesky, :388).
"""

import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lgp():
    import lsqfitgp_amd
    lsqfitgp_amd.Context.default()
    return lsqfitgp_amd


def dev_view(lgp, a, pad):
    """ device copy of `a` embedded in a wider allocation: leading dimension = cols + pad, offset 1 element when pad is odd
    (16-byte misaligned rows -> the kernel's general load path) """
    rows, cols = a.shape
    left = 1 if pad % 2 else 0
    wide = np.zeros((rows, cols + pad), a.dtype)
    wide[:, left:left + cols] = a
    return lgp.DeviceArray.from_host(wide).block(0, rows, left, left + cols)


def run_gemm(lgp, ta, tb, alpha, A, B, beta, C, pad=0):
    from lsqfitgp_amd import _lib
    lib = _lib.load()
    ctx = lgp.Context.default()
    f, F = _lib.fn('lgp_gemm', A.dtype)
    m = A.shape[1] if ta else A.shape[0]
    k = A.shape[0] if ta else A.shape[1]
    n = B.shape[0] if tb else B.shape[1]
    Ad, Bd, Cd = dev_view(lgp, A, pad), dev_view(lgp, B, pad), dev_view(lgp, C, pad)
    _lib.check(f(ctx.handle, int(ta), int(tb), m, n, k, F(alpha), Ad.ptr, Ad.ld, Bd.ptr, Bd.ld, F(beta), Cd.ptr, Cd.ld))
    return Cd.to_host()


SHAPES = [(1, 1, 1), (16, 16, 4), (64, 64, 16), (128, 128, 16), (129, 127, 33), (200, 72, 130), (257, 300, 64), (1024, 101, 256),
          (640, 640, 1024), (96, 1100, 48), (2100, 2050, 40)]


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('m,n,k', SHAPES)
def test_gemm_shapes(lgp, rng, dtype, ta, tb, m, n, k):
    tol = 1e-12 if dtype == np.float64 else 2e-5
    for alpha, beta, pad in [(1., 0., 0), (-1., 1., 0), (1., 1., 4), (0.5, -2., 3), (-1., 1., 1)]:
        A = rng.standard_normal((k, m) if ta else (m, k)).astype(dtype)
        B = rng.standard_normal((n, k) if tb else (k, n)).astype(dtype)
        C = rng.standard_normal((m, n)).astype(dtype)
        opA = A.T if ta else A
        opB = B.T if tb else B
        want = alpha * (opA.astype(float) @ opB.astype(float)) + beta * C.astype(float)
        got = run_gemm(lgp, ta, tb, alpha, A, B, beta, C, pad)
        scale = np.abs(opA.astype(float)) @ np.abs(opB.astype(float)) * abs(alpha) + abs(beta) * np.abs(C)
        assert np.all(np.abs(got - want) <= tol * np.maximum(scale, 1e-300) * max(k, 1) ** 0.5 + 0), (alpha, beta, pad)


def test_gemm_beta_zero_ignores_nan_in_c(lgp, rng):
    A = rng.standard_normal((70, 40)); B = rng.standard_normal((40, 90))
    C = np.full((70, 90), np.nan)
    got = run_gemm(lgp, 0, 0, 1., A, B, 0., C)
    np.testing.assert_allclose(got, A @ B, rtol=1e-12, atol=1e-12)


def test_gemm_k_zero(lgp, rng):
    from lsqfitgp_amd import _lib
    lib = _lib.load(); ctx = lgp.Context.default()
    C = rng.standard_normal((33, 45))
    Cd = lgp.DeviceArray.from_host(C)
    dummy = lgp.DeviceArray.zeros((1, 1))
    _lib.check(lib.lgp_gemm_f64(ctx.handle, 0, 0, 33, 45, 0, 1.0, dummy.ptr, 1, dummy.ptr, 45, 2.0, Cd.ptr, Cd.ld))
    np.testing.assert_allclose(Cd.to_host(), 2 * C, rtol=1e-15)


def test_gemm_split_k(lgp, rng):
    """ few tiles, deep K: the deterministic split-K slab path (X^T X over N rows) """
    n, m = 20000, 101
    X = rng.standard_normal((n, m))
    got = run_gemm(lgp, 1, 0, 1., X, X, 0., np.zeros((m, m)))
    np.testing.assert_allclose(got, X.T @ X, rtol=1e-11, atol=1e-9)
    again = run_gemm(lgp, 1, 0, 1., X, X, 0., np.zeros((m, m)))
    np.testing.assert_array_equal(got, again)          # deterministic reduction order


@pytest.mark.parametrize('rows,cols,k', [(300, 300, 128), (1000, 260, 128), (2304, 2304, 1024), (2500, 1024, 1024), (4000, 4000, 200), (130, 5, 7)])
def test_update_nt_lower(lgp, rng, rows, cols, k):
    """ C (rows x cols, C(0,0) on the diagonal) -= A B^T on and below the diagonal only; the strict upper part is untouched """
    from lsqfitgp_amd import _lib
    lib = _lib.load(); ctx = lgp.Context.default()
    A = rng.standard_normal((rows, k)); B = rng.standard_normal((cols, k)); C = rng.standard_normal((rows, cols))
    Ad, Bd, Cd = (lgp.DeviceArray.from_host(v) for v in (A, B, C))
    _lib.check(lib.lgp_update_nt_f64(ctx.handle, rows, cols, k, Ad.ptr, Ad.ld, Bd.ptr, Bd.ld, Cd.ptr, Cd.ld, 1))
    got = Cd.to_host()
    full = C - A @ B.T
    mask = np.tril(np.ones((rows, cols), bool))
    np.testing.assert_allclose(got[mask], full[mask], rtol=1e-11, atol=1e-10)
    np.testing.assert_array_equal(got[~mask], C[~mask])
    # general (non-lower) form of the same entry point
    Cd2 = lgp.DeviceArray.from_host(C)
    _lib.check(lib.lgp_update_nt_f64(ctx.handle, rows, cols, k, Ad.ptr, Ad.ld, Bd.ptr, Bd.ld, Cd2.ptr, Cd2.ld, 0))
    np.testing.assert_allclose(Cd2.to_host(), full, rtol=1e-11, atol=1e-10)


def test_gemm_argument_errors(lgp):
    from lsqfitgp_amd import _lib
    lib = _lib.load(); ctx = lgp.Context.default()
    a = lgp.DeviceArray.zeros((8, 8))
    assert lib.lgp_gemm_f64(ctx.handle, 0, 0, 8, 8, 8, 1.0, a.ptr, 4, a.ptr, 8, 0.0, a.ptr, 8) < 0     # lda < k
    assert lib.lgp_gemm_f64(ctx.handle, 0, 0, 8, 8, 8, 1.0, a.ptr, 8, a.ptr, 8, 0.0, a.ptr, 4) < 0     # ldc < n
    assert lib.lgp_gemm_f64(ctx.handle, 0, 0, 8, 8, 8, 1.0, None, 8, a.ptr, 8, 0.0, a.ptr, 8) < 0      # null operand
    assert lib.lgp_gemm_f64(None, 0, 0, 8, 8, 8, 1.0, a.ptr, 8, a.ptr, 8, 0.0, a.ptr, 8) < 0           # null context


@pytest.mark.parametrize('n,nb,G,rank,q0,k', [(1500, 128, 2, 1, 0, 128), (2600, 256, 3, 0, 1, 256), (4000, 1024, 2, 0, 0, 1024), (3000, 128, 4, 3, 2, 130),
                                              (5000, 256, 1, 0, 3, 256)])
def test_update_nt_cyclic(lgp, rng, n, nb, G, rank, q0, k):
    """ lgp_update_nt_cyclic_f64: one launch updates every block column a rank owns (from local block q0 on) in a
    block-column-cyclic layout; against a per-block numpy restatement.  Local block q = global block q * G + rank. """
    from lsqfitgp_amd import _lib
    lib = _lib.load(); ctx = lgp.Context.default()
    nblocks = -(-n // nb)
    owned = [b for b in range(nblocks) if b % G == rank]
    widths = [min(nb, n - b * nb) for b in owned]
    if q0 >= len(owned):
        pytest.skip('no block to update')
    g0 = owned[q0] * nb                                    # global row/col of C(0, 0)
    rows = n - g0
    ncols = sum(widths[q0:])
    P = rng.standard_normal((rows, k))                     # panel rows g0 .. n-1
    Cloc = rng.standard_normal((rows, ncols))
    want = Cloc.copy()
    col = 0
    for q in range(q0, len(owned)):
        c0 = owned[q] * nb - g0                            # row (relative to g0) of this block's diagonal
        w = widths[q]
        upd = P[c0:] @ P[c0:c0 + w].T                      # rows c0.., lower trapezoid of the block column
        mask = np.tril(np.ones((rows - c0, w), bool))
        want[c0:, col:col + w][mask] -= upd[mask]
        col += w
    Pd, Cd = lgp.DeviceArray.from_host(P), lgp.DeviceArray.from_host(Cloc)
    _lib.check(lib.lgp_update_nt_cyclic_f64(ctx.handle, rows, ncols, k, Pd.ptr, Pd.ld, Cd.ptr, Cd.ld, nb, G))
    got = Cd.to_host()
    np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-10)
    # everything above the staircase is untouched, bit for bit
    col = 0
    for q in range(q0, len(owned)):
        c0 = owned[q] * nb - g0
        w = widths[q]
        np.testing.assert_array_equal(got[:c0, col:col + w], Cloc[:c0, col:col + w])
        col += w


@pytest.mark.parametrize('n,nb,G,rank,q0,k,slab', [(3000, 128, 2, 1, 0, 128, 512), (5200, 256, 3, 0, 1, 256, 1024), (6000, 1024, 2, 0, 0, 1024, 2048),
                                                   (3000, 128, 4, 3, 2, 130, 384), (5000, 256, 1, 0, 3, 256, 768)])
def test_update_nt_slab(lgp, rng, n, nb, G, rank, q0, k, slab):
    """ lgp_update_nt_slab_f64: the staircase update of `test_update_nt_cyclic` produced slab by slab (rows [roff, roff + rows)
    below the first block column's diagonal per launch, the lower mask shifted by the row offset) -- how the sharded
    factorisation consumes the pieces of a panel as they arrive (csrc/dist.hip).  The slabs together must give exactly the
    one-launch result, and each launch must leave every other row untouched. """
    from lsqfitgp_amd import _lib
    lib = _lib.load(); ctx = lgp.Context.default()
    nblocks = -(-n // nb)
    owned = [b for b in range(nblocks) if b % G == rank]
    widths = [min(nb, n - b * nb) for b in owned]
    if q0 >= len(owned):
        pytest.skip('no block to update')
    g0 = owned[q0] * nb
    rows = n - g0
    ncols = sum(widths[q0:])
    P = rng.standard_normal((rows, k))
    Cloc = rng.standard_normal((rows, ncols))
    Pd, Cd, Cs = lgp.DeviceArray.from_host(P), lgp.DeviceArray.from_host(Cloc), lgp.DeviceArray.from_host(Cloc)
    _lib.check(lib.lgp_update_nt_cyclic_f64(ctx.handle, rows, ncols, k, Pd.ptr, Pd.ld, Cd.ptr, Cd.ld, nb, G))
    one = Cd.to_host()
    prev = Cloc
    for r0 in range(0, rows, slab):
        r1 = min(r0 + slab, rows)
        _lib.check(lib.lgp_update_nt_slab_f64(ctx.handle, r1 - r0, ncols, k, Pd.ptr + r0 * Pd.ld * 8, Pd.ld, Pd.ptr, Pd.ld,
                                              Cs.ptr + r0 * Cs.ld * 8, Cs.ld, nb, G, r0))
        now = Cs.to_host()
        np.testing.assert_array_equal(now[:r0], prev[:r0])
        np.testing.assert_array_equal(now[r1:], prev[r1:])
        prev = now
    np.testing.assert_array_equal(prev, one)

"""
GPU parity tests of the decomposition (C-ABI: lgp_chol_factor, lgp_trsm, lgp_gemm,
lgp_colsumsq, lgp_mlnd_value ...) through the host mirror ``lsqfitgp_amd.Chol``:
the reference's identity tests (tests/linalg/test_decomp.py) and direct
comparison with the CPU oracle on the same seeded inputs.
"""

import ctypes

import numpy as np
import pytest

import oracle
from scipy import linalg
from . import util
from .test_oracle_decomp import DecompTestBase

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lgp():
    import lsqfitgp_amd
    lsqfitgp_amd.Context.default()          # fails loudly if the HIP library or the GPU is missing
    return lsqfitgp_amd


class TestDeviceChol(DecompTestBase):
    """ the reference's TestChol, run on the device implementation """

    def make(self, K, **kw):
        import lsqfitgp_amd
        return lsqfitgp_amd.Chol(K, **kw)

    def test_isinstance(self, decomp):
        import lsqfitgp_amd
        assert isinstance(decomp, lsqfitgp_amd.Decomposition)


    # the reference's derivative tests as it writes them, through `make_derivs` (tests/linalg/test_decomp.py:227-261); jax.jacrev /
    # jacfwd of the likelihood are replaced by central differences of it, `_linalg.solve_batched` is the device one
    def test_make_derivs_gradrev(self, likelihood, s, decomp, r, r_factory, K_factory):
        _, _, kw = decomp.make_derivs(K_factory, r_factory, s, gradrev=True)
        _, result, _, _, _ = decomp.minus_log_normal_density(r, gradrev=True, **kw)
        util.assert_close_matrices(np.atleast_1d(result), np.atleast_1d(self._fd(likelihood, s)), rtol=1e-5, atol=1e-7)

    def test_make_derivs_gradfwd(self, likelihood, s, decomp, r, r_factory, K_factory):
        _, _, kw = decomp.make_derivs(K_factory, r_factory, s, gradfwd=True)
        _, _, result, _, _ = decomp.minus_log_normal_density(r, gradfwd=True, **kw)
        util.assert_close_matrices(np.atleast_1d(result), np.atleast_1d(self._fd(likelihood, s)), rtol=1e-5, atol=1e-7)

    def test_make_derivs_fisher_and_fishvec(self, s, decomp, r, K, K_factory, r_factory, rng):
        import lsqfitgp_amd
        Kout, rout, kw = decomp.make_derivs(K_factory, r_factory, s, fisher=True)
        np.testing.assert_array_equal(Kout, K)
        np.testing.assert_array_equal(rout, r)
        assert kw['dK'].shape == K.shape + s.shape and kw['dr'].shape == r.shape + s.shape
        _, _, _, result, _ = decomp.minus_log_normal_density(r, fisher=True, **kw)
        dK, dr = kw['dK'], kw['dr']
        invK_dK = lsqfitgp_amd._linalg.solve_batched(K, np.moveaxis(dK, 2, 0), assume_a='pos')
        expected = 1/2 * np.einsum('kij,qji->kq', invK_dK, invK_dK) + dr.T @ linalg.solve(K, dr, assume_a='pos')
        util.assert_close_matrices(np.atleast_2d(result), expected, rtol=1e-9)
        vec = rng.standard_normal(s.shape)
        _, _, kw = decomp.make_derivs(K_factory, r_factory, s, fishvec=True, vec=vec)
        assert sorted(kw) == ['dK_jvp_vec', 'dK_vjp', 'dr_jvp_vec', 'dr_vjp']
        _, _, _, _, fv = decomp.minus_log_normal_density(r, fishvec=True, **kw)
        util.assert_close_matrices(np.atleast_1d(fv), expected @ vec, rtol=1e-9)


def randpd(rng, n, dtype=np.float64, cond_shift=0.5):
    Q = rng.standard_normal((n, n))
    return (Q @ Q.T / n + np.eye(n) * cond_shift).astype(dtype)


@pytest.mark.parametrize('n', [1, 2, 3, 17, 127, 128, 129, 255, 256, 257, 511, 513, 777, 1024, 1025, 1500, 2049])
def test_factor_vs_oracle_f64(lgp, n, rng):
    """ L, eps and every product against the oracle; sizes straddle the 128 / 1024 block edges """
    K = randpd(rng, n) * rng.choice([0.01, 1.0, 37.0])      # exercises the power-of-two diagonal scaling
    d = lgp.Chol(K)
    o = oracle.Chol(K)
    L = d.factor().to_host()
    assert np.all(np.triu(L, 1) == 0)
    util.assert_close_matrices(L, o._L, rtol=1e-13)
    util.assert_allclose(d.eps, o.eps, rtol=1e-14)
    m = 9
    A = rng.standard_normal((n, m)); r = rng.standard_normal(n)
    util.assert_close_matrices(d.ginv_linear(A), o.ginv_linear(A), rtol=1e-11)
    util.assert_close_matrices(d.ginv_linear(r), o.ginv_linear(r), rtol=1e-11)
    util.assert_close_matrices(d.pinv_bilinear(A, r), o.pinv_bilinear(A, r), rtol=1e-11)
    util.assert_close_matrices(d.ginv_quad(A), o.ginv_quad(A), rtol=1e-11)
    util.assert_close_matrices(d.ginv_diagquad(A), o.ginv_diagquad(A), rtol=1e-11)
    util.assert_close_matrices(d.correlate(A), o.correlate(A), rtol=1e-13)
    util.assert_close_matrices(d.back_correlate(A), o.back_correlate(A), rtol=1e-13)
    util.assert_close_matrices(d.pinv_correlate(r), o.pinv_correlate(r), rtol=1e-11)
    v1 = d.minus_log_normal_density(r, value=True)[0]
    v2 = o.minus_log_normal_density(r, value=True)[0]
    util.assert_allclose(v1, v2, rtol=1e-12)


@pytest.mark.parametrize('n', [1, 5, 128, 300, 1025])
def test_factor_vs_oracle_f32(lgp, n, rng):
    K = randpd(rng, n, np.float32, 1.0)
    d = lgp.Chol(K)
    o = oracle.Chol(K)
    assert d.factor().dtype == np.float32
    util.assert_close_matrices(d.factor().to_host(), o._L, rtol=2e-5)
    util.assert_allclose(d.eps, o.eps, rtol=1e-5)
    A = rng.standard_normal((n, 4)).astype(np.float32)
    util.assert_close_matrices(d.ginv_quad(A), o.ginv_quad(A), rtol=1e-3)


@pytest.mark.parametrize('epsrel,epsabs', [('auto', 0), (0, 'auto'), (1e-3, 1e-5), (0, 0)])
def test_eps_arguments(lgp, epsrel, epsabs, rng):
    """ ``_parseeps`` (``_decomp.py:245-255``) """
    K = randpd(rng, 50) * 5
    d = lgp.Chol(K, epsrel=epsrel, epsabs=epsabs)
    o = oracle.Chol(K, epsrel=epsrel, epsabs=epsabs)
    util.assert_allclose(d.eps, o.eps, rtol=1e-14)
    util.assert_close_matrices(d.factor().to_host(), o._L, rtol=1e-13)


def test_not_posdef_raises(lgp, rng):
    """ ``_decomp.py:389-391`` """
    for K in (-np.eye(5), np.array([[1., 2.], [2., 1.]]), np.diag([1., np.nan, 1.])):
        with pytest.raises(np.linalg.LinAlgError, match='cholesky decomposition not finite'):
            lgp.Chol(K)
    Q = randpd(rng, 300)
    Q[200, 200] = -5.
    with pytest.raises(np.linalg.LinAlgError):
        lgp.Chol(Q)
    with pytest.raises(ValueError):
        lgp.Chol(np.ones((3, 4)))


def test_zero_diagonal_scale(lgp):
    """ s = 1 where K_ii == 0 (``_decomp.py:360``) """
    K = np.diag([4., 0., 9.]) + 1e-300
    K = np.diag([4., 0., 9.])
    d = lgp.Chol(K, epsabs=1e-3)
    o = oracle.Chol(K, epsabs=1e-3)
    util.assert_allclose(d.factor().to_host(), o._L, rtol=1e-15)
    util.assert_allclose(d.eps, o.eps, rtol=1e-15)


def test_near_singular_expquad(lgp):
    """ the jitter rule on a numerically singular Gram matrix: must factor (LAPACK does) and L L' ~ K + eps """
    rng = np.random.default_rng(7)
    x = rng.uniform(-5, 5, 700)
    K = np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2)
    d = lgp.Chol(K)
    o = oracle.Chol(K)
    util.assert_allclose(d.eps, o.eps, rtol=1e-13)
    L = d.factor().to_host()
    resid = L @ L.T - K - d.eps * np.eye(700)
    assert np.linalg.norm(resid, 2) <= 50 * 700 * np.finfo(float).eps * np.linalg.norm(K, 2)
    # the factors agree to the conditioning of the problem, the reconstructed matrices to rounding
    util.assert_close_matrices(L @ L.T, o._L @ o._L.T, rtol=1e-13)


def test_device_operands(lgp, rng):
    """ DeviceArray inputs are used in place; device=True keeps results in HBM """
    n = 200
    K = randpd(rng, n)
    Kd = lgp.DeviceArray.from_host(K)
    d = lgp.Chol(Kd)
    assert d.matrix() is Kd
    A = rng.standard_normal((n, 5))
    Ad = lgp.DeviceArray.from_host(A)
    out = d.ginv_quad(Ad, device=True)
    assert isinstance(out, lgp.DeviceArray)
    util.assert_close_matrices(out.to_host(), oracle.Chol(K).ginv_quad(A), rtol=1e-11)
    np.testing.assert_array_equal(Ad.to_host(), A)      # operand untouched
    sub = lgp.DeviceArray.from_host(np.pad(A, ((0, 0), (2, 3)))).block(0, n, 2, 7)   # strided view
    util.assert_close_matrices(d.ginv_quad(sub), oracle.Chol(K).ginv_quad(A), rtol=1e-11)


def test_convert_dtype(lgp, rng):
    """ lgp_convert_f32_f64 / f64_f32 on strided views: bit-exact against numpy's casts """
    from lsqfitgp_amd import _lib
    lib = _lib.load()
    ctx = lgp.Context.default()
    a = rng.standard_normal((37, 300))
    a[3, 5], a[0, 0], a[36, 299] = np.inf, 1e-50, 3e38
    src = lgp.DeviceArray.from_host(np.pad(a, ((0, 0), (1, 2)))).block(0, 37, 1, 301)          # ld = 303
    dst = lgp.DeviceArray.from_host(np.zeros((37, 310), np.float32)).block(0, 37, 4, 304)
    _lib.check(lib.lgp_convert_f64_f32(ctx.handle, 37, 300, dst.ptr, dst.ld, src.ptr, src.ld))
    with np.errstate(over='ignore'):
        np.testing.assert_array_equal(dst.to_host(), a.astype(np.float32))
    back = lgp.DeviceArray.empty((37, 300))
    _lib.check(lib.lgp_convert_f32_f64(ctx.handle, 37, 300, back.ptr, back.ld, dst.ptr, dst.ld))
    with np.errstate(over='ignore'):
        np.testing.assert_array_equal(back.to_host(), a.astype(np.float32).astype(np.float64))
    assert lib.lgp_convert_f32_f64(ctx.handle, 2, 8, back.ptr, 4, dst.ptr, dst.ld) < 0      # ld < cols is refused


def test_full_size_properties(lgp):
    """ N = 8192 (config C2 size): size-independent properties instead of an O(N^3) CPU recomputation:
    (L L') v == (K + eps) v on random vectors, K^-1 (K v) == v, log-det against the diagonal """
    n = 8192
    rng = np.random.default_rng(2)
    x = rng.uniform(0, n / 16, n)
    K = lgp.ExpQuad().gram(x, x)
    lib = __import__('lsqfitgp_amd')._lib.load()
    ctx = lgp.Context.default()
    assert lib.lgp_add_diag_f64(ctx.handle, n, K.ptr, K.ld, None, ctypes.c_double(0.01)) == 0
    d = lgp.Chol(K)
    V = rng.standard_normal((n, 3))
    KV = np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2) @ V + (0.01 + d.eps) * V
    LLV = d.correlate(d.back_correlate(V))
    util.assert_close_matrices(LLV, KV, rtol=1e-12)
    util.assert_close_matrices(d.ginv_linear(KV), V, rtol=1e-9)
    Ldiag = np.diag(d.factor().to_host())
    val = d.minus_log_normal_density(KV[:, 0], value=True)[0]
    expected = 0.5 * (n * np.log(2 * np.pi) + 2 * np.sum(np.log(Ldiag)) + KV[:, 0] @ V[:, 0])
    util.assert_allclose(val, expected, rtol=1e-10)


def test_full_size_c3_properties(lgp):
    """ BASELINE configuration C3 at full size (Matern-5/2, 3-D, N = 32768) through size-independent identities:
    L L^T V = (K + eps I) V with K V formed by an independent device product, K^-1 (K V) = V, and the
    log-density against its definition from diag(L) and v^T K v """
    from lsqfitgp_amd import _lib
    from lsqfitgp_amd._GP import _gemm
    n = 32768
    rng = np.random.default_rng(3)
    pts = np.empty(n, [('f0', float), ('f1', float), ('f2', float)])
    for f in pts.dtype.names:
        pts[f] = rng.uniform(0, 16, n)
    K = lgp.Maternp(p=2, scale=1.).gram(pts, pts)
    lib = _lib.load()
    ctx = lgp.Context.default()
    assert lib.lgp_add_diag_f64(ctx.handle, n, K.ptr, K.ld, None, ctypes.c_double(0.01)) == 0
    d = lgp.Chol(K)
    assert 0 < d.eps < 1e-6
    V = rng.standard_normal((n, 3))
    KV = _gemm(False, False, K, lgp.DeviceArray.from_host(V)).to_host() + d.eps * V
    LLV = d.correlate(d.back_correlate(V))
    util.assert_close_matrices(LLV, KV, rtol=1e-12)
    util.assert_close_matrices(d.ginv_linear(KV), V, rtol=1e-8)
    Lf = d.factor()
    val = d.minus_log_normal_density(KV[:, 0], value=True)[0]
    # the log-determinant part of the density against its definition from the diagonal of the factor
    quad = KV[:, 0] @ V[:, 0]
    logdet = 2 * val - n * np.log(2 * np.pi) - quad
    full_diag = np.empty(n)
    for r0 in range(0, n, 4096):          # the diagonal in 8 strips of 4096 x 4096 blocks
        full_diag[r0:r0 + 4096] = np.diag(Lf.block(r0, r0 + 4096, r0, r0 + 4096).to_host())
    assert np.all(full_diag > 0)
    util.assert_allclose(logdet, 2 * np.sum(np.log(full_diag)), rtol=1e-10)
    # strict upper triangle of the factor is zero (first strip)
    blk = Lf.block(0, 2048, 0, 4096).to_host()
    assert np.all(np.triu(blk, 1) == 0)


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('n,q', [(130, 1), (300, 5), (1024, 101), (1500, 130), (2600, 101), (3333, 7)])
def test_carried_rhs(lgp, rng, dtype, n, q):
    """ Chol(K, rhs=B): the forward solve carried through the factorisation (lgp_chol_factor_rhs) gives the same
    factor, bit for bit, as Chol(K) and L^-1 B as good as the separate substitution (reference semantics:
    pinv_correlate, _linalg/_decomp.py:437-439) """
    K = randpd(rng, n)
    d = np.exp(rng.uniform(-3, 3, n))                  # non-trivial diagonal: the power-of-two scaling path
    K = (K * d[:, None] * d[None, :]).astype(dtype)
    B = rng.standard_normal((n, q)).astype(dtype)
    plain = lgp.Chol(K)
    carried = lgp.Chol(K, rhs=B)
    np.testing.assert_array_equal(carried.factor().to_host(), plain.factor().to_host())
    assert carried.eps == plain.eps
    want = oracle.Chol(K.astype(float)).pinv_correlate(B.astype(float))
    got = carried.solved()
    sep = plain.pinv_correlate(B)
    # float32: the automatic jitter n * eps32 * (Gershgorin bound) of the float32 decomposition alone moves the
    # solution by ~1e-2 relative to the float64 oracle; carried vs separate (same matrix) differ by rounding only
    tol_oracle, tol_same = (1e-9, 1e-10) if dtype == np.float64 else (6e-2, 1e-2)
    scale = np.linalg.norm(want, axis=0)
    assert np.all(np.linalg.norm(got - want, axis=0) <= tol_oracle * scale)
    assert np.all(np.linalg.norm(got - sep, axis=0) <= tol_same * scale)
    # the other methods work on the carried factor as on the plain one
    util.assert_close_matrices(carried.ginv_quad(B[:, :1]), plain.ginv_quad(B[:, :1]), rtol=1e-9 if dtype == np.float64 else 1e-2)
    with pytest.raises(ValueError):
        plain.solved()
    with pytest.raises(ValueError):
        lgp.Chol(K, rhs=B[:-1])


def test_odd_size_many_panels_with_carried_rhs(lgp):
    """ N = 9001 (9 outer panels, ragged last panel and edge tiles everywhere) with 37 carried right-hand sides:
    factorisation identities instead of the oracle (O(N^3) on the CPU) """
    from lsqfitgp_amd._GP import _gemm
    n, q = 9001, 37
    rng = np.random.default_rng(9)
    x = rng.uniform(0, n / 16, n)
    K = lgp.ExpQuad().gram(x, x)
    lib = __import__('lsqfitgp_amd')._lib.load()
    ctx = lgp.Context.default()
    assert lib.lgp_add_diag_f64(ctx.handle, n, K.ptr, K.ld, None, ctypes.c_double(0.02)) == 0
    B = rng.standard_normal((n, q))
    d = lgp.Chol(K, rhs=B)
    plain = lgp.Chol(K)
    np.testing.assert_array_equal(d.factor().to_host(), plain.factor().to_host())
    W = d.solved()                                          # L^-1 B
    # L W = B and the separate substitution agree
    util.assert_close_matrices(d.correlate(W), B, rtol=1e-11)
    util.assert_close_matrices(W, plain.pinv_correlate(B), rtol=1e-10)
    # L L^T V = (K + eps) V with K V from an independent product
    V = rng.standard_normal((n, 2))
    KV = _gemm(False, False, K, lgp.DeviceArray.from_host(V)).to_host() + d.eps * V
    util.assert_close_matrices(d.correlate(d.back_correlate(V)), KV, rtol=1e-12)
    util.assert_close_matrices(d.ginv_linear(KV), V, rtol=1e-9)


def test_full_size_c4_properties(lgp):
    """ BASELINE configuration C4 at full size on one GPU (ExpQuad 1-D fp64, N = 65536, seed 4, x = U(0, 4096),
    noise 0.01; K and L are 32 GiB each): the identities of test_full_size_c3_properties -- L L^T V = (K + eps I) V with
    K V from an independent device product, K^-1 (K V) = V, the log-density against its definition from diag(L) -- and
    the posterior/likelihood of the carried solve (lgp_chol_factor_rhs) against the separate solve (lgp_posterior) """
    from lsqfitgp_amd import _lib
    from lsqfitgp_amd._GP import _gemm
    n, m = 65536, 100
    rng = np.random.default_rng(4)
    x = rng.uniform(0, 4096, n)
    xs = rng.uniform(0, 4096, m)
    y = np.sin(x / 4) + 0.1 * rng.standard_normal(n)
    kern = lgp.ExpQuad()
    K = kern.gram(x, x)
    lib = _lib.load()
    ctx = lgp.Context.default()
    assert lib.lgp_add_diag_f64(ctx.handle, n, K.ptr, K.ld, None, ctypes.c_double(0.01)) == 0
    Kxs = kern.gram(x, xs)
    W = lgp.DeviceArray.empty((n, m + 1))
    W.block(0, n, 0, m).assign(Kxs)
    W.block(0, n, m, m + 1).assign(lgp.DeviceArray.from_host(y.reshape(n, 1)))
    d = lgp.Chol(K, rhs=W)                       # factor with [Kxxs | y] carried through it
    assert 0 < d.eps < 1e-5
    V = rng.standard_normal((n, 3))
    KV = _gemm(False, False, K, lgp.DeviceArray.from_host(V)).to_host() + d.eps * V
    LLV = d.correlate(d.back_correlate(V))
    util.assert_close_matrices(LLV, KV, rtol=1e-12)
    util.assert_close_matrices(d.ginv_linear(KV), V, rtol=1e-8)
    Lf = d.factor()
    val = d.minus_log_normal_density(KV[:, 0], value=True)[0]
    quad = KV[:, 0] @ V[:, 0]
    logdet = 2 * val - n * np.log(2 * np.pi) - quad
    full_diag = np.empty(n)
    for r0 in range(0, n, 4096):
        full_diag[r0:r0 + 4096] = np.diag(Lf.block(r0, r0 + 4096, r0, r0 + 4096).to_host())
    assert np.all(full_diag > 0)
    util.assert_allclose(logdet, 2 * np.sum(np.log(full_diag)), rtol=1e-10)
    assert np.all(np.triu(Lf.block(0, 2048, 0, 4096).to_host(), 1) == 0)
    # posterior and likelihood: carried solve vs the separate forward solve on the same factor
    S = d.solved()                                # L^-1 [Kxxs | y], (n, m + 1)
    mean_c = S[:, :m].T @ S[:, m]
    Kss = np.exp(-0.5 * (xs[:, None] - xs[None, :]) ** 2)
    cov_c = Kss - S[:, :m].T @ S[:, :m]
    mlnd_c = 0.5 * (n * np.log(2 * np.pi) + 2 * np.sum(np.log(full_diag)) + S[:, m] @ S[:, m])
    mean_s = d.pinv_bilinear(Kxs, y)
    cov_s = Kss - d.ginv_quad(Kxs)
    mlnd_s = d.minus_log_normal_density(y, value=True)[0]
    assert np.linalg.norm(mean_c - mean_s) <= 1e-10 * np.linalg.norm(mean_s)
    assert np.linalg.norm(cov_c - cov_s, 2) <= 1e-10 * np.linalg.norm(Kss, 2)
    util.assert_allclose(mlnd_c, mlnd_s, rtol=1e-12)
    # the posterior mean interpolates the smooth signal at in-sample density 16 points per lengthscale
    assert np.sqrt(np.mean((mean_s - np.sin(xs / 4)) ** 2)) < 0.05


@pytest.mark.parametrize('n', [40, 300, 1100])
def test_normal_density_derivs_on_device(lgp, n, rng):
    """ lgp_mlnd_grad (value + forward-mode gradient + Fisher matrix in one device call, _decomp.py:461-561) on
    K(theta) = a ExpQuad(scale)(x, x) + s I,  r(theta) = y - mu, with the kernel tangent from lgp_gram_isotropic_dscale:
    against the oracle's restatement of the reference formulas fed the same derivative arrays, and against central
    differences of the oracle's value """
    x = rng.uniform(0, n / 8, n)
    y = np.sin(x) + 0.3 * rng.standard_normal(n)
    a, sc, s2, mu = 1.3, 0.8, 0.05, 0.1

    def host_K(a_, sc_, s2_):
        return a_ * oracle.ExpQuad(scale=sc_)(x[:, None], x[None, :]) + s2_ * np.eye(n)

    E, dE = lgp.ExpQuad(scale=sc).gram_dscale(x, x)
    Eh, dEh = E.to_host(), dE.to_host()
    K = a * Eh + s2 * np.eye(n)
    dK = np.stack([Eh, a * dEh, np.eye(n), np.zeros((n, n))], axis=2)
    dr = np.zeros((n, 4)); dr[:, 3] = -1
    r = y - mu
    d = lgp.Chol(K)
    o = oracle.Chol(K)
    val, grad, fish = d.normal_density_derivs(r, dK, dr, fisher=True)
    vo, _, go, fo, _ = o.minus_log_normal_density(r, dK=dK, dr=dr, value=True, gradfwd=True, fisher=True)
    util.assert_allclose(val, vo, rtol=1e-11)
    util.assert_close_matrices(grad, go, rtol=1e-9)
    util.assert_close_matrices(fish, fo, rtol=1e-9)
    # the same through the reference-shaped entry point, device-resident derivative matrices
    dKdev = [lgp.DeviceArray.from_host(np.ascontiguousarray(dK[:, :, k])) for k in range(4)]
    v2, _, g2, f2, _ = d.minus_log_normal_density(r, dK=dKdev, dr=dr, value=True, gradfwd=True, fisher=True)
    assert v2 == val
    np.testing.assert_array_equal(g2, grad)
    np.testing.assert_array_equal(f2, fish)
    # only dr / only dK
    _, _, g3, f3, _ = d.minus_log_normal_density(r, dr=dr, gradfwd=True, fisher=True)
    _, _, g3o, f3o, _ = o.minus_log_normal_density(r, dr=dr, gradfwd=True, fisher=True)
    util.assert_close_matrices(g3, g3o, rtol=1e-9)
    util.assert_close_matrices(f3, f3o, rtol=1e-9, atol=1e-300)
    # central differences of the oracle's value (epsrel=0: the jitter must not move with the parameters)
    def value(a_, sc_, s2_, mu_):
        return oracle.Chol(host_K(a_, sc_, s2_), epsrel=0).minus_log_normal_density(y - mu_, value=True)[0]
    th = np.array([a, sc, s2, mu])
    d0 = lgp.Chol(K, epsrel=0)
    _, g0, _ = d0.normal_density_derivs(r, dK, dr)
    for k in range(4):
        e = np.zeros(4); e[k] = 1e-6 * max(1, abs(th[k]))
        fd = (value(*(th + e)) - value(*(th - e))) / (2 * e[k])
        assert abs(g0[k] - fd) <= 1e-5 * max(1.0, abs(fd))


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('n,d,q', [(1, 1, 0), (130, 1, 3), (700, 3, 0), (1025, 2, 101), (2600, 5, 7)])
def test_fused_gram_factor(lgp, rng, dtype, n, d, q):
    """ lgp_gram_chol_factor_rhs: Gram build + regularisation + factorisation without materialising K, against the separate
    path (lgp_gram_isotropic + add_diag + lgp_chol_factor_rhs) on the same inputs: eps to 1e-14 (the Gershgorin sums are
    added in a different, fixed order), factor and carried solve to rounding; vector and scalar noise; kernel * c (power-of-two
    diagonal scaling) """
    from lsqfitgp_amd import _lib
    lib = _lib.load()
    ctx = lgp.Context.default()
    f64 = dtype == np.float64
    suf, F = ('f64', ctypes.c_double) if f64 else ('f32', ctypes.c_float)
    x = rng.uniform(0, max(n, 16) / 16, (d, n)).astype(dtype)
    xd = lgp.DeviceArray.from_host(x)
    B = rng.standard_normal((q, n)).astype(dtype)
    for amp, noise, noisev in ((1., 0.01, None), (37., 0.5, rng.uniform(0.1, 0.9, n).astype(dtype)), (0.004, 1e-4, None)):
        kern = lgp.Maternp(p=2, scale=dtype(1.3), loc=dtype(0.2)) * dtype(amp)
        # separate path
        K = lgp.DeviceArray.empty((n, n), dtype)
        codes = (ctypes.c_int * 1)(_lib.LGP_OP_MUL); vals = (F * 1)(amp)
        _lib.check(getattr(lib, 'lgp_gram_isotropic_' + suf)(ctx.handle, _lib.LGP_KERNEL_MATERNP, 2, 0, None, d, xd.ptr, xd.ld, n, xd.ptr, xd.ld, n,
                                                            1, F(0.2), 1, F(1.3), 1, codes, vals, 1, K.ptr, K.ld))
        nv = None if noisev is None else lgp.DeviceArray.from_host(noisev)
        _lib.check(getattr(lib, 'lgp_add_diag_' + suf)(ctx.handle, n, K.ptr, K.ld, None if nv is None else nv.ptr, F(1. if nv is not None else noise)))
        nx = -(-q // 128) * 128 if q else 0
        outs = []
        for fused in (False, True):
            L = lgp.DeviceArray.zeros((n + nx, n), dtype)
            if q:
                L.block(n, n + q, 0, n).assign(lgp.DeviceArray.from_host(B))
            dblk = lgp.DeviceArray.empty((lib.lgp_chol_dblk_size(n),), dtype)
            eps = F()
            if fused:
                st = getattr(lib, 'lgp_gram_chol_factor_rhs_' + suf)(ctx.handle, _lib.LGP_KERNEL_MATERNP, 2, 0, None, d, xd.ptr, xd.ld, n, 1, F(0.2), 1, F(1.3),
                                                                      1, codes, vals, F(noise), None if nv is None else nv.ptr, F(0), F(0),
                                                                      _lib.LGP_EPS_AUTO_REL, L.ptr, L.ld, nx, dblk.ptr, ctypes.byref(eps))
            else:
                st = getattr(lib, 'lgp_chol_factor_rhs_' + suf)(ctx.handle, n, K.ptr, K.ld, F(0), F(0), _lib.LGP_EPS_AUTO_REL, L.ptr, L.ld, nx, dblk.ptr,
                                                                 ctypes.byref(eps))
            assert _lib.check(st) == 0
            outs.append((eps.value, L.to_host()))
        (e0, L0), (e1, L1) = outs
        tol = 1e-14 if f64 else 1e-6
        assert abs(e1 - e0) <= tol * abs(e0)
        assert np.all(np.triu(L1[:n], 1) == 0)
        util.assert_close_matrices(L1[:n], L0[:n], rtol=1e-12 if f64 else 2e-5)
        if q:
            util.assert_close_matrices(L1[n:n + q], L0[n:n + q], rtol=1e-9 if f64 else 2e-2)


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('n,d,q', [(1, 1, 0), (130, 1, 3), (1025, 2, 101), (2600, 3, 7), (4200, 1, 1)])
def test_fused_gram_factor_lower_only(lgp, rng, dtype, n, d, q):
    """ LGP_FACTOR_LOWER_ONLY: the fused entry point leaves the entries above the diagonal alone (no stores of zeros).  The storage
    is filled with NaN first: the factor below the diagonal, the carried solve, eps and a later pair of triangular solves are
    bit for bit those of the call without the flag -- nothing the library does reads above the diagonal outside the diagonal
    blocks -- and lgp_zero_upper writes the zeros a product with the full matrix needs (Chol._full_factor) """
    from lsqfitgp_amd import _lib
    lib = _lib.load()
    ctx = lgp.Context.default()
    f64 = dtype == np.float64
    suf, F = ('f64', ctypes.c_double) if f64 else ('f32', ctypes.c_float)
    x = rng.uniform(0, max(n, 16) / 16, (d, n)).astype(dtype)
    xd = lgp.DeviceArray.from_host(x)
    B = rng.standard_normal((q, n)).astype(dtype)
    R = rng.standard_normal((n, 3)).astype(dtype)
    nx = -(-q // 128) * 128 if q else 0
    outs = []
    for flag in (0, _lib.LGP_FACTOR_LOWER_ONLY):
        L = lgp.DeviceArray.from_host(np.full((n + nx, n), np.nan, dtype))
        if nx:
            L.block(n, n + nx, 0, n).fill_zero()
        if q:
            L.block(n, n + q, 0, n).assign(lgp.DeviceArray.from_host(B))
        dblk = lgp.DeviceArray.empty((lib.lgp_chol_dblk_size(n),), dtype)
        eps = F()
        st = getattr(lib, 'lgp_gram_chol_factor_rhs_' + suf)(ctx.handle, _lib.LGP_KERNEL_EXPQUAD, 0, 0, None, d, xd.ptr, xd.ld, n, 0, F(0), 1, F(0.7),
                                                              0, None, None, F(0.01), None, F(0), F(0), _lib.LGP_EPS_AUTO_REL | flag, L.ptr, L.ld,
                                                              nx, dblk.ptr, ctypes.byref(eps))
        assert _lib.check(st) == 0
        Rd = lgp.DeviceArray.from_host(R)
        for trans in (0, 1):
            _lib.check(getattr(lib, 'lgp_trsm_' + suf)(ctx.handle, n, L.ptr, L.ld, dblk.ptr, trans, Rd.ptr, Rd.ld, 3))
        Lh = L.to_host()
        if flag:
            _lib.check(getattr(lib, 'lgp_zero_upper_' + suf)(ctx.handle, n, L.ptr, L.ld))
            assert np.all(np.triu(L.to_host()[:n], 1) == 0)
            np.testing.assert_array_equal(np.tril(L.to_host()[:n]), np.tril(Lh[:n]))
        else:
            assert np.all(np.triu(Lh[:n], 1) == 0)
        outs.append((eps.value, Lh, Rd.to_host()))
    (e0, L0, R0), (e1, L1, R1) = outs
    assert e0 == e1
    np.testing.assert_array_equal(np.tril(L1[:n]), np.tril(L0[:n]))
    np.testing.assert_array_equal(L1[n:], L0[n:])
    np.testing.assert_array_equal(R1, R0)
    assert np.all(np.isfinite(R1))
    if n > 1024:
        assert np.isnan(L1[0, n - 1])       # (the flag did something)


@pytest.fixture(scope='module')
def driver_contexts(lgp):
    """ two contexts on the same GPU: the launch-per-operation driver of the factorisation (LGP_CHAIN=0 when the context is
    created) and the default (chain kernel up to N = 12288) """
    import os
    old = os.environ.get('LGP_CHAIN')
    os.environ['LGP_CHAIN'] = '0'
    try:
        launch = lgp.Context(0)
    finally:
        if old is None:
            del os.environ['LGP_CHAIN']
        else:
            os.environ['LGP_CHAIN'] = old
    return launch, lgp.Context(0)


@pytest.mark.parametrize('n,q', [(256, 0), (1152, 7), (2048, 0), (2432, 3)])
def test_chain_kernel_against_launch_per_operation_driver(lgp, rng, driver_contexts, n, q):
    """ the two drivers of the blocked factorisation (csrc/potrf.hip: the chain kernel whose resident workers factor the
    diagonal squares, default up to N = 12288, and the launch-per-operation driver, LGP_CHAIN=0) on the same matrix: the
    same factor to rounding (the order of the additions differs), both satisfying L L^T = K + eps, and the same carried
    solve.  LAPACK potrf in the reference (_linalg/_decomp.py:388) is the third opinion (the oracle). """
    K = randpd(rng, n) * 37.0
    B = rng.standard_normal((n, q)) if q else None
    ctx_launch, ctx_chain = driver_contexts
    a = lgp.Chol(K, ctx=ctx_launch, rhs=B)
    b = lgp.Chol(K, ctx=ctx_chain, rhs=B)
    o = oracle.Chol(K)
    La, Lb = a.factor().to_host(), b.factor().to_host()
    assert a.eps == b.eps == pytest.approx(o.eps, rel=1e-14)
    util.assert_close_matrices(La, o._L, rtol=1e-13)
    util.assert_close_matrices(Lb, o._L, rtol=1e-13)
    util.assert_close_matrices(Lb, La, rtol=1e-13)
    assert np.all(np.triu(Lb, 1) == 0)
    Keps = K + b.eps * np.eye(n)
    assert np.linalg.norm(Lb @ Lb.T - Keps) <= 1e-13 * n * np.linalg.norm(Keps)
    if q:
        util.assert_close_matrices(b.solved(), a.solved(), rtol=1e-10)
        util.assert_close_matrices(b.solved(), o.pinv_correlate(B), rtol=1e-10)


@pytest.mark.parametrize('n,q', [(9216, 5), (12288, 0)])
def test_chain_driver_left_looking_above_8192(lgp, rng, driver_contexts, n, q):
    """ the sizes the chain driver took over in round 5 with LEFT-looking updates (potrf_chain, LGP_CHAIN_LEFT_MAX_N = 12288; right-looking
    above 8192 before): the same factor and carried solve as the launch-per-operation driver to rounding, repeatably (two runs bit for
    bit), and a solve that satisfies K x = b to the accuracy the conditioning allows """
    x = np.sort(rng.uniform(0, n / 16, n))
    K = np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2) + 0.01 * np.eye(n)
    B = rng.standard_normal((n, q)) if q else None
    ctx_launch, ctx_chain = driver_contexts
    a = lgp.Chol(K, ctx=ctx_launch, rhs=B)
    b = lgp.Chol(K, ctx=ctx_chain, rhs=B)
    c = lgp.Chol(K, ctx=ctx_chain, rhs=B)
    La, Lb, Lc = a.factor().to_host(), b.factor().to_host(), c.factor().to_host()
    assert a.eps == b.eps
    np.testing.assert_array_equal(Lb, Lc)
    assert np.max(np.abs(Lb - La)) <= 1e-12 * np.max(np.abs(La))
    assert np.all(np.triu(Lb, 1) == 0)
    if q:
        util.assert_close_matrices(b.solved(), a.solved(), rtol=1e-9)
    r = rng.standard_normal(n)
    sol = b.ginv_linear(r)
    res = K @ sol + b.eps * sol - r
    assert np.linalg.norm(res) <= 1e-9 * np.linalg.norm(r)


def test_chain_kernel_reports_a_bad_pivot(lgp, rng):
    """ a matrix that is not positive definite: the worker that meets the pivot records it and the factorisation goes on to
    its end (no worker is left waiting); the host raises the reference's error (_linalg/_decomp.py:389-391) """
    n = 1536
    K = randpd(rng, n)
    K[700, 700] = -1.0
    with pytest.raises(np.linalg.LinAlgError):
        lgp.Chol(K, epsrel=0)
    lgp.Chol(randpd(rng, n))          # and the context is usable afterwards


def _context_with_env(lgp, **env):
    import os
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return lgp.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def test_chain_kernel_stall_is_reported_not_hung(lgp, rng):
    """ a worker of the chain kernel that never gets an input (here: the driver withholds the flag word u1 of the second
    panel, LGP_CHAIN_TEST_STALL) gives up after the time-out, every stream-side wait on the kernel's flag words (ready[] and
    flagP[]) is released, and the call returns an error instead of hanging; the context then switches to the
    launch-per-operation driver and stays usable """
    import time
    ctx = _context_with_env(lgp, LGP_CHAIN_TEST_STALL=1, LGP_CHAIN_TIMEOUT_MS=300)
    n = 2304                     # three panels: far rows of panels 0 and 1 wait on flagP / ready words of squares that never run
    K = randpd(rng, n)
    t0 = time.perf_counter()
    with pytest.raises(lgp.LibraryError, match='stalled'):
        lgp.Chol(K, ctx=ctx)
    assert time.perf_counter() - t0 < 20
    c = lgp.Chol(K, ctx=ctx)     # launch-per-operation driver from here on
    util.assert_close_matrices(c.factor().to_host(), oracle.Chol(K)._L, rtol=1e-13)


def test_chain_kernel_single_worker_setting(lgp, rng):
    """ LGP_CHAIN_WORKERS=1 would leave no worker for the tasks the chain's own tasks wait for: the launch takes two """
    ctx = _context_with_env(lgp, LGP_CHAIN_WORKERS=1, LGP_CHAIN_TIMEOUT_MS=2000)
    K = randpd(rng, 640)
    c = lgp.Chol(K, ctx=ctx)
    util.assert_close_matrices(c.factor().to_host(), oracle.Chol(K)._L, rtol=1e-13)


def test_gradient_large_and_repeatable(lgp, rng):
    """ `normal_density_derivs` at n = 4096 (K^-1 formed in a 2 n^2 workspace of the library), four times in a row on the same
    inputs: the same numbers every time, equal to the direct formulas 1/2 (tr(K^-1 dK) - z' dK z), z = K^-1 r
    (_linalg/_decomp.py:526-528).  Regression test of round 3: with the workspace taken from the stream-ordered pool
    (hipMallocAsync) every other call returned a wrong gradient at this size; the workspaces are now blocks of the library's own
    (csrc/api.hip, lgp_ws_alloc) """
    n = 4096
    A = rng.standard_normal((n, n))
    K = A @ A.T / n + np.eye(n)
    r = rng.standard_normal(n)
    dKh = []
    for _ in range(2):
        d = rng.standard_normal((n, n))
        dKh.append(d + d.T)
    dec = lgp.Chol(K)
    dK = [lgp.DeviceArray.from_host(d) for d in dKh]
    Kinv = np.linalg.inv(K + dec.eps * np.eye(n))
    z = Kinv @ r
    gref = np.array([0.5 * (np.sum(Kinv * d) - z @ d @ z) for d in dKh])
    outs = [dec.normal_density_derivs(r, dK, None) for _ in range(4)]
    for v, g, _ in outs:
        assert v == outs[0][0]
        np.testing.assert_array_equal(np.asarray(g), np.asarray(outs[0][1]))
    np.testing.assert_allclose(np.asarray(outs[0][1]), gref, rtol=1e-9)


def test_split_driver_hands_its_tail_to_the_chain_driver(lgp, driver_contexts):
    """ above N = 16384 the launch-per-block driver factors the leading panels and hands the last <= 8192 rows (a multiple of 128,
    here 137 x 128 = 17536 -> 7296 left after ten panels, so the chain kernel's last panel is ragged) to the chain-kernel
    driver, whose first square starts beside the previous trailing update (csrc/potrf.hip, LGP_SPLIT_CHAIN_TAIL): against the
    context that never uses the chain kernel, with 37 carried right-hand sides; and a bad pivot inside the tail is reported """
    from lsqfitgp_amd._GP import _gemm
    n, q = 17536, 37
    rng = np.random.default_rng(17)
    x = rng.uniform(0, n / 16, n)
    ctx_launch, ctx_chain = driver_contexts
    lib = __import__('lsqfitgp_amd')._lib.load()
    B = rng.standard_normal((n, q))
    V = rng.standard_normal((n, 3))
    res = []
    for ctx in (ctx_launch, ctx_chain):
        K = lgp.ExpQuad().gram(x, x, ctx=ctx)
        assert lib.lgp_add_diag_f64(ctx.handle, n, K.ptr, K.ld, None, ctypes.c_double(0.02)) == 0
        d = lgp.Chol(K, ctx=ctx, rhs=B)
        KV = _gemm(False, False, K, lgp.DeviceArray.from_host(V, ctx=ctx)).to_host() + d.eps * V
        util.assert_close_matrices(d.correlate(d.back_correlate(V)), KV, rtol=1e-12)          # L L^T V = (K + eps) V
        util.assert_close_matrices(d.correlate(d.solved()), B, rtol=1e-11)                     # L (L^-1 B) = B
        res.append((d.solved(), d.back_correlate(V), d.minus_log_normal_density(B[:, 0], value=True)[0], d.eps))
        del d, K
    util.assert_close_matrices(res[1][0], res[0][0], rtol=1e-10)
    util.assert_close_matrices(res[1][1], res[0][1], rtol=1e-12)
    assert res[1][2] == pytest.approx(res[0][2], rel=1e-12) and res[1][3] == res[0][3]
    K = lgp.ExpQuad().gram(x, x, ctx=ctx_chain)
    assert lib.lgp_add_diag_f64(ctx_chain.handle, n, K.ptr, K.ld, None, ctypes.c_double(0.02)) == 0
    Kh = K.block(15000, 15001, 15000, 15001)
    Kh.assign(lgp.DeviceArray.from_host(np.array([[-1.0]]), ctx=ctx_chain))
    with pytest.raises(np.linalg.LinAlgError):
        lgp.Chol(K, ctx=ctx_chain, epsrel=0)


@pytest.mark.parametrize('n,q', [(14464, 0), (15000, 3), (16512, 0), (18048, 5), (21120, 0)])
def test_driver_selection_boundaries(lgp, n, q):
    """ sizes around the switches between the drivers of the blocked factorisation (csrc/potrf.hip): 14464 = 113 x 128 just above
    the chain-kernel driver's limit (split driver + chain tail of 6272 rows); 15000 is no multiple of 128 (no chain tail); 16512,
    18048 and 21120 = 129, 141 and 165 x 128 (ragged tails of 8128, 7808 and 7808 rows); with and without carried right-hand sides:
    L L^T V = (K + eps) V and L (L^-1 B) = B """
    from lsqfitgp_amd._GP import _gemm
    rng = np.random.default_rng(n)
    x = rng.uniform(0, n / 16, n)
    ctx = lgp.Context.default()
    lib = __import__('lsqfitgp_amd')._lib.load()
    K = lgp.ExpQuad().gram(x, x)
    assert lib.lgp_add_diag_f64(ctx.handle, n, K.ptr, K.ld, None, ctypes.c_double(0.02)) == 0
    B = rng.standard_normal((n, q)) if q else None
    d = lgp.Chol(K, rhs=B)
    V = rng.standard_normal((n, 2))
    KV = _gemm(False, False, K, lgp.DeviceArray.from_host(V)).to_host() + d.eps * V
    util.assert_close_matrices(d.correlate(d.back_correlate(V)), KV, rtol=1e-12)
    util.assert_close_matrices(d.ginv_linear(KV), V, rtol=1e-9)
    if q:
        util.assert_close_matrices(d.correlate(d.solved()), B, rtol=1e-11)


def _full_size_vs_cpu_backend(lgp, name, threads):
    """ a BASELINE configuration at FULL size, posterior mean / covariance / log marginal likelihood of the device step (bench.py's
    `Workload`: fused Gram + regularisation + factorisation with [Kxxs | y] carried, through the C-ABI) against the C++ / OpenMP CPU
    backend of oracle/ on the same inputs, at the north-star tolerances (VERDICT r03 "missing" 5).  The CPU side uses ONE n x n buffer
    (Gram built in place, factored in place).  Its worker threads flush subnormal numbers to zero (oracle/cpu_backend.cpp: FTZ | DAZ, as
    XLA's CPU runtime does): entries of the Gram matrix below 2.2e-308 are 0 on the CPU side and subnormal on the device -- a difference
    of at most 2.2e-308 per entry, 300 orders of magnitude below the 1e-8 that is asserted here. """
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from oracle import cpu as cpu_backend
    cfg = bench.CONFIGS[name]
    n, m = cfg['n'], cfg['m']
    ctx = lgp.Context.default()
    wl = bench.Workload(cfg, ctx)
    wl.step()
    ctx.sync()
    mean_d, cov_d = wl.posterior_host()
    mll_d = wl.results['mll']
    del wl
    # ---- the same pass on the host
    x, xs, y = bench.synth(cfg)
    be = cpu_backend.CpuBackend(threads)
    kind, p = (be.EXPQUAD, 0) if cfg['kernel'] == 'expquad' else (be.MATERNP, 2)
    L = np.empty((n + m + 1, n))
    be.gram(kind, p, x, x, out=L[:n])
    L[np.diag_indices(n)] += 0.01
    L[n:n + m] = be.gram(kind, p, xs, x)
    L[n + m] = y
    Kss = be.gram(kind, p, xs, xs)
    be.chol_inplace(L, n)
    S = L[n:]                                        # (L^-1 [Kxxs | y])^T
    G = S @ S.T
    mean_h = G[:m, m]
    cov_h = Kss - G[:m, :m]
    logdet = 2 * np.sum(np.log(L[np.arange(n), np.arange(n)]))
    mll_h = -0.5 * (n * np.log(2 * np.pi) + logdet + G[m, m])
    assert np.linalg.norm(mean_d - mean_h) <= 1e-8 * np.linalg.norm(mean_h), np.linalg.norm(mean_d - mean_h) / np.linalg.norm(mean_h)
    assert np.linalg.norm(cov_d - cov_h, 2) <= 1e-8 * np.linalg.norm(Kss, 2), np.linalg.norm(cov_d - cov_h, 2) / np.linalg.norm(Kss, 2)
    assert abs(mll_d - mll_h) <= 1e-8 * abs(mll_h) + 1e-6, (mll_d, mll_h)


def _host_budget():
    import os
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 0
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 1
    return avail, cpus


def test_config_c3_full_vs_cpu_backend(lgp):
    """ C3 (Matern-5/2, 3-D, N = 32768, M = 100) at full size against the CPU backend: ~10 s of host work on 16 threads, 8.6 GB """
    avail, cpus = _host_budget()
    if avail < 14 << 30:
        pytest.skip(f'needs 9 GB of host memory for the CPU backend ({avail >> 30} GB available)')
    _full_size_vs_cpu_backend(lgp, 'c3', min(cpus, 16))


def test_config_c4_full_vs_cpu_backend(lgp):
    """ C4's workload (ExpQuad 1-D, N = 65536, M = 100) at full size on ONE GPU against the CPU backend: ~85 s of host work on 16
    threads, one 34 GB host buffer """
    avail, cpus = _host_budget()
    if avail < 44 << 30:
        pytest.skip(f'needs 34 GB of host memory for the CPU backend ({avail >> 30} GB available)')
    _full_size_vs_cpu_backend(lgp, 'c4', min(cpus, 16))


@pytest.mark.parametrize('ashape,bshape', [((), (5, 'n', 'n')), ((3, 1), (2, 'n', 4)), ((2,), ('n',)), ((), ('n', 3)), ((2, 3), (2, 3, 'n', 1))])
def test_solve_batched_matmul_like(lgp, rng, ashape, bshape):
    """ `_linalg.solve_batched` (_linalg/_decomp.py:330-347) as the reference's tests use it (tests/linalg/test_decomp.py:246:
    ``solve_batched(K, moveaxis(dK, 2, 0), assume_a='pos')``): leading axes broadcast matmul-like, a vector stays a vector; numpy's
    batched LAPACK solve is the restatement of ``jnp.vectorize(jlinalg.solve)`` """
    from lsqfitgp_amd import _linalg
    n = 150
    a = np.empty(ashape + (n, n))
    for idx in np.ndindex(*ashape):
        a[idx] = randpd(rng, n)
    bs = tuple(n if v == 'n' else v for v in bshape)
    b = rng.standard_normal(bs)
    got = _linalg.solve_batched(a, b, assume_a='pos')
    if b.ndim < 2:
        want = np.linalg.solve(a, np.broadcast_to(b[:, None], ashape + (n, 1)))[..., 0]
    else:
        batch = np.broadcast_shapes(ashape, bs[:-2])
        want = np.linalg.solve(np.broadcast_to(a, batch + (n, n)), np.broadcast_to(b, batch + bs[-2:]))
    assert got.shape == want.shape
    util.assert_close_matrices(got.reshape(-1, got.shape[-1]), want.reshape(-1, want.shape[-1]), rtol=1e-10)

"""
The C++ / OpenMP CPU restatement of the hot path (oracle/cpu_backend.cpp: bench.py's CPU baseline) pinned
against the numpy / scipy oracle: Gram entries, the regularised Cholesky (scaling, jitter, info), the solves, the posterior and
the log-density.  CPU only.
"""

import numpy as np
import pytest

import oracle
from oracle import cpu


@pytest.fixture(scope='module')
def be():
    return cpu.CpuBackend(4)


@pytest.mark.parametrize('n,m,d,kernel,p', [(300, 77, 1, 0, 0), (257, 31, 3, 1, 2), (64, 64, 2, 1, 0), (100, 5, 5, 1, 3)])
def test_gram(be, rng, n, m, d, kernel, p):
    x = rng.uniform(0, 6, (d, n)); y = rng.uniform(0, 6, (d, m))
    got = be.gram(kernel, p, x, y, scale=1.3)
    k = oracle.ExpQuad(scale=1.3) if kernel == 0 else oracle.Maternp(p=p, scale=1.3)
    dt = [(f'f{i}', float) for i in range(d)]
    a = np.empty(n, dt); b = np.empty(m, dt)
    for i, f in enumerate(a.dtype.names):
        a[f] = x[i]; b[f] = y[i]
    want = k(a[:, None], b[None, :])
    np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-300)


@pytest.mark.parametrize('n,q,amp', [(50, 0, 1.), (300, 7, 37.), (777, 3, 1.), (1100, 101, 0.004)])
def test_chol_and_solves(be, rng, n, q, amp):
    A = rng.standard_normal((n, n))
    K = amp * (A @ A.T / n + 0.5 * np.eye(n))
    B = rng.standard_normal((n, q)) if q else None
    L, eps = be.chol(K, rhs=B)
    o = oracle.Chol(K)
    assert eps == pytest.approx(o.eps, rel=1e-14)
    np.testing.assert_allclose(np.tril(L[:n]), o._L, rtol=1e-10, atol=1e-13 * np.abs(o._L).max())
    assert np.all(np.triu(L[:n], 1) == 0)
    if q:
        np.testing.assert_allclose(L[n:].T, o.pinv_correlate(B), rtol=1e-9, atol=1e-11)
    R = rng.standard_normal((n, 5))
    y = be.trsm(L[:n], R)
    np.testing.assert_allclose(y, o.pinv_correlate(R), rtol=1e-9, atol=1e-11)
    z = be.trsm(L[:n], y, trans=True)
    np.testing.assert_allclose(z, o.ginv_linear(R), rtol=1e-8, atol=1e-10)


def test_bad_pivot(be, rng):
    A = rng.standard_normal((300, 300))
    K = A @ A.T
    K[200, 200] = -1
    with pytest.raises(np.linalg.LinAlgError):
        be.chol(K, epsrel=0.)


@pytest.mark.parametrize('shape', [(33, 45, 17, False, False), (300, 257, 513, False, True), (129, 200, 1000, True, False), (64, 12, 300, True, True)])
def test_gemm(be, rng, shape):
    m, n, k, ta, tb = shape
    A = rng.standard_normal((k, m) if ta else (m, k)); B = rng.standard_normal((n, k) if tb else (k, n))
    want = (A.T if ta else A) @ (B.T if tb else B)
    np.testing.assert_allclose(be.gemm(A, B, ta, tb), want, rtol=1e-12, atol=1e-12)


def test_fit_predict_against_the_oracle(be, rng):
    """ the bench leg: Gram blocks -> Chol -> posterior mean / covariance / log marginal likelihood, against oracle.GP """
    n, m = 900, 40
    x = rng.uniform(0, n / 16, (1, n)); xs = rng.uniform(0, n / 16, (1, m))
    y = np.sin(x[0] / 4) + 0.1 * rng.standard_normal(n)
    Kxx = be.gram(0, 0, x, x)
    Kxx[np.diag_indices(n)] += 0.01
    L, eps = be.chol(Kxx)
    W = np.concatenate([be.gram(0, 0, x, xs), y[:, None]], axis=1)
    mean, cov, mlnd = be.posterior(L[:n], W, be.gram(0, 0, xs, xs))
    go = oracle.GP(oracle.ExpQuad(), checkpos=False, checksym=False).addx(x[0], 'd').addx(xs[0], 't')
    gc = {('d', 'd'): 0.01 * np.eye(n)}
    mo, co = go.predfromdata({'d': y}, 't', gc, raw=True)
    assert np.linalg.norm(mean - mo) <= 1e-8 * np.linalg.norm(mo)
    assert np.linalg.norm(cov - co, 2) <= 1e-8
    assert -mlnd == pytest.approx(go.marginal_likelihood({'d': y}, gc), rel=1e-10)

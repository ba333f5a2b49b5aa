"""
GPU tests of the callers of the hot path: `empbayes_fit` over the device GP (known answers
from the reference's user guide, analytic device gradients against finite differences), `raniter` / `sample`
(the reference's tests/test_raniter.py identities), and `bayestree.bart` (affine equivariance,
tests/bayestree/test_bart.py:65-78, and the C5 workload).
"""

import numpy as np
import pytest
from scipy import stats

import oracle
from . import util
from .test_fit_host import X11, Y11, fit_scale, fit_sdev_scale, check_scale, check_sdev_scale, fit_period, check_period

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lgp():
    import lsqfitgp_amd
    lsqfitgp_amd.Context.default()
    return lsqfitgp_amd


# ---- empbayes_fit ----------------------------------------------------------------------------------------------------

def test_known_answer_scale_fit_device(lgp):
    """ docs/userguide/hyper.rst:49-70: ``2.16(17)``, objective evaluated on the device """
    fit = fit_scale(lgp)
    check_scale(fit)
    ref = fit_scale(oracle)
    assert abs(fit.pmean['log(scale)'] - ref.pmean['log(scale)']) <= 2e-6


def test_known_answer_period_fit_device(lgp):
    """ ``docs/userguide/hyper.rst:406-447``: ``log(period) 1.8373(29)``, ``period 6.280(18)`` with every evaluation on the device,
    and the same mode as the oracle's run of the same fit """
    fit = fit_period(lgp)
    check_period(fit)
    assert abs(fit.pmean['log(period)'] - fit_period(oracle).pmean['log(period)']) <= 2e-6


def test_known_answer_sdev_scale_fit_device(lgp):
    """ docs/userguide/hyper.rst:124-156,252-257 """
    check_sdev_scale(fit_sdev_scale(lgp))


def _grad_sdev_scale(lgp):
    """ analytic derivatives for the model of hyper.rst:124-139: K = sdev^2 E(scale); dK/dsdev = 2 sdev E,
    dK/dscale = sdev^2 dE/dscale (lgp_gram_isotropic_dscale) """
    def gradfactory(hp):
        k = lgp.ExpQuad(scale=hp['scale']) * hp['sdev'] ** 2
        K, dKs = k.gram_dscale(X11, X11)
        dKa = lgp.ExpQuad(scale=hp['scale']) * (2 * hp['sdev'])
        return lgp.Chol(K), Y11, [dKa.gram(X11, X11), dKs], None, ['sdev', 'scale']
    return gradfactory


def test_analytic_gradient_matches_finite_differences(lgp):
    fit = fit_sdev_scale(lgp, gradfactory=_grad_sdev_scale(lgp), covariance='fisher')
    check_sdev_scale(fit)
    z = np.array([0.3, -0.4])
    f0, g = fit._fun_and_jac(z)
    assert f0 == pytest.approx(fit._fun(z), rel=1e-12)
    for i in range(2):
        e = np.zeros(2); e[i] = 1e-6
        fd = (fit._fun(z + e) - fit._fun(z - e)) / 2e-6
        assert g[i] == pytest.approx(fd, rel=2e-6, abs=1e-7)
    # Fisher + prior precision: symmetric positive definite, and the reported covariance is its inverse
    F = fit._fisher(fit.zmean)
    assert np.allclose(F, F.T) and np.all(np.linalg.eigvalsh(F) > 0)
    np.testing.assert_allclose(fit.zcov, np.linalg.inv(F), rtol=1e-10)
    # Newton steps on the Fisher matrix reach the same mode
    newton = fit_sdev_scale(lgp, gradfactory=_grad_sdev_scale(lgp), method='fisher')
    assert abs(newton.pmean['log(sdev)'] - fit.pmean['log(sdev)']) <= 1e-4
    assert abs(newton.pmean['log(scale)'] - fit.pmean['log(scale)']) <= 1e-4


def test_fit_larger_problem_recovers_scale(lgp, rng):
    """ N = 600 noisy samples of a GP with scale 1.7: the fit (analytic device gradient) lands near the truth and agrees with
    the finite-difference fit """
    n = 600
    x = np.sort(rng.uniform(0, 60, n))
    Ktrue = oracle.ExpQuad(scale=1.7)(x[:, None], x[None, :]) + 0.04 * np.eye(n)
    y = np.linalg.cholesky(Ktrue) @ rng.standard_normal(n)
    prior = {'log(scale)': (0., 1.), 'log(s2)': (np.log(0.1), 1.)}

    def makegp(hp):
        return (lgp.GP(lgp.ExpQuad(scale=hp['scale']), checkpos=False, checksym=False).addx(x, 'f')
                .addcov(hp['s2'] * np.eye(n), 'noise').addtransf({'f': 1, 'noise': 1}, 'y'))

    def gradfactory(hp):
        K, dK = lgp.ExpQuad(scale=hp['scale']).gram_dscale(x, x)
        eye = lgp.DeviceArray.from_host(np.eye(n))
        from lsqfitgp_amd._GP import _axpby
        Kn = _axpby(1., K, hp['s2'], eye)
        return lgp.Chol(Kn, epsrel=0), y, [dK, eye], None, ['scale', 's2']

    fa = lgp.empbayes_fit(prior, makegp, {'y': y}, gradfactory=gradfactory, mlkw=dict(epsrel=0), raises=False)
    assert abs(fa.pmean['scale'] - 1.7) <= 0.25 and abs(fa.pmean['s2'] - 0.04) <= 0.015
    ff = lgp.empbayes_fit(prior, makegp, {'y': y}, mlkw=dict(epsrel=0), raises=False)
    assert abs(ff.pmean['log(scale)'] - fa.pmean['log(scale)']) <= 1e-4
    assert abs(ff.pmean['log(s2)'] - fa.pmean['log(s2)']) <= 1e-4


# ---- raniter / sample (reference tests/test_raniter.py) ---------------------------------------------------------------------

def make_mean_cov(rng, n):
    mean = rng.standard_normal(n)
    a = rng.standard_normal((n, n))
    return mean, a.T @ a


def test_raniter_randomness(lgp, rng):
    """ ref :55-66 """
    n = 40
    mean, cov = make_mean_cov(rng, n)
    samples = list(lgp.raniter(mean, cov, 10 * n, rng=rng))
    smean = np.mean(samples, axis=0)
    scov = np.cov(samples, rowvar=False, ddof=1)
    w, v = np.linalg.eigh(scov / len(samples))
    vsm = v.T @ (smean - mean)
    eps = n * 1e-12 * np.max(w)
    q = (vsm.T / np.maximum(w, eps)) @ vsm
    assert stats.chi2(n).sf(q) > 1e-5
    assert stats.chi2(n).cdf(q) > 1e-5


def test_raniter_packing_and_oracle_stream(lgp, rng):
    """ ref :68-84: dictionary and array inputs give the same stream; and the stream is the oracle's ``mean + L z`` """
    n = 3
    mean, cov = make_mean_cov(rng, n + 4)
    dmean = {'a': mean[:n], 'b': mean[n:]}
    dcov = {('a', 'a'): cov[:n, :n], ('a', 'b'): cov[:n, n:], ('b', 'a'): cov[n:, :n], ('b', 'b'): cov[n:, n:]}
    seed = int(rng.integers(2 ** 63))
    s1 = next(lgp.raniter(mean, cov, rng=seed))
    s2 = next(lgp.raniter(dmean, dcov, rng=seed))
    assert isinstance(s2, dict)
    np.testing.assert_array_equal(s1, np.concatenate([s2['a'], s2['b']]))
    z = np.random.default_rng(seed).standard_normal(n + 4)
    util.assert_close_matrices(s1, mean + oracle.Chol(cov).correlate(z), rtol=1e-12)
    assert np.ndim(lgp.sample(2., 9., rng=seed)) == 0
    assert lgp.sample(np.zeros((2, 2)), np.eye(4).reshape(2, 2, 2, 2), rng=seed).shape == (2, 2)


def test_raniter_counts_and_errors(lgp, rng):
    """ ref :86-120: n bounds the iterations; an indefinite matrix raises with the reference's message """
    mean, cov = make_mean_cov(rng, 5)
    assert len(list(lgp.raniter(mean, cov, 7, rng=rng))) == 7
    with pytest.raises(np.linalg.LinAlgError, match='covariance matrix not positive definite with eps=None'):
        next(lgp.raniter(mean, -cov, rng=rng))
    # a singular covariance is sampled thanks to the eps regularisation (samples stay in its range, to rounding)
    a = rng.standard_normal((5, 2))
    sing = a @ a.T
    s = lgp.sample(np.zeros(5), sing, eps=1e-10, rng=rng)
    w, v = np.linalg.eigh(sing)
    assert np.max(np.abs(v[:, :3].T @ s)) <= 1e-3 * np.max(np.abs(s))


def test_posterior_sampling_large(lgp, rng):
    """ the O(m^3) sampling step on a 3000-point posterior covariance: sample moments against the posterior """
    n, m = 200, 3000
    x = rng.uniform(0, 20, n); xs = np.linspace(0, 20, m)
    y = np.sin(x)
    gp = lgp.GP(lgp.ExpQuad(), checkpos=False, checksym=False).addx(x, 'd').addx(xs, 't')
    mean, cov = gp.predfromdata({'d': y}, 't', {('d', 'd'): 0.01 * np.eye(n)}, raw=True)
    draws = np.array(list(lgp.raniter(mean, cov, 64, rng=rng)))
    sd = np.sqrt(np.maximum(np.diag(cov), 0))
    zscore = (draws.mean(axis=0) - mean) / (sd / 8 + 1e-9)
    assert np.abs(zscore).max() < 6
    # pointwise sample variances against the posterior variances, averaged over the points
    ratio = draws.var(axis=0, ddof=1).sum() / np.diag(cov).sum()
    assert 0.8 < ratio < 1.25


# ---- bayestree.bart ---------------------------------------------------------------------------------------------------------

def bart_problem(rng, n=101, p=11):
    X = rng.uniform(-2, 2, (n, p))
    y = np.sum(np.cos(2 * np.pi / 2 * X), axis=1) + 0.1 * rng.standard_normal(n)
    return X, y


def test_bart_gradient_matches_finite_differences(lgp, rng):
    """ the analytic derivatives of bayestree.bart's GP (BART alpha/beta tangents, k, sigma2 through the copula /
    log parametrisation) against central differences of the device objective """
    X, y = bart_problem(rng, 60, 4)
    b = lgp.bayestree.bart(X, y, fitkw=dict(verbosity=0, minkw=dict(method='l-bfgs-b', options=dict(maxiter=2))))
    fit = b.fit
    z = np.array([0.2, -0.3, 0.1, -0.2])
    f0, g = fit._fun_and_jac(z)
    assert f0 == pytest.approx(fit._fun(z), rel=1e-10)
    for i in range(4):
        e = np.zeros(4); e[i] = 1e-5
        fd = (fit._fun(z + e) - fit._fun(z - e)) / 2e-5
        assert g[i] == pytest.approx(fd, rel=1e-5, abs=1e-6)
    # explicit mean instead of the marginalised one: five hyperparameters, dr/dmean = -1
    b2 = lgp.bayestree.bart(X, y, marginalize_mean=False, fitkw=dict(verbosity=0, minkw=dict(method='l-bfgs-b', options=dict(maxiter=2))))
    z = np.array([0.2, -0.3, 0.1, -0.2, 0.15])
    f0, g = b2.fit._fun_and_jac(z)
    for i in range(5):
        e = np.zeros(5); e[i] = 1e-5
        fd = (b2.fit._fun(z + e) - b2.fit._fun(z - e)) / 2e-5
        assert g[i] == pytest.approx(fd, rel=1e-5, abs=1e-6)


def test_bart_scale_shift(lgp, rng):
    """ reference tests/bayestree/test_bart.py:65-78: the regression is equivariant under affine maps of the outcome, at the
    reference's own tolerances (rtol 1e-10 on the mean, 1e-11 on the covariance): the two L-BFGS runs see objectives that differ
    by a constant and gradients equal to rounding, and follow the same path (measured: 3e-11 / 8e-14, tools/bart_equivariance.py) """
    X, y = bart_problem(rng)
    kw = dict(fitkw=dict(verbosity=0))
    bart1 = lgp.bayestree.bart(X, y, **kw)
    offset, scale = 0.4703189, 0.5294714
    bart2 = lgp.bayestree.bart(X, offset + y * scale, **kw)
    m1, cov1 = bart1.pred()
    m2, cov2 = bart2.pred()
    util.assert_allclose(m2, offset + m1 * scale, rtol=1e-10)
    util.assert_allclose(cov2, cov1 * scale ** 2, rtol=1e-11, atol=1e-12 * np.max(np.abs(cov2)))
    assert bart2.fit.pmean['alpha'] == pytest.approx(bart1.fit.pmean['alpha'], rel=1e-10)
    assert bart2.sigma.mean == pytest.approx(bart1.sigma.mean * scale, rel=1e-10)
    # sanity of the regression itself: in-sample residuals are of the order of the noise
    assert np.std(m1 - y) < 0.5 * np.std(y)
    text = repr(bart1)
    assert 'alpha =' in text and 'error sdev =' in text


def test_bart_interface(lgp, rng):
    """ gp / data / pred with test points, weights, error=True, hp dictionaries and samples (``_bart.py:242-404``) """
    X, y = bart_problem(rng, 80, 3)
    w = rng.uniform(0.5, 2, 80)
    b = lgp.bayestree.bart(X, y, weights=w, fitkw=dict(verbosity=0))
    Xt = rng.uniform(-2, 2, (15, 3))
    m, c = b.pred(x_test=Xt)
    assert m.shape == (15,) and c.shape == (15, 15)
    me, ce = b.pred(x_test=Xt, error=True, weights=np.full(15, 2.))
    # 'test' = 'testmean' + 'testnoise' + the marginalised intercept (``_bart.py:298-302``): the outcome's posterior mean
    # differs from the latent one by the posterior mean of the intercept, the same number at every point
    assert np.ptp(me - m) <= 1e-9 * np.max(np.abs(m)) and ce.shape == (15, 15)
    gp = b.gp(x_test=Xt)
    mm, cc = gp.predfromdata(b.data(), 'testmean', raw=True)
    np.testing.assert_allclose(mm + b.mean, m, rtol=1e-12)
    hp = dict(b.fit.pmean)
    m2, _ = b.pred(hp=hp, x_test=Xt)
    np.testing.assert_array_equal(m2, m)
    ms, _ = b.pred(hp='sample', x_test=Xt, rng=3)
    assert ms.shape == (15,) and not np.array_equal(ms, m)
    with pytest.raises(NotImplementedError):
        b.pred(format='gvar')
    with pytest.raises(KeyError):
        b.pred(hp='nope')
    # the same fit through the oracle's GP at the device's MAP: posterior parity of the caller's output
    from numpy.lib import recfunctions
    Xs, Xts = recfunctions.unstructured_to_structured(X), recfunctions.unstructured_to_structured(Xt)
    gpo = (oracle.GP(oracle.BART(splits=oracle.BART.splits_from_coord(X), alpha=hp['alpha'], beta=hp['beta'], maxd=10, reset=[2, 4, 6, 8])
                     * (((y.max() - y.min()) / 2) / hp['k']) ** 2, checkpos=False, checksym=False)
           .addx(Xs, 'trainmean').addcov(np.diag(hp['sigma2'] / w), 'trainnoise').addcov(((y.max() - y.min()) / 2) ** 2, 'mean')
           .addtransf({'trainmean': 1, 'trainnoise': 1, 'mean': 1}, 'train').addx(Xts, 'testmean'))
    mo, co = gpo.predfromdata({'train': y - (y.max() + y.min()) / 2}, 'testmean', raw=True)
    util.assert_close_matrices(m - (y.max() + y.min()) / 2, mo, rtol=1e-8)
    util.assert_close_matrices(c, co, rtol=1e-8)
    gpo = gpo.addcov(np.diag(hp['sigma2'] / np.full(15, 2.)), 'testnoise').addtransf({'testmean': 1, 'testnoise': 1, 'mean': 1}, 'test')
    meo, ceo = gpo.predfromdata({'train': y - (y.max() + y.min()) / 2}, 'test', raw=True)
    util.assert_close_matrices(me - (y.max() + y.min()) / 2, meo, rtol=1e-8)
    util.assert_close_matrices(ce, ceo, rtol=1e-8)


def test_bart_on_the_c5_data(lgp):
    """ bayestree.bart on the reference's examples/bart-data.txt (first 500 rows, like examples/bart.py:57): the fit runs to
    convergence on the device and predicts held-out rings better than the training mean """
    from workloads import load_abalone
    X, y = load_abalone()
    X = X.astype(float); y = y.astype(float)
    b = lgp.bayestree.bart(X[:500], y[:500], fitkw=dict(verbosity=0))
    assert 0 < b.alpha.mean < 1 and b.beta.mean > 0 and b.sigma.mean > 0
    # error=True: the outcome 'test' = latent + noise + marginalised intercept (the latent 'testmean' alone leaves the
    # posterior of the intercept out, in the reference as here, ``_bart.py:388-404``)
    m, c = b.pred(x_test=X[500:700], error=True)
    rmse = np.sqrt(np.mean((m - y[500:700]) ** 2))
    assert rmse < 0.8 * np.std(y[500:700])
    assert np.all(np.diag(c) > 0)


def test_bart_fit_at_c5_size(lgp):
    """ BASELINE configuration 5 as the caller runs it: ``bayestree.bart`` on the first 4096 rows of the reference's
    examples/bart-data.txt (float64, the reference's default), the whole L-BFGS fit on the device (every iteration: BART Gram
    matrix with its alpha / beta tangents, factorisation, likelihood gradient), then the posterior at the 81 held-out rows
    against the ORACLE's GP built at the device's MAP hyperparameters (north-star tolerances), and the objective of the fit
    at the MAP against the oracle's """
    import time
    from numpy.lib import recfunctions
    from workloads import load_abalone
    X, y = load_abalone()
    X = X.astype(float); y = y.astype(float)
    n = 4096
    t0 = time.perf_counter()
    b = lgp.bayestree.bart(X[:n], y[:n], fitkw=dict(verbosity=0))
    fit_s = time.perf_counter() - t0
    hp = dict(b.fit.pmean)
    calls = b.fit._calls['grad']
    print(f'bart fit N={n}: {fit_s:.2f} s, {b.fit.minresult.nit} iterations, {calls} objective+gradient evaluations, '
          f'alpha={hp["alpha"]:.4f} beta={hp["beta"]:.4f} k={hp["k"]:.4f} sigma2={hp["sigma2"]:.4f}')
    assert 0 < hp['alpha'] < 1 and hp['beta'] > 0 and hp['sigma2'] > 0 and b.fit.minresult.nit >= 3
    # the fit moved to a better point than the prior mean, and its gradient there is small against the one it started with
    z0 = np.zeros(4)
    f0, g0 = b.fit._fun_and_jac(z0)
    f1, g1 = b.fit._fun_and_jac(b.fit.zmean)
    assert f1 < f0 and np.linalg.norm(g1) < 0.05 * np.linalg.norm(g0)
    m, c = b.pred(x_test=X[n:])
    me, ce = b.pred(x_test=X[n:], error=True)
    rmse = np.sqrt(np.mean((me - y[n:]) ** 2))
    assert rmse < 0.95 * np.std(y[n:])        # (0.90 on these 81 rows; the in-sample fit explains far more, below)
    mt, _ = b.pred()
    assert np.std(mt - y[:n]) < 0.65 * np.std(y[:n])
    # the oracle's GP at the same hyperparameters (numpy/scipy restatement of the reference's path; ~30 s of host time)
    ymin, ymax = y[:n].min(), y[:n].max()
    centre, half = (ymax + ymin) / 2, (ymax - ymin) / 2
    Xs, Xt = recfunctions.unstructured_to_structured(X[:n]), recfunctions.unstructured_to_structured(X[n:])
    kern = oracle.BART(splits=oracle.BART.splits_from_coord(X[:n]), alpha=hp['alpha'], beta=hp['beta'], maxd=10, reset=[2, 4, 6, 8]) * (half / hp['k']) ** 2
    gpo = (oracle.GP(kern, checkpos=False, checksym=False)
           .addx(Xs, 'trainmean').addcov(np.diag(np.full(n, hp['sigma2'])), 'trainnoise').addcov(half ** 2, 'mean')
           .addtransf({'trainmean': 1, 'trainnoise': 1, 'mean': 1}, 'train').addx(Xt, 'testmean'))
    given = {'train': y[:n] - centre}
    mo, co = gpo.predfromdata(given, 'testmean', raw=True)
    assert np.linalg.norm(m - centre - mo) <= 1e-8 * np.linalg.norm(mo)
    kss = np.linalg.norm(np.asarray(gpo.prior('testmean', raw=True)), 2)
    assert np.linalg.norm(c - co, 2) <= 1e-8 * kss
    lo = gpo.marginal_likelihood(given, epsrel=0)
    ld = b.gp().marginal_likelihood(b.data(), epsrel=0)
    assert abs(ld - lo) <= 1e-8 * abs(lo) + 1e-6
    # the objective the minimiser saw at the MAP: minus log likelihood + the standard-normal prior of z
    z = b.fit.zmean
    assert f1 == pytest.approx(-lo + 0.5 * (len(z) * np.log(2 * np.pi) + z @ z), rel=1e-9)


# ---- derivatives of the factory's own arithmetic (`lsqfitgp_amd.Dual`, the stand-in for jax.jacfwd of the factory) ------------

def test_kernel_tangents_match_finite_differences(lgp, rng):
    """ `Kernel.gram_tangents`: scale (through a chain of host arithmetic), the constants of kernel * c and kernel + c """
    D = lgp.Dual
    x = rng.uniform(0, 10, 300)
    y = rng.uniform(0, 10, 200)

    def build(mod, z, a, c, cls, **kw):
        return (cls(scale=mod_exp(z), **kw) * a ** 2 + c) * 0.5

    mod_exp = np.exp
    for cls, kw in ((lgp.ExpQuad, {}), (lgp.Maternp, dict(p=2)), (lgp.GammaExp, dict(gamma=1.3)), (lgp.Cauchy, dict(alpha=1.5, beta=2.5))):
        z0, a0, c0 = 0.3, 1.4, 0.2
        kern = build(lgp, D(z0, {'z': 1.}), D(a0, {'a': 1.}), D(c0, {'c': 1.}), cls, **kw)
        assert sorted(kern.tangent_keys()) == ['a', 'c', 'z']
        K, tans = kern.gram_tangents(x, y)
        plain = lambda z, a, c: build(lgp, z, a, c, cls, **kw).gram(x, y).to_host()
        np.testing.assert_array_equal(K.to_host(), plain(z0, a0, c0))          # the values are those of the plain kernel
        h = 1e-6
        fd = {'z': (plain(z0 + h, a0, c0) - plain(z0 - h, a0, c0)) / (2 * h),
              'a': (plain(z0, a0 + h, c0) - plain(z0, a0 - h, c0)) / (2 * h),
              'c': (plain(z0, a0, c0 + h) - plain(z0, a0, c0 - h)) / (2 * h)}
        for key in 'zac':
            np.testing.assert_allclose(tans[key].to_host(), fd[key], rtol=2e-7, atol=2e-9, err_msg=f'{cls.__name__} {key}')
    # a kernel without tangents, and one whose tangent is not available
    K, tans = lgp.ExpQuad(scale=2.).gram_tangents(x, y)
    assert tans == {}
    with pytest.raises(NotImplementedError):
        lgp.Wendland(scale=D(2., {'s': 1.})).gram_tangents(x, y)


def test_fit_without_gradfactory_uses_the_tangents_of_the_factory(lgp):
    """ docs/userguide/hyper.rst:124-156 again, with no gradfactory: the gradient comes from the Dual hyperparameters """
    fit = fit_sdev_scale(lgp, covariance='fisher')
    assert fit._autodiff and fit._autodiff_checked
    check_sdev_scale(fit)
    z = np.array([0.3, -0.4])
    f0, g = fit._fun_and_jac(z)
    assert f0 == pytest.approx(fit._fun(z), rel=1e-12)
    for i in range(2):
        e = np.zeros(2); e[i] = 1e-6
        assert g[i] == pytest.approx((fit._fun(z + e) - fit._fun(z - e)) / 2e-6, rel=2e-6, abs=1e-7)
    # the same numbers as the hand-written gradfactory, Fisher matrix included
    ref = fit_sdev_scale(lgp, gradfactory=_grad_sdev_scale(lgp), covariance='fisher')
    np.testing.assert_allclose(fit._fun_and_jac(z)[1], ref._fun_and_jac(z)[1], rtol=1e-9)
    np.testing.assert_allclose(fit._fisher(z), ref._fisher(z), rtol=1e-8)
    np.testing.assert_allclose(fit.zcov, ref.zcov, rtol=1e-4)
    # and Newton steps on the Fisher matrix work without a gradfactory
    newton = fit_sdev_scale(lgp, method='fisher')
    assert abs(newton.pmean['log(scale)'] - fit.pmean['log(scale)']) <= 1e-4


def _composite_model(lgp, rng, n=160, lose_tangent=False, kernel=None):
    """ kernel scale and amplitude, a noise block, a transformation coefficient, a givencov block and data that all depend on
    hyperparameters """
    x = np.sort(rng.uniform(0, 20, n))
    y = np.sin(x) + 0.3 * rng.standard_normal(n)
    prior = {'log(scale)': (np.log(2.), 0.5), 'log(amp)': (0., 0.5), 'log(s2)': (np.log(0.1), 0.5), 'w': (1., 0.2), 'mu': (0.1, 0.5),
             'log(g)': (np.log(0.05), 0.3)}

    def makegp(hp):
        k = (kernel or lgp.Maternp)(scale=hp['scale'], **({} if kernel else dict(p=2))) * hp['amp']
        noise = np.diag(np.full(n, float(hp['s2']))) if lose_tangent else lgp.diag(hp['s2'] * np.ones(n))
        return (lgp.GP(k, checkpos=False).addx(x, 'f').addcov(noise, 'noise').addtransf({'f': 1., 'noise': hp['w']}, 'y'))

    def data(hp):
        return {'y': y - hp['mu']}, {('y', 'y'): hp['g'] * np.eye(n)}

    return prior, makegp, data


def test_tangents_through_blocks_transformations_and_data(lgp, rng):
    prior, makegp, data = _composite_model(lgp, rng)
    fit = lgp.empbayes_fit(prior, makegp, data, raises=False, minkw=dict(options=dict(maxiter=3)))
    assert fit._autodiff and fit._autodiff_checked
    z = np.array([0.2, -0.3, 0.1, 0.4, -0.2, 0.3])
    value, g, F = fit._auto_derivs(z, True)
    np.testing.assert_allclose(g, fit._fd_gradient(z), rtol=2e-6, atol=1e-6)
    assert value + fit._prior_value(z) == pytest.approx(fit._fun(z), rel=1e-12)
    # (s2, w and g only enter through w^2 s2 + g: two flat directions; the mean shift has information of its own, uncorrelated
    # with the covariance parameters)
    w = np.linalg.eigvalsh(F)
    assert np.allclose(F, F.T) and np.all(w > -1e-10 * w[-1]) and np.sum(w > 1e-8 * w[-1]) == 4
    assert F[4, 4] > 0 and np.all(F[4, [0, 1, 2, 3, 5]] == 0)
    assert np.all(np.linalg.eigvalsh(fit._fisher(z)) >= 1 - 1e-9)


def test_uncovered_factory_falls_back_to_central_differences(lgp, rng):
    prior, makegp, data = _composite_model(lgp, rng, kernel=lgp.Wendland)       # no closed-form scale tangent for this core
    fit = lgp.empbayes_fit(prior, makegp, data, raises=False, minkw=dict(options=dict(maxiter=2)))
    assert not fit._autodiff
    assert np.all(np.isfinite(fit._fun_and_jac(fit.zmean)[1]))


def test_lost_tangent_is_noticed(lgp, rng):
    """ ``float(hp['s2'])`` turns the hyperparameter into a plain number: the first gradient is checked against central
    differences and the fit goes on with those """
    prior, makegp, data = _composite_model(lgp, rng, lose_tangent=True)
    fit = lgp.empbayes_fit(prior, makegp, data, raises=False, minkw=dict(options=dict(maxiter=2)))
    assert fit._autodiff_checked and not fit._autodiff
    z = fit.zmean
    np.testing.assert_allclose(fit._fun_and_jac(z)[1] - z, fit._fd_gradient(z), rtol=1e-6, atol=1e-7)


def test_bart_hyperparameters_as_duals(lgp, rng):
    """ BART(alpha=Dual, beta=Dual): the tangents of `gram_jvp` combined along the keys; a factory written with them gets an
    analytic gradient from `empbayes_fit` without a gradfactory """
    D = lgp.Dual
    n, p = 150, 3
    Xp = rng.uniform(0, 1, (n, p))
    y = np.sin(3 * Xp[:, 0]) + Xp[:, 1] ** 2 + 0.1 * rng.standard_normal(n)
    X = np.empty(n, [(f'x{i}', float) for i in range(p)])
    for i in range(p):
        X[f'x{i}'] = Xp[:, i]
    splits = lgp.BART.splits_from_coord(X)
    za, zb = 0.4, 0.7
    f = lambda a, b: lgp.BART(splits=splits, alpha=a, beta=b, maxd=4, reset=2)
    k = f(D(za, {0: 1., 1: 0.5}), D(zb, {1: 2.}))
    assert sorted(k.tangent_keys()) == [0, 1]
    K, tans = k.gram_tangents(X, X)
    plain = lambda a, b: f(a, b).gram(X, X).to_host()
    h = 1e-5
    da = (plain(za + h, zb) - plain(za - h, zb)) / (2 * h)
    db = (plain(za, zb + h) - plain(za, zb - h)) / (2 * h)
    np.testing.assert_allclose(K.to_host(), plain(za, zb), rtol=1e-12)
    np.testing.assert_allclose(tans[0].to_host(), da, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(tans[1].to_host(), 0.5 * da + 2. * db, rtol=1e-5, atol=1e-7)

    prior = {'alpha': lgp.copula.beta(2, 1), 'beta': lgp.copula.invgamma(1, 1), 'log(s2)': (np.log(0.05), 1.)}

    def makegp(hp):
        kern = lgp.BART(splits=splits, alpha=hp['alpha'], beta=hp['beta'], maxd=4, reset=2)
        return lgp.GP(kern, checkpos=False).addx(X, 'f').addcov(lgp.diag(hp['s2'] * np.ones(n)), 'e').addtransf({'f': 1, 'e': 1}, 'y')

    fit = lgp.empbayes_fit(prior, makegp, {'y': y}, raises=False, minkw=dict(options=dict(maxiter=4)))
    assert fit._autodiff and fit._autodiff_checked
    z = np.array([0.2, -0.1, 0.3])
    np.testing.assert_allclose(fit._auto_derivs(z, False)[1], fit._fd_gradient(z), rtol=1e-5, atol=1e-6)


def test_fit_of_a_sum_of_kernels_differentiates_itself(lgp, rng):
    """ a factory with a sum and a product of kernels: the gradient comes from the tangents of the composite """
    n = 200
    x = np.sort(rng.uniform(0, 20, n))
    y = np.sin(x) + 0.5 * np.sin(0.2 * x) + 0.1 * rng.standard_normal(n)
    prior = {'log(ls)': (np.log(2.), 0.5), 'log(ll)': (np.log(8.), 0.5), 'log(al)': (np.log(0.5), 0.5), 'log(s2)': (np.log(0.02), 0.5)}

    def makegp(hp):
        k = lgp.ExpQuad(scale=hp['ls']) * lgp.Periodic(scale=1.) + lgp.Maternp(p=2, scale=hp['ll']) * hp['al']
        return lgp.GP(k, checkpos=False).addx(x, 'f').addcov(lgp.diag(hp['s2'] * np.ones(n)), 'e').addtransf({'f': 1, 'e': 1}, 'y')

    fit = lgp.empbayes_fit(prior, makegp, {'y': y}, raises=False, minkw=dict(options=dict(maxiter=5)))
    assert fit._autodiff and fit._autodiff_checked
    z = np.array([0.1, -0.2, 0.3, 0.2])
    np.testing.assert_allclose(fit._auto_derivs(z, False)[1], fit._fd_gradient(z), rtol=3e-6, atol=1e-6)


def test_fit_on_derivative_observations_differentiates_itself(lgp, rng):
    """ values and slopes observed, kernel a product with a fitted scale in each factor: the gradient of the likelihood comes
    from the tangents of the DERIVED blocks (scale tangents through `diff`: what jax.jacfwd gives the reference,
    _fit.py:598-616 over _Kernel/_ops.py:115-211), not from central differences """
    n = 60
    x = np.sort(rng.uniform(0, 12, n))
    f, df = np.sin(x) * np.exp(-0.05 * x), (np.cos(x) - 0.05 * np.sin(x)) * np.exp(-0.05 * x)
    yf, ydf = f + 0.05 * rng.standard_normal(n), df + 0.05 * rng.standard_normal(n)
    prior = {'log(ls)': (np.log(1.5), 0.5), 'log(lc)': (np.log(6.), 0.5), 'log(amp)': (0., 0.5)}

    def makegp(hp):
        k = lgp.ExpQuad(scale=hp['ls']) * lgp.Cauchy(alpha=2, beta=3., scale=hp['lc']) * hp['amp']
        return (lgp.GP(k, checkpos=False, checksym=False).addx(x, 'f').addx(x, 'df', deriv=1)
                .addcov({('ef', 'ef'): 0.05 ** 2 * np.eye(n), ('edf', 'edf'): 0.05 ** 2 * np.eye(n), ('ef', 'edf'): np.zeros((n, n))})
                .addtransf({'f': 1, 'ef': 1}, 'yf').addtransf({'df': 1, 'edf': 1}, 'ydf'))

    fit = lgp.empbayes_fit(prior, makegp, {'yf': yf, 'ydf': ydf}, raises=False, minkw=dict(options=dict(maxiter=8)))
    assert fit._autodiff and fit._autodiff_checked
    z = np.array([0.2, -0.1, 0.3])
    np.testing.assert_allclose(fit._auto_derivs(z, False)[1], fit._fd_gradient(z), rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize('a,b', [(1, 0), (1, 1), (2, 1), (2, 2)])
def test_scale_tangents_of_derived_blocks(lgp, rng, a, b):
    """ `gram_tangents(..., deriv=)`: d/dscale and d/d(factor) of derived blocks against central differences in the hyperparameters:
    ExpQuad (closed form, any order up to 4 in all), Cauchy (Taylor arithmetic: the extra order goes to the side that has room),
    and their product (Leibniz) """
    from lsqfitgp_amd._dual import Dual
    x, y = rng.uniform(0, 3, 17), rng.uniform(0, 3, 13)
    cases = {
        'expquad': lambda s, c: lgp.ExpQuad(scale=s) * c,
        'cauchy': lambda s, c: lgp.Cauchy(alpha=2, beta=2.5, scale=s) * c,
        'product': lambda s, c: (lgp.ExpQuad(scale=s) * lgp.Cauchy(alpha=2, beta=2.5, scale=1.7)) * c + 0.3,
    }
    for name, make in cases.items():
        if a == 2 and b == 2:
            # no room for the extra order: the Taylor cores stop at the second per side, ExpQuad's Hermite form at a total of four
            with pytest.raises(NotImplementedError):
                make(Dual(1.3, {'s': 1.}), 0.8).gram_tangents(x, y, deriv=(a, b))
            continue
        s0, c0, h = 1.3, 0.8, 1e-5
        K, tans = make(Dual(s0, {'s': 1.}), Dual(c0, {'c': 1.})).gram_tangents(x, y, deriv=(a, b))
        np.testing.assert_allclose(K.to_host(), make(s0, c0).gram(x, y, deriv=(a, b)).to_host(), rtol=0, atol=1e-13)
        for key, fd in (('s', (make(s0 + h, c0).gram(x, y, deriv=(a, b)).to_host() - make(s0 - h, c0).gram(x, y, deriv=(a, b)).to_host()) / (2 * h)),
                        ('c', (make(s0, c0 + h).gram(x, y, deriv=(a, b)).to_host() - make(s0, c0 - h).gram(x, y, deriv=(a, b)).to_host()) / (2 * h))):
            got = tans[key].to_host()
            np.testing.assert_allclose(got, fd, rtol=0, atol=2e-8 * max(np.abs(fd).max(), 1.), err_msg=f'{name} d/d{key} of the ({a}, {b}) block')


# ---- the reference's tests/test_fit.py on the device classes ------------------------------------------------------------------

def _quad(A, v):
    w, u = np.linalg.eigh(A)
    utv = u.T @ v
    eps = len(A) * 1e-12 * np.max(w)
    return (utv.T / np.maximum(w, eps)) @ utv


def _check_fit(lgp, rng, hyperprior, gpfactory, alpha=1e-5, **fitkw):
    """ ref :61-76: hyperparameters drawn from the prior, data drawn from the GP they define, and the fit must be compatible
    with them (chi-square of the pulls under the fit's own covariance) """
    true = {k: m + s * rng.standard_normal() for k, (m, s) in hyperprior.items()}
    hp = dict(true)
    for k, v in true.items():
        if k.startswith('log('):
            hp[k[4:-1]] = np.exp(v)
    gp = gpfactory(hp)
    cov = gp.prior('x', raw=True)
    n = len(cov)
    data = np.linalg.cholesky(cov + 1e-12 * np.trace(cov) / n * np.eye(n)) @ rng.standard_normal(n)
    fit = lgp.empbayes_fit(hyperprior, gpfactory, {'x': data}, raises=False, **fitkw)
    names = list(hyperprior)
    diff = np.array([fit.pmean[k] - true[k] for k in names])
    C = np.array([[fit.pcov[a, b] for b in names] for a in names])
    q = _quad(C, diff)
    assert stats.chi2(len(names)).sf(q) > alpha / 2
    assert stats.chi2(len(names)).cdf(q) > alpha / 2
    return fit


def test_fit_recovers_scale(lgp, rng):
    """ ref :87-94 """
    x = np.linspace(0, 2 * np.pi * 5, 20)
    _check_fit(lgp, rng, {'log(scale)': (np.log(3), 0.2 / 3)}, lambda hp: lgp.GP(lgp.ExpQuad(scale=hp['scale'])).addx(x, 'x'), alpha=1e-7)


def test_fit_recovers_sdev(lgp, rng):
    """ ref :96-103 """
    x = np.linspace(0, 5, 10)
    _check_fit(lgp, rng, {'log(sdev)': (0., 1.)}, lambda hp: lgp.GP(lgp.ExpQuad() * hp['sdev'] ** 2).addx(x, 'x'), alpha=1e-8)


def test_fit_recovers_scale_and_sdev(lgp, rng):
    """ ref :106-115 """
    x = np.linspace(0, 2 * np.pi * 5, 20)
    fit = _check_fit(lgp, rng, {'log(scale)': (np.log(3), 0.2 / 3), 'log(sdev)': (0., 1.)},
                     lambda hp: lgp.GP(hp['sdev'] ** 2 * lgp.ExpQuad(scale=hp['scale'])).addx(x, 'x'), alpha=1e-7)
    assert fit._autodiff


def _sdev_problem(lgp, rng):
    x = np.linspace(0, 5, 10)
    gpfactory = lambda hp: lgp.GP(lgp.ExpQuad() * hp['sdev'] ** 2).addx(x, 'x')
    true = rng.standard_normal()
    cov = gpfactory({'sdev': np.exp(true)}).prior('x', raw=True)
    data = np.linalg.cholesky(cov + 1e-12 * np.eye(10)) @ rng.standard_normal(10)
    return {'log(sdev)': (0., 1.)}, gpfactory, data, true


def test_fit_methods_agree(lgp, rng):
    """ ref :142-176: Nelder-Mead, BFGS, L-BFGS-B, the Fisher dogleg and trust-constr land on the same mode, with data that are
    fixed and with data that depend on the hyperparameters """
    hp, gpfactory, data_fixed, true = _sdev_problem(lgp, rng)
    data_variable = lambda h: {'x': data_fixed + h['log(sdev)']}
    for data in ({'x': data_fixed}, data_variable):
        kws = [dict(method='nograd', minkw=dict(options=dict(xatol=1e-6))), dict(method='gradient'), dict(method='fisher'),
               dict(method='fisher', minkw=dict(method='trust-constr')), dict(method='gradient', minkw=dict(method='l-bfgs-b'))]
        fits = []
        for kw in kws:
            kw = dict(kw)
            kw.setdefault('minkw', {}).update(x0=np.array([true]))
            fits.append(lgp.empbayes_fit(hp, gpfactory, data, **kw))
        p = fits[0].minresult.x
        for fit in fits[1:]:
            util.assert_allclose(fit.minresult.x, p, atol=1e-5)


def test_fit_checks_and_int_data(lgp):
    """ ref :178-190 """
    with pytest.raises(KeyError):
        lgp.empbayes_fit({'a': (0., 1.)}, lambda hp: None, {'x': 0.}, method='cippa')
    makegp = lambda hp: lgp.GP(lgp.ExpQuad()).addx(hp['a'], 'x')
    with pytest.raises(RuntimeError) as err:
        lgp.empbayes_fit({'a': (0., 1.)}, makegp, {'x': 0.}, minkw=dict(options=dict(maxiter=0)))
    assert 'minimization failed: ' in str(err.value)
    lgp.empbayes_fit({'a': (0., 1.)}, makegp, {'x': 0})


def test_fit_data_formats(lgp, rng):
    """ ref :192-244 (without gvars): dictionaries, tuples with and without a covariance, callables of the hyperparameters """
    hp, gpfactory, data, _ = _sdev_problem(lgp, rng)
    d = {'x': data}
    zerocov = {('x', 'x'): np.zeros((10, 10))}
    err = 0.1
    d_err = {'x': data + err * rng.standard_normal(10)}
    cov = {('x', 'x'): err ** 2 * np.eye(10)}
    for datasets in ([d, (d,), (d, zerocov), lambda _: d, lambda _: (d,), lambda _: (d, zerocov)],
                     [(d_err, cov), lambda _: (d_err, cov)]):
        fits = [lgp.empbayes_fit(hp, gpfactory, ds) for ds in datasets]
        for fit in fits[1:]:
            util.assert_allclose(fit.minresult.x, fits[0].minresult.x, atol=1e-6)


def test_fit_additional_loss(lgp, rng):
    """ ref :246-358: a zero loss changes nothing, a constant shifts the value only, a loss with its minimum elsewhere pulls the
    mode there, and the Fisher matrix refuses a user loss """
    hp, gpfactory, data, _ = _sdev_problem(lgp, rng)
    common = dict(hyperprior=hp, gpfactory=gpfactory, data={'x': data}, minkw=dict(method='bfgs'))
    fit0 = lgp.empbayes_fit(**common)
    util.assert_allclose(lgp.empbayes_fit(**common, additional_loss=lambda _: 0.).minresult.fun, fit0.minresult.fun)
    fit1 = lgp.empbayes_fit(**common, additional_loss=lambda _: 100.)
    util.assert_allclose(fit0.minresult.fun + 100., fit1.minresult.fun, rtol=1e-13)
    util.assert_allclose(fit0.minresult.x, fit1.minresult.x, rtol=1e-9, atol=1e-9)
    loc = fit0.pmean['log(sdev)'] + 10
    fit2 = lgp.empbayes_fit(**common, additional_loss=lambda h: (h['log(sdev)'] - loc) ** 2)
    assert fit2.minresult.fun > fit0.minresult.fun
    assert abs(fit2.pmean['log(sdev)'] - loc) < abs(fit0.pmean['log(sdev)'] - loc)
    with pytest.raises(NotImplementedError):
        lgp.empbayes_fit(hp, gpfactory, {'x': data}, method='fisher', additional_loss=lambda _: 0.)


def test_fit_of_an_autoregressive_coefficient(lgp, rng):
    """ a discrete-time model through the fit driver: y an AR(1) series plus white noise, hyperparameters the coefficient (a bounded
    copula variable) and the noise level.  The coefficient reaches `AR` as a `Dual`: the tangent of the autocovariance table (the
    Yule-Walker system and the recursion differentiated on the host) gives the tangent of the Gram matrix by the same device
    look-up, so the fit differentiates itself; with ``float(hp['phi'])`` the tangent is lost, the first gradient check notices
    and the fit goes on by central differences -- same mode either way, the oracle's too, and where the data say """
    import oracle
    n, phi_true = 300, 0.7
    e = rng.standard_normal(n + 200)
    y = np.zeros(n + 200)
    for i in range(1, len(y)):
        y[i] = phi_true * y[i - 1] + e[i]
    y = y[200:] + 0.3 * rng.standard_normal(n)
    t = np.arange(n)
    prior = {'phi': lgp.copula.uniform(-0.95, 0.95), 'log(s2)': (np.log(0.1), 1.)}
    out = {}
    for name, mod, conv in (('dual', lgp, lambda v: v), ('float', lgp, float), ('oracle', oracle, float)):
        def makegp(hp, mod=mod, conv=conv):
            k = mod.AR(phi=[conv(hp['phi'])], maxlag=n) + mod.White() * hp['s2']
            return mod.GP(k, checkpos=False, checksym=False).addx(t, 'y')
        out[name] = lgp.empbayes_fit(prior, makegp, {'y': y}, raises=False)
    assert out['dual']._autodiff
    z = out['dual'].zmean + 0.1
    np.testing.assert_allclose(out['dual']._fun_and_jac(z)[1] - z, out['dual']._fd_gradient(z), rtol=1e-6, atol=1e-7)
    assert out['float']._autodiff_checked and not out['float']._autodiff
    assert abs(out['dual'].pmean['phi'] - phi_true) < 0.12
    for other in ('float', 'oracle'):
        assert out['dual'].pmean['phi'] == pytest.approx(out[other].pmean['phi'], abs=2e-4)
        assert out['dual'].pmean['log(s2)'] == pytest.approx(out[other].pmean['log(s2)'], abs=5e-3)


def test_tangents_of_the_discrete_time_kernels(lgp, rng):
    """ MA weights and AR coefficients as `Dual`, with normalisation and output operations: against central differences """
    x = rng.integers(0, 40, 60)
    y = rng.integers(0, 40, 25)
    h = 1e-6

    def check(make, keys):
        K, tan = make({k: 0. for k in keys}, True).gram_tangents(x, y)
        assert sorted(tan) == sorted(keys)
        for k in keys:
            up = make({q: (h if q == k else 0.) for q in keys}, False).gram(x, y).to_host()
            dn = make({q: (-h if q == k else 0.) for q in keys}, False).gram(x, y).to_host()
            np.testing.assert_allclose(tan[k].to_host(), (up - dn) / (2 * h), rtol=0, atol=2e-8 * max(1., np.abs(K.to_host()).max()))

    D = lambda v, k, dual: lgp.Dual(v, {k: 1.}) if dual else v
    check(lambda d, dual: lgp.AR(phi=[D(0.5 + d['a'], 'a', dual), D(-0.2 + d['b'], 'b', dual)], maxlag=50) * D(1.5 + d['c'], 'c', dual) + 0.1, ['a', 'b', 'c'])
    check(lambda d, dual: lgp.AR(phi=[D(0.6 + d['a'], 'a', dual)], maxlag=50, norm=True), ['a'])
    check(lambda d, dual: lgp.MA(w=[1., D(0.5 + d['b'], 'b', dual), D(-0.3 + d['c'], 'c', dual)]), ['b', 'c'])
    check(lambda d, dual: lgp.MA(w=[1., D(0.5 + d['b'], 'b', dual), -0.3], norm=True) * 2., ['b'])

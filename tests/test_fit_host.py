"""
CPU tests of the host logic around the hot path's callers: hyperprior parametrisation, copula
transforms, value(sdev) formatting, sample packing -- and the fit driver itself, run here over the CPU ORACLE's GP (the
driver only needs ``gp._prior_decomp`` and ``decomp.minus_log_normal_density``, which the oracle's classes provide too).

Three of these are KNOWN-ANSWER tests against outputs of the reference stored in its user guide
(``docs/userguide/hyper.rst:49-70``, ``:124-156,252-257`` and ``:406-447``): fits of the scale (and amplitude) of an ExpQuad
GP, and of the period of a Periodic one, to sin(x) on 11 points, printed by the reference as ``2.16(17)``,
``log(sdev) 0.89(33)  log(scale) 1.051(77)`` and ``log(period) 1.8373(29)  period 6.280(18)``.  They
pin the oracle's Gram build + regularised Cholesky + marginal likelihood, through their argmax, to numbers produced by
the reference itself.
"""

import numpy as np
import pytest
from scipy import stats

import oracle
from lsqfitgp_amd import copula, UVal
from lsqfitgp_amd._fit import empbayes_fit, _HyperSpace
from lsqfitgp_amd._fastraniter import _flatten


def test_uval_format_is_gvar_compact_notation():
    cases = [((2.1634, 0.171), '2.16(17)'), ((0.8912, 0.334), '0.89(33)'), ((1.0512, 0.0771), '1.051(77)'),
             ((6.2801, 0.0181), '6.280(18)'), ((1.83731, 0.00291), '1.8373(29)'), ((0, 95.2), '0(95)'),
             ((5.86042324, 5.9e-7), '5.86042324(59)'), ((12.3, 1.4), '12.3(1.4)'), ((2.44, 0.81), '2.44(81)')]
    for (m, s), text in cases:
        assert str(UVal(m, s)) == text
    u = UVal(4., 0.4)
    assert abs(u.sqrt().sdev - 0.1) < 1e-15 and abs((3 / u).sdev - 3 * 0.4 / 16) < 1e-15
    assert abs(u.exp().sdev - np.exp(4) * 0.4) < 1e-12 and abs(u.log().sdev - 0.1) < 1e-15


@pytest.mark.parametrize('dist,ref', [
    (copula.beta(2, 1), stats.beta(2, 1)), (copula.beta(0.7, 3.2), stats.beta(0.7, 3.2)),
    (copula.invgamma(1, 1), stats.invgamma(1)), (copula.invgamma(2.5, 0.3), stats.invgamma(2.5, scale=0.3)),
    (copula.gamma(3, 2), stats.gamma(3, scale=0.5)), (copula.halfcauchy(1.5), stats.halfcauchy(scale=1.5)),
    (copula.halfnorm(0.7), stats.halfnorm(scale=0.7)), (copula.uniform(-1, 3), stats.uniform(-1, 4)),
    (copula.lognorm(0.3, 1.2), stats.lognorm(1.2, scale=np.exp(0.3))),
])
def test_copula_transforms(dist, ref):
    z = np.linspace(-5, 5, 41)
    x = dist.invfcn(z)
    # (a bounded variable stored as a double cannot resolve its own tails: a + (b - a) Phi(z) cancels at the edges)
    rtol = 1e-4 if isinstance(dist, copula.uniform) else 1e-9
    np.testing.assert_allclose(ref.cdf(x), stats.norm.cdf(z), rtol=rtol, atol=1e-300)       # x = F^-1(Phi(z))
    if not isinstance(dist, (copula.uniform, copula.beta)):
        np.testing.assert_allclose(ref.sf(x), stats.norm.sf(z), rtol=rtol, atol=1e-300)     # ... accurately in the upper tail too
    np.testing.assert_allclose(dist.fcn(x), z, rtol=1e-7, atol=1e-9)
    h = 1e-6
    zc = z[np.abs(z) <= 3]           # (central differences of a saturating map are noise in the far tails)
    np.testing.assert_allclose(dist.dinvfcn(zc), (dist.invfcn(zc + h) - dist.invfcn(zc - h)) / (2 * h), rtol=2e-5)
    assert np.ndim(dist.invfcn(0.3)) == 0


def test_hyperspace_names_and_slopes():
    prior = {'log(scale)': (np.log(3), 1), 'sqrt(v)': (2., 0.5), 'alpha': copula.beta(2, 1), 'plain': UVal(1., 2.)}
    sp = _HyperSpace(prior, {'v': True}, 'priormean')
    assert sp.nfree == 3
    z = np.array([0.3, -0.2, 1.1])
    hp = sp.unflat(z)
    assert set(hp) == {'log(scale)', 'scale', 'sqrt(v)', 'v', 'alpha', 'plain'}
    assert hp['scale'] == pytest.approx(3 * np.exp(0.3)) and hp['v'] == 4. and hp['plain'] == pytest.approx(1 + 2 * 1.1)
    sl = sp.slopes(z)
    assert 'v' not in sl and 'sqrt(v)' not in sl                         # fixed
    for nm, (i, d) in sl.items():
        e = np.zeros(3); e[i] = 1e-6
        fd = (sp.unflat(z + e)[nm] - sp.unflat(z - e)[nm]) / 2e-6
        assert d == pytest.approx(fd, rel=1e-6)
    with pytest.raises(ValueError):
        _HyperSpace({'log(a)': (0, 1), 'a': (0, 1)}, None, 'priormean')
    with pytest.raises(KeyError):
        _HyperSpace({'a': (0, 1)}, {'b': True}, 'priormean')
    sp2 = _HyperSpace({'log(a)': (0., 2.), 'b': copula.invgamma(1, 1)}, None, {'log(a)': 1., 'b': 0.5})
    assert sp2.unflat(sp2.z0)['a'] == pytest.approx(np.e) and sp2.unflat(sp2.z0)['b'] == pytest.approx(0.5)


def test_raniter_flatten():
    mean = {'a': np.arange(3.), 'b': np.ones((2, 2))}
    cov = {(i, j): np.zeros(np.shape(mean[i]) + np.shape(mean[j])) for i in mean for j in mean}
    cov['a', 'a'] = np.eye(3) * 2
    cov['b', 'b'] = np.eye(4).reshape(2, 2, 2, 2) * 3
    flat, sq, unpack = _flatten(mean, cov)
    assert flat.shape == (7,) and sq.shape == (7, 7) and np.all(np.diag(sq) == [2, 2, 2, 3, 3, 3, 3])
    back = unpack(np.arange(7.))
    assert back['a'].shape == (3,) and back['b'].shape == (2, 2) and back['b'][1, 0] == 5
    flat, sq, unpack = _flatten(2., 9.)
    assert flat.shape == (1,) and unpack(np.array([4.])) == 4.


# ---- known answers from the reference's user guide -----------------------------------------------------------------

X11 = np.linspace(-5, 5, 11)
Y11 = np.sin(X11)


def fit_scale(mod, **kw):
    """ docs/userguide/hyper.rst:49-64 """
    def makegp(hp):
        return mod.GP(mod.ExpQuad(scale=hp['scale'])).addx(X11, 'sine')
    hyperprior = {'log(scale)': (np.log(3), 1)}                   # gvar.log(gvar.gvar(3, 3)): mean log 3, sdev 3 / 3
    return empbayes_fit(hyperprior, makegp, {'sine': Y11}, raises=False, **kw)


def fit_sdev_scale(mod, **kw):
    """ docs/userguide/hyper.rst:124-139 """
    def makegp(hp):
        return mod.GP(mod.ExpQuad(scale=hp['scale']) * hp['sdev'] ** 2).addx(X11, 'sine')
    hyperprior = {'log(sdev)': (0., 1.), 'log(scale)': (np.log(3), 1.)}
    return empbayes_fit(hyperprior, makegp, {'sine': Y11}, raises=False, **kw)


def fit_period(mod, **kw):
    """ docs/userguide/hyper.rst:406-424: the same eleven values of sin(x) under a Periodic kernel whose period is the
    hyperparameter; hyperprior log(period) = log(2 pi gvar(1, 1)), i.e. mean log 2 pi, standard deviation 1 """
    def makegp(hp):
        return mod.GP(mod.Periodic(scale=hp['period'] / (2 * np.pi))).addx(X11, 'sine')
    hyperprior = {'log(period)': (np.log(2 * np.pi), 1.)}
    return empbayes_fit(hyperprior, makegp, {'sine': Y11}, raises=False, **kw)


def check_period(fit):
    """ reference output (``docs/userguide/hyper.rst:441-447``): ``log(period) 1.8373(29)``, ``period 6.280(18)`` -- the mode to the
    printed precision; the standard deviation from the curvature of the objective at the mode (the reference prints its BFGS
    estimate, which is path dependent) """
    assert abs(fit.pmean['log(period)'] - 1.8373) <= 0.00012
    assert abs(fit.pmean['period'] - 6.280) <= 0.0012
    z = fit.zmean
    h = 1e-4
    curv = (fit._fun(z + h) - 2 * fit._fun(z) + fit._fun(z - h)) / h ** 2
    sd = 1 / np.sqrt(curv)                                    # z = log(period) - log(2 pi): unit prior standard deviation
    assert abs(sd - 0.0029) <= 0.0004
    assert str(UVal(fit.pmean['log(period)'], sd)) in ('1.8373(29)', '1.8373(28)', '1.8373(30)', '1.8372(29)', '1.8374(29)')
    assert str(UVal(fit.pmean['period'], fit.pmean['period'] * sd)) in ('6.280(18)', '6.280(17)', '6.280(19)', '6.279(18)', '6.281(18)')


def check_scale(fit):
    """ reference output ``2.16(17)``: the posterior mode to the printed precision; the standard deviation comes from the
    BFGS inverse-Hessian estimate (path dependent, ``covariance='auto'``), so it is compared loosely and, tightly, with the
    curvature of the objective at the mode """
    assert abs(fit.pmean['scale'] - 2.16) <= 0.006
    assert str(fit.p['scale']) == '2.16(17)'
    z = fit.zmean
    h = 1e-3
    curv = (fit._fun(z + h) - 2 * fit._fun(z) + fit._fun(z - h)) / h ** 2
    sd_exact = fit.pmean['scale'] / np.sqrt(curv)
    assert abs(sd_exact - 0.17) <= 0.012
    assert str(UVal(fit.pmean['scale'], sd_exact)) == '2.16(17)'


def check_sdev_scale(fit):
    """ reference outputs ``log(sdev) 0.89(33)``, ``log(scale) 1.051(77)``, ``sdev 2.44(81)``, ``scale 2.86(22)`` and
    P(sdev < 1) = 0.00383 """
    assert abs(fit.pmean['log(sdev)'] - 0.89) <= 0.006
    assert abs(fit.pmean['log(scale)'] - 1.051) <= 0.0012
    assert abs(fit.pmean['sdev'] - 2.44) <= 0.012 and abs(fit.pmean['scale'] - 2.86) <= 0.006
    # exact curvature at the mode instead of the BFGS estimate
    z = fit.zmean
    h = 1e-3
    H = np.empty((2, 2))
    for i in range(2):
        for j in range(2):
            ei = np.eye(2)[i] * h; ej = np.eye(2)[j] * h
            H[i, j] = (fit._fun(z + ei + ej) - fit._fun(z + ei - ej) - fit._fun(z - ei + ej) + fit._fun(z - ei - ej)) / (4 * h * h)
    sd = np.sqrt(np.diag(np.linalg.inv(H)))          # z = (log sdev, (log scale - log 3)): unit prior sdevs
    assert abs(sd[0] - 0.33) <= 0.03 and abs(sd[1] - 0.077) <= 0.008
    prob = stats.norm.cdf(0, loc=fit.pmean['log(sdev)'], scale=sd[0])
    assert 0.001 <= prob <= 0.012                    # reference: 0.00383 with its BFGS standard deviation
    if fit.gradfactory is None and not fit._autodiff:     # finite-difference mode reports the differenced Hessian
        assert str(fit.p['log(sdev)']) in ('0.89(33)', '0.89(32)', '0.89(34)')
        assert str(fit.p['log(scale)']) in ('1.051(77)', '1.051(76)', '1.051(78)')


def test_known_answer_scale_fit_oracle():
    check_scale(fit_scale(oracle))


def test_known_answer_sdev_scale_fit_oracle():
    check_sdev_scale(fit_sdev_scale(oracle))


def test_known_answer_period_fit_oracle():
    check_period(fit_period(oracle))


def test_fit_options_on_oracle():
    f = fit_scale(oracle, method='nograd')
    assert abs(f.pmean['scale'] - 2.16) <= 0.01 and np.isnan(f.p['scale'].sdev)
    g = fit_scale(oracle, fix={'scale': True}, initial={'log(scale)': 0.5})
    assert g.pmean['scale'] == pytest.approx(np.exp(0.5)) and g.p['scale'].sdev == 0
    h = fit_scale(oracle, additional_loss=lambda hp: 50 * (hp['log(scale)'] - 1.0) ** 2)
    assert 2.16 < h.pmean['scale'] < np.e                        # pulled towards exp(1)
    with pytest.raises(NotImplementedError):
        fit_scale(oracle, method='fisher')
    with pytest.raises(KeyError):
        fit_scale(oracle, covariance='what')

"""
The reference's checks of the discrete-time kernels' own arithmetic (``tests/kernels/test_autoreg.py:32-185``): Yule-Walker in both
directions, the recursion, the roots parametrisation (phase degeneracy, real roots written as complex ones, roots at infinity),
AR(0), AR(1), normalisations -- on the device package (table computed on the host, looked up on the GPU) and on the oracle.
"""

import numpy as np
import pytest
from scipy import linalg

from . import util


@pytest.fixture(scope='module', params=[pytest.param('device', marks=pytest.mark.gpu), 'oracle'])
def lgp(request):
    if request.param == 'oracle':
        import oracle
        return oracle
    import lsqfitgp_amd
    lsqfitgp_amd.Context.default()
    return lsqfitgp_amd


def gen_ar_acf(p, rng):
    """ ref :28-38: a positive definite autocovariance at lag 0..p from p random damped oscillations """
    if p:
        mod = rng.uniform(1.1, 10, p)
        phase = rng.uniform(0, 2 * np.pi, p)
        root = mod * np.exp(1j * phase)
        ampl = np.abs(rng.standard_normal(p))
        tau = np.arange(p + 1)
        return np.sum(ampl * root ** -tau[:, None], axis=1).real
    return np.abs(rng.standard_normal(1))


PLIST = [0, 1, 2, 10, 30, 100]


def test_generated_acf_is_positive(rng):
    for p in PLIST:
        acf = gen_ar_acf(p, rng)
        assert acf.ndim == 1 and acf.size == 1 + p
        w = linalg.eigvalsh(linalg.toeplitz(acf))
        assert np.min(w) >= -np.max(w) * (p + 1) * np.finfo(float).eps


def test_yule_walker_there_and_back(lgp, rng):
    """ ref :50-76 """
    for p in PLIST:
        acf = gen_ar_acf(p, rng)
        phi = lgp.AR.phi_from_gamma(acf)
        acf2 = lgp.AR.gamma_from_phi(phi)
        util.assert_allclose(acf2 / acf2[0], acf / acf[0], rtol=1e-12)
        acf3 = lgp.AR.gamma_from_phi(np.pad(phi, (0, 1 + p)))           # trailing zero coefficients change nothing
        util.assert_allclose(acf3[:len(acf2)], acf2, rtol=1e-13)
        acf4 = lgp.AR.extend_gamma(acf3[:1 + p], phi, 1 + p)             # ... and the recursion gives their higher lags
        util.assert_allclose(acf4, acf3, atol=1e-300, rtol=1e-12)
    util.assert_allclose(lgp.AR.gamma_from_phi(np.empty(0)), [1], rtol=1e-15)
    phi = rng.uniform(-1 + 1e-8, 1 - 1e-8)
    util.assert_allclose(lgp.AR.gamma_from_phi([phi]), 1 / ((1 - phi) * (1 + phi)) * phi ** np.arange(2), rtol=1e-15)


def test_phase_degeneracy(lgp):
    """ ref :88-103: the argument of a complex root is defined modulo 2 pi and up to its sign """
    lag = np.arange(100)
    for ph1, ph2 in [([1], [-1]), ([1], [1 + 2 * np.pi]), ([1], [1 - 2 * np.pi]), ([1, -1], [1, 1]), ([0.01], [0.01 + 4 * np.pi]), ([0.01], [0.01 - 4 * np.pi])]:
        c1 = lgp.AR(slnr=[], lnc=0.1 + 1j * np.array(ph1))(0, lag)
        c2 = lgp.AR(slnr=[], lnc=0.1 + 1j * np.array(ph2))(0, lag)
        util.assert_allclose(c2, c1, atol=0, rtol=1e-11)


def test_real_roots_written_as_complex(lgp):
    """ ref :105-113: an exactly real complex root is a pair of identical real roots """
    lag = np.arange(100)
    for r in np.logspace(-1.5, 0, 6):
        for n in range(3):
            for m in range(2):
                add = list(np.arange(1, m + 1) * 0.1 + 1j)
                c1 = lgp.AR(slnr=2 * n * [r], lnc=add)(0, lag)
                c2 = lgp.AR(slnr=[], lnc=add + n * [r])(0, lag)
                util.assert_allclose(c2, c1, atol=0, rtol=1e-7)


def test_ar0_and_ar1(lgp):
    """ ref :115-142 """
    lag = np.arange(100)
    for kw in (dict(phi=[], maxlag=lag.size), dict(gamma=[1], maxlag=lag.size), dict(slnr=[], lnc=[])):
        util.assert_allclose(lgp.AR(**kw)(0, lag), np.where(lag, 0, 1), atol=0, rtol=0)
    for phi in np.logspace(-5, -0.001, 10):
        acf = 1 / ((1 - phi) * (1 + phi)) * phi ** lag
        for kw in (dict(phi=[phi], maxlag=lag.size), dict(gamma=acf[:2], maxlag=lag.size), dict(slnr=[-np.log(phi)], lnc=[])):
            util.assert_allclose(lgp.AR(**kw)(0, lag), acf, atol=1e-300, rtol=1e-12)
            util.assert_allclose(lgp.AR(**kw, norm=True)(0, lag), acf / acf[0], atol=1e-300, rtol=1e-12)


def test_roots_at_infinity(lgp):
    """ ref :154-159: slnr = +-0 is the root +-1 ... of the INVERSE polynomial: exactly the coefficients of (z -+ 1)^p """
    for p in range(1, 10):
        for s in (1, -1):
            np.testing.assert_equal(np.asarray(lgp.AR.phi_from_roots(p * [s * 0.], [])), -np.atleast_1d(np.poly(p * [s]))[1:])


def test_normalisations_and_bad_parameter_sets(lgp):
    """ ref :161-185 """
    x = np.arange(10)[:, None]
    w = [0.1, 0.3, -0.6]
    util.assert_allclose(lgp.MA(w=w)(x, x.T), lgp.MA(w=w, norm=True)(x, x.T) * np.sum(np.square(w)), rtol=1e-15)
    for kw in (dict(phi=[1.4, 0.3], maxlag=100), dict(slnr=[0.1], lnc=[0.2 - 1.j])):
        v1 = lgp.AR(**kw)(x, x.T)
        util.assert_allclose(v1, lgp.AR(**kw, norm=True)(x, x.T) * v1[0, 0], rtol=1e-14)
    kernel = lgp.AR(phi=[1, 2, 3])                     # no maxlag: refused when evaluated
    with pytest.raises(ValueError):
        kernel(np.array([1, 2, 3]), np.array([3, 2, 1]))

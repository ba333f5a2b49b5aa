"""
The special functions behind the Zeta and Color kernels against GOLDEN VALUES THE REFERENCE HOLDS for them
(tests/golden/reference_special/: mpmath values cached by the reference's own tests, ``tests/test_special.py:146-226,415-424``,
copied as data; inputs regenerated as the reference generates them, tests/reference_special.py) -- VERDICT r03 "missing" 3.
CPU part: the oracle's restatements, and the host side of the device's Zeta core (``lsqfitgp_amd/_zeta.py``: the coefficients)
summed here exactly as ``special_core`` of csrc/gram.hip sums them.  The device itself: tests/test_gpu_kernels.py
(TestReferenceSpecialValues).
"""
import numpy as np
import pytest

from oracle import kernels as K
from lsqfitgp_amd import _zeta

from . import reference_special as R

EPS = np.finfo(float).eps


def device_zeta_core(x, nu):
    """ numpy mirror of the LGP_KERNEL_PSERIES branch of special_core (csrc/gram.hip): Re Li_s(e^{2 pi i x}) / zeta(s), s = 1 + 2 nu """
    mode, uselog, table, (A, pw, eps, D) = _zeta.coefficients(nu)
    y = np.abs(np.asarray(x, float)) % 1
    y = np.where(y > 0.5, 1 - y, y)
    th = 2 * np.pi * y
    if mode == _zeta.MODE_INTEGERS:
        return np.where(y == 0, 1., 0.)
    if mode == _zeta.MODE_FOURIER:
        acc = np.zeros(th.shape)
        for j in range(len(table), 0, -1):
            acc += table[j - 1] * np.cos(j * th)
        return acc
    acc = np.zeros(th.shape)
    for n in range(len(table) - 1, -1, -1):
        acc = acc * th * th + table[n]
    with np.errstate(all='ignore'):
        lt = np.log(th)
        gl = (lt if eps == 0 else np.expm1(eps * lt)) if uselog else 1.
        extra = np.where(th > 0, np.exp(pw * lt) * (A * gl + D), (D - A) if (pw == 0 and uselog and eps > 0) else 0.)
    return acc + extra


@pytest.mark.parametrize('case', R.periodic_zeta_cases(), ids=lambda c: c[0])
def test_periodic_zeta_against_the_reference_values(case):
    """ the reference's own tolerance: 150 eps max|F(., s)| per s (``tests/test_special.py:222-226``) """
    cid, x, s, z = case
    for si, zi in zip(s, z):
        want = zi.real
        tol = 150 * EPS * np.max(np.abs(want))
        got = K.periodic_zeta_real(x, float(si))
        assert np.max(np.abs(got - want)) < tol, ('oracle', cid, si)
        # the product's host code: normalised by zeta(s) = F(integer, s)
        zs = want[0]                                           # x[0] = -1
        nu = (float(si) - 1) / 2
        dev = device_zeta_core(x, nu) * zs
        assert np.max(np.abs(dev - want)) < tol, ('device coefficients', cid, si, np.max(np.abs(dev - want)) / (EPS * np.max(np.abs(want))))


def test_expn_against_the_reference_values():
    """ E_n(-ix), n = 2 .. 11: the oracle restates the reference's two-branch evaluation and is held to the tolerance the reference
    holds itself to (rtol 1e-7 of the largest entry, ``tests/test_special.py:415-424`` with ``tests/util.py`` semantics); the device
    kernel (series + continued fraction) is held to 1e-12 in tests/test_gpu_kernels.py """
    n, x, sol = R.expn_case()
    got = np.array([K.expn_imag(int(k), x) for k in n])
    assert np.max(np.abs(got - sol)) <= 1e-7 * np.max(np.abs(sol))
    # what the Color kernel takes from it: (n - 1) Re E_n(-ix)
    assert np.max(np.abs(got.real - sol.real)) <= 1e-7 * np.max(np.abs(sol.real))
    # E_1 through sine / cosine integrals (the reference's exp1_imag): E_n by the recurrence from the golden E_2
    e1 = K.exp1_imag(x)
    e2 = np.exp(1j * x) - (-1j * x) * e1                      # n E_{n+1}(z) = e^-z - z E_n(z), z = -ix
    np.testing.assert_allclose(e2, sol[0], rtol=1e-12)


def test_hurwitz_zeta_against_the_reference_values():
    """ zeta(s, a) + zeta(s, 1 - a), the combination the periodic zeta function is made of (``_special/_zeta.py:133-177``), against
    the reference's mpmath table of zeta(s, a) on s in [-10, 0], a in [0, 1] (tolerance of the reference: 350 eps max, per s) """
    s, a, z = R.hurwitz_case()
    half = a <= 0.5
    for si, zi in zip(s, z):
        if si == 0 or abs(si - round(si)) < 5e-3:
            continue                                           # (s1 = 0 is not reached by s > 1; next to integers: see periodic_zeta_real)
        want = zi[half] + zi[::-1][half]                       # a and 1 - a are both on the grid
        got = K.hurwitz_zeta_pair(float(si), a[half])
        ok = a[half] > 0                                       # zeta(s, 0) is the limit a -> 0 of the table, not of the pair
        assert np.max(np.abs(got[ok] - want[ok])) < 2 * 350 * EPS * np.max(np.abs(zi)), si

"""
Derivative observations (GP.addx(..., deriv=n) -> kernel.linop('diff', ...), src/lsqfitgp/_GP/_elements.py:158-242,565 and
src/lsqfitgp/_Kernel/_ops.py:115-211) on the oracle, pinned by an output of the reference itself: the worked example of
docs/userguide/integrals.rst:34-82 (ExpQuad(scale=2), the primitive of cos from 11 values of its derivative) prints

    area                     ->  -1.9157(27)
    area - true_area         ->  0.0022(27)

i.e. a posterior mean difference and its standard deviation from the posterior covariance: Gram blocks of the kernel and of its
derivatives, the jittered Cholesky and the posterior formulas all on the line.
"""

import numpy as np
import pytest

import oracle
from lsqfitgp_amd import UVal


def integral_example(mod, **kw):
    x = np.linspace(-5, 5, 11)
    y = np.cos(x)
    xplot = np.linspace(-5, 5, 200)
    gp = mod.GP(mod.ExpQuad(scale=2), **kw).addx(xplot, 'primitive').addx(x, 'cosine', deriv=1)
    mean, cov = gp.predfromdata({'cosine': y}, 'primitive', raw=True)
    area = mean[-1] - mean[0]
    sdev = np.sqrt(cov[-1, -1] + cov[0, 0] - 2 * cov[-1, 0])
    true_area = np.sin(xplot[-1]) - np.sin(xplot[0])
    return area, sdev, true_area


def integral_by_transformation(mod):
    """ the "not obvious way" of integrals.rst:84-91: the difference between the ends as a linear transformation of the process
    values (there `addlintransf(lambda x: x[-1] - x[0], ...)`; here the same functional as the coefficient tensor of `addtransf`) """
    x = np.linspace(-5, 5, 11)
    xplot = np.linspace(-5, 5, 200)
    coeff = np.zeros(200)
    coeff[-1], coeff[0] = 1, -1
    gp = (mod.GP(mod.ExpQuad(scale=2)).addx(xplot, 'primitive').addx(x, 'cosine', deriv=1)
          .addtransf({'primitive': coeff}, 'integral'))
    mean, cov = gp.predfromdata({'cosine': np.cos(x)}, 'integral', raw=True)
    return float(np.reshape(mean, ())), float(np.sqrt(np.reshape(cov, ())))


def test_reference_user_guide_integral_on_the_oracle():
    assert str(UVal(*integral_by_transformation(oracle))) == '-1.9157(27)'      # integrals.rst:91
    area, sdev, true_area = integral_example(oracle)
    assert str(UVal(area, sdev)) == '-1.9157(27)'                     # integrals.rst:76
    assert repr(float(true_area)) == '-1.917848549326277'             # integrals.rst:82
    assert str(UVal(area - true_area, sdev)) == '0.0022(27)'          # integrals.rst:82


@pytest.mark.parametrize('a,b', [(1, 0), (0, 1), (1, 1), (2, 0), (0, 2), (2, 1), (2, 2)])
def test_oracle_derivative_kernel_against_differences(a, b):
    """ the closed form of the derived ExpQuad kernel against central differences of the kernel one order below """
    rng = np.random.default_rng(a * 10 + b)
    k = 1.7 * oracle.ExpQuad(scale=1.3, loc=0.4) + 0.25
    x = rng.uniform(-2, 2, 7)
    y = rng.uniform(-2, 2, 5)
    got = k.diff(a, b)(x[:, None], y[None, :])
    h = 1e-5
    if a > 0:
        lo = k.diff(a - 1, b)
        want = (lo(x[:, None] + h, y[None, :]) - lo(x[:, None] - h, y[None, :])) / (2 * h)
    else:
        lo = k.diff(a, b - 1)
        want = (lo(x[:, None], y[None, :] + h) - lo(x[:, None], y[None, :] - h)) / (2 * h)
    np.testing.assert_allclose(got, want, rtol=1e-7, atol=1e-8)


def test_oracle_derivative_blocks_are_consistent():
    """ cov(f', f) is the transpose of cov(f, f') and the joint matrix of (f, f') is positive semidefinite """
    x = np.linspace(-1, 1, 9)
    gp = oracle.GP(oracle.ExpQuad(scale=0.7)).addx(x, 'f').addx(x, 'df', deriv=1)
    prior = gp.prior(raw=True)
    np.testing.assert_allclose(prior['f', 'df'], prior['df', 'f'].T, rtol=0, atol=1e-15)
    K = np.block([[prior['f', 'f'], prior['f', 'df']], [prior['df', 'f'], prior['df', 'df']]])
    assert np.linalg.eigvalsh(K).min() > -1e-12 * np.linalg.eigvalsh(K).max()


def test_derivatives_only_where_there_is_a_closed_form():
    with pytest.raises(NotImplementedError):
        oracle.GammaExp(gamma=1.3).diff(1, 0)
    with pytest.raises(ValueError):
        oracle.Maternp(p=1).diff(2, 0)          # derivable p times on each side (_kernels/_matern.py:26-27)


@pytest.mark.parametrize('p,a,b', [(1, 1, 0), (1, 0, 1), (1, 1, 1), (2, 1, 0), (2, 2, 0), (2, 1, 1), (2, 2, 2), (3, 3, 2), (3, 0, 3)])
def test_oracle_maternp_derivative_against_differences(p, a, b):
    """ the closed form of the derived Maternp kernel (Leibniz on exp(-t) P_p(t)) against central differences of the kernel one
    order below, away from the kink of |u| at zero and on it """
    rng = np.random.default_rng(100 * p + 10 * a + b)
    k = 1.3 * oracle.Maternp(p=p, scale=0.8, loc=0.1) + 0.5
    x = rng.uniform(-2, 2, 7)
    y = rng.uniform(-2, 2, 5)
    y[0] = x[0]                                 # one coincident pair
    got = k.diff(a, b)(x[:, None], y[None, :])
    h = 1e-5
    if a > 0:
        lo = k.diff(a - 1, b)
        want = (lo(x[:, None] + h, y[None, :]) - lo(x[:, None] - h, y[None, :])) / (2 * h)
    else:
        lo = k.diff(a, b - 1)
        want = (lo(x[:, None], y[None, :] + h) - lo(x[:, None], y[None, :] - h)) / (2 * h)
    np.testing.assert_allclose(got[1:], want[1:], rtol=2e-6, atol=2e-6 * np.abs(want).max())
    # at the coincident pair the derivative of order 2p is where the kernel stops being smooth: the central difference of the
    # order below is first-order accurate there
    np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=1e-4 * np.abs(want).max())


@pytest.mark.parametrize('dx,dy', [({'a': 1}, {}), ({'b': 2}, {'a': 1}), ({'a': 1, 'b': 1}, {'b': 1}), ({}, {'c': 2}), ({'a': 2, 'c': 1}, {'a': 2})])
def test_oracle_named_field_derivatives_against_differences(dx, dy):
    """ ExpQuad derived along named fields of structured points: the product of Hermite factors against central differences """
    rng = np.random.default_rng(7)
    dt = [('a', float), ('b', float), ('c', float)]
    x = np.empty(6, dt); y = np.empty(4, dt)
    for f in 'abc':
        x[f] = rng.uniform(-1.5, 1.5, 6); y[f] = rng.uniform(-1.5, 1.5, 4)
    k = 0.9 * oracle.ExpQuad(scale=1.2)
    got = k.diff(dx, dy)(x[:, None], y[None, :])
    side, dv = ('x', dx) if dx else ('y', dy)
    f = next(iter(dv))
    lower = dict(dv); lower[f] -= 1
    lo = k.diff(lower, dy) if side == 'x' else k.diff(dx, lower)
    h = 1e-5

    def shifted(arr, field, step):
        out = arr.copy(); out[field] = out[field] + step
        return out
    if side == 'x':
        want = (lo(shifted(x, f, h)[:, None], y[None, :]) - lo(shifted(x, f, -h)[:, None], y[None, :])) / (2 * h)
    else:
        want = (lo(x[:, None], shifted(y, f, h)[None, :]) - lo(x[:, None], shifted(y, f, -h)[None, :])) / (2 * h)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-8)


def test_oracle_addlintransf_equals_addtransf():
    """ a linear map given as a function (_elements.py:348-406) against the same map given as tensors (:248-346) """
    rng = np.random.default_rng(3)
    x = np.linspace(0, 3, 7)
    A = rng.standard_normal((4, 7)); B = rng.standard_normal((4, 5))
    base = oracle.GP(oracle.ExpQuad(), checkpos=False).addx(x, 'a').addx(x[:5] + 0.3, 'b')
    g1 = base.addtransf({'a': A, 'b': B}, 't')
    g2 = base.addlintransf(lambda u, v: A @ u + B @ v, ['a', 'b'], 't')
    p1, p2 = g1.prior(['t', 'a'], raw=True), g2.prior(['t', 'a'], raw=True)
    for key in p1:
        np.testing.assert_allclose(p2[key], p1[key], rtol=1e-13, atol=1e-14)

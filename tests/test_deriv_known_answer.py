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
        oracle.Maternp(p=2).diff(1, 0)

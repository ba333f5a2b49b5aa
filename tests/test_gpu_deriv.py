"""
GPU parity of derivative observations (GP.addx(..., deriv=n), src/lsqfitgp/_GP/_elements.py:158-242,565 over
src/lsqfitgp/_Kernel/_ops.py:115-211): the derived ExpQuad Gram blocks from the device functor (LGP_KERNEL_EXPQUAD_DIFF)
against the oracle, the posterior against the oracle, and the output the reference prints for this case in
docs/userguide/integrals.rst:76,82 reproduced by the device path.
"""

import numpy as np
import pytest

import oracle
from . import util
from .test_deriv_known_answer import integral_example, integral_by_transformation

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lgp():
    import lsqfitgp_amd
    lsqfitgp_amd.Context.default()
    return lsqfitgp_amd


@pytest.mark.parametrize('a,b', [(1, 0), (0, 1), (1, 1), (2, 0), (0, 2), (2, 1), (1, 2), (2, 2)])
@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_derived_gram_vs_oracle(lgp, a, b, dtype):
    rng = np.random.default_rng(100 + 10 * a + b)
    x = rng.uniform(-3, 3, 300).astype(dtype)
    y = rng.uniform(-3, 3, 777).astype(dtype)
    kd = 1.7 * lgp.ExpQuad(scale=1.3, loc=0.4) + 0.25
    ko = 1.7 * oracle.ExpQuad(scale=1.3, loc=0.4) + 0.25
    got = kd.gram(x, y, deriv=(a, b)).to_host()
    want = ko.diff(a, b)(x.astype(float)[:, None], y.astype(float)[None, :])
    assert got.dtype == dtype
    # He_p(u) exp(-u²/2) is O(1); float32 points move u by 1e-7 relative
    tol = 5e-15 if dtype == np.float64 else 2e-5
    np.testing.assert_allclose(got, want, rtol=0, atol=tol * np.abs(want).max())


def test_reference_user_guide_integral_on_the_device(lgp):
    area, sdev, true_area = integral_example(lgp)
    assert str(lgp.UVal(area, sdev)) == '-1.9157(27)'                  # integrals.rst:76
    assert str(lgp.UVal(area - true_area, sdev)) == '0.0022(27)'       # integrals.rst:82
    assert str(lgp.UVal(*integral_by_transformation(lgp))) == '-1.9157(27)'      # integrals.rst:91
    ao, so, _ = integral_example(oracle)
    assert abs(area - ao) <= 1e-9 and abs(sdev - so) <= 1e-9 * so + 1e-12


def test_posterior_with_derivative_data_vs_oracle(lgp):
    """ values and first derivatives observed together, predictions of both """
    rng = np.random.default_rng(5)
    x = np.sort(rng.uniform(0, 6, 40))
    xs = np.linspace(0, 6, 25)
    y = np.sin(x) + 0.05 * rng.standard_normal(40)
    dy = np.cos(x) + 0.05 * rng.standard_normal(40)
    z = np.zeros((40, 40))
    gc = {('f', 'f'): 0.05 ** 2 * np.eye(40), ('df', 'df'): 0.05 ** 2 * np.eye(40), ('f', 'df'): z, ('df', 'f'): z}

    def run(mod):
        gp = (mod.GP(0.8 * mod.ExpQuad(scale=1.1), checkpos=False)
              .addx(x, 'f').addx(x, 'df', deriv=1).addx(xs, 'fs').addx(xs, 'dfs', deriv=1))
        m, c = gp.predfromdata({'f': y, 'df': dy}, ['fs', 'dfs'], gc, raw=True)
        return m, c, gp.marginal_likelihood({'f': y, 'df': dy}, gc)

    md, cd, ld = run(lgp)
    mo, co, lo = run(oracle)
    for k in ('fs', 'dfs'):
        util.assert_close_matrices(md[k], mo[k], rtol=1e-8)
        for k2 in ('fs', 'dfs'):
            util.assert_close_matrices(cd[k, k2], co[k, k2], rtol=0, atol=1e-8 * np.abs(co['fs', 'fs']).max())
    assert abs(ld - lo) <= 1e-8 * abs(lo)


def test_what_is_not_available_says_so(lgp):
    x = np.linspace(0, 1, 5)
    with pytest.raises(NotImplementedError):
        lgp.GP(lgp.Maternp(p=2)).addx(x, 'a', deriv=1).prior('a', raw=True)
    with pytest.raises(NotImplementedError):
        lgp.GP(lgp.ExpQuad()).addx(x, 'a', deriv='f0')
    xs = np.zeros(5, [('f0', float), ('f1', float)])
    with pytest.raises(NotImplementedError):
        lgp.GP(lgp.ExpQuad()).addx(xs, 'a', deriv=1)
    with pytest.raises(ValueError):
        lgp.GP(lgp.ExpQuad()).addx(x, 'a', deriv=-1)

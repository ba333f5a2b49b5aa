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
    with pytest.raises(ValueError):                                             # not derivable: the reference's error (_Kernel/_ops.py:85-90)
        lgp.GP(lgp.GammaExp(gamma=1.2)).addx(x, 'a', deriv=1).prior('a', raw=True)
    with pytest.raises(NotImplementedError):                                     # no derivable flag in the reference either (_basic.py:156): not built
        lgp.GP(lgp.Gibbs()).addx(x, 'a', deriv=1).prior('a', raw=True)
    with pytest.raises(ValueError):
        lgp.GP(lgp.Zeta(nu=0.7)).addx(x, 'a', deriv=1).prior('a', raw=True)
    with pytest.raises(ValueError):                                             # ceil(nu) - 1 times derivable (_kernels/_matern.py:51-53)
        lgp.GP(lgp.Matern(nu=1.7)).addx(x, 'a', deriv=2).prior('a', raw=True)
    with pytest.raises(ValueError):
        lgp.GP(lgp.Matern(nu=0.8)).addx(x, 'a', deriv=1).prior('a', raw=True)
    with pytest.raises(ValueError):
        lgp.GP(lgp.Maternp(p=1)).addx(x, 'a', deriv=2).prior('a', raw=True)     # derivable p times on each side
    with pytest.raises(ValueError):
        lgp.GP(lgp.ExpQuad()).addx(x, 'a', deriv='f0')                           # a named field of points without fields
    xs = np.zeros(5, [('f0', float), ('f1', float)])
    with pytest.raises(ValueError):
        lgp.GP(lgp.ExpQuad()).addx(xs, 'a', deriv=1)                             # an implicit variable of points with fields
    with pytest.raises(ValueError):
        lgp.GP(lgp.ExpQuad()).addx(xs, 'a', deriv='nope')
    with pytest.raises(ValueError):
        lgp.GP(lgp.ExpQuad()).addx(x, 'a', deriv=-1)
    with pytest.raises(ValueError):
        lgp.GP(lgp.ExpQuad()).addx(xs, 'a', deriv=[2])                           # dangling order (_Deriv.py:88-89)


@pytest.mark.parametrize('p,a,b', [(1, 1, 0), (1, 1, 1), (2, 1, 0), (2, 0, 2), (2, 2, 1), (2, 2, 2), (3, 3, 3), (5, 2, 4)])
@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_derived_maternp_gram_vs_oracle(lgp, p, a, b, dtype):
    """ Maternp is derivable p times on each side (_kernels/_matern.py:26-49): the device functor LGP_KERNEL_MATERNP_DIFF against the
    oracle's closed form (itself checked against central differences on the CPU) """
    rng = np.random.default_rng(1000 * p + 10 * a + b)
    x = rng.uniform(-3, 3, 200).astype(dtype)
    y = rng.uniform(-3, 3, 333).astype(dtype)
    y[:50] = x[:50]                                            # coincident points: the odd derivatives vanish there
    kd = 1.3 * lgp.Maternp(p=p, scale=0.8, loc=0.1) + 0.5
    ko = 1.3 * oracle.Maternp(p=p, scale=0.8, loc=0.1) + 0.5
    got = kd.gram(x, y, deriv=(a, b)).to_host()
    want = ko.diff(a, b)(x.astype(float)[:, None], y.astype(float)[None, :])
    assert got.dtype == dtype
    tol = 1e-13 if dtype == np.float64 else 5e-5
    np.testing.assert_allclose(got, want, rtol=0, atol=tol * np.abs(want).max())


@pytest.mark.parametrize('dx,dy', [('a', None), ((2, 'b'), 'a'), (('a', 'b'), 'b'), (None, (2, 'c')), ((2, 'a', 'c'), (2, 'a'))])
def test_derivatives_along_named_fields_vs_oracle(lgp, dx, dy):
    """ GP.addx(..., deriv=<fields>) on structured points (what Deriv of the reference accepts, _Deriv.py:24-97): prior blocks of
    the derived ExpQuad process against the oracle """
    rng = np.random.default_rng(11)
    dt = [('a', float), ('b', float), ('c', float)]
    x = np.empty(40, dt); y = np.empty(31, dt)
    for f in 'abc':
        x[f] = rng.uniform(-1.5, 1.5, 40); y[f] = rng.uniform(-1.5, 1.5, 31)

    def run(mod):
        gp = mod.GP(0.9 * mod.ExpQuad(scale=1.2), checkpos=False, checksym=False).addx(x, 'u', deriv=dx).addx(y, 'v', deriv=dy).addx(x, 'w')
        return gp.prior(['u', 'v', 'w'], raw=True)
    pd, po = run(lgp), run(oracle)
    for key in po:
        np.testing.assert_allclose(np.asarray(pd[key]), po[key], rtol=0, atol=2e-14 * max(1., np.abs(po[key]).max()))


def test_posterior_with_gradient_observations_2d(lgp):
    """ a function of two variables observed through both partial derivatives (named fields), predicted itself: device vs oracle """
    rng = np.random.default_rng(2)
    dt = [('s', float), ('t', float)]
    x = np.empty(60, dt); xs = np.empty(20, dt)
    x['s'], x['t'] = rng.uniform(0, 3, 60), rng.uniform(0, 3, 60)
    xs['s'], xs['t'] = rng.uniform(0, 3, 20), rng.uniform(0, 3, 20)
    f0 = np.sin(x['s']) * np.cos(x['t'])
    fs, ft = np.cos(x['s']) * np.cos(x['t']), -np.sin(x['s']) * np.sin(x['t'])
    noise = 1e-4 * np.eye(60)
    z = np.zeros((60, 60))
    keys = ['f', 'ds', 'dt']
    gc = {(a, b): (noise if a == b else z) for a in keys for b in keys}

    def run(mod):
        gp = (mod.GP(mod.ExpQuad(scale=1.5), checkpos=False).addx(x, 'f').addx(x, 'ds', deriv='s').addx(x, 'dt', deriv='t').addx(xs, 'out'))
        m, c = gp.predfromdata({'f': f0, 'ds': fs, 'dt': ft}, 'out', gc, raw=True)
        return m, c
    md, cd = run(lgp)
    mo, co = run(oracle)
    util.assert_close_matrices(md, mo, rtol=1e-7)
    util.assert_close_matrices(cd, co, rtol=0, atol=1e-8)
    assert np.max(np.abs(md - np.sin(xs['s']) * np.cos(xs['t']))) < 5e-3


def test_addlintransf_on_the_device(lgp):
    """ GP.addlintransf (_elements.py:348-406): a linear map given as a function -- here a finite-difference second derivative and
    a weighted sum of two processes -- evaluated once on the host and applied to the device blocks; against the oracle, against
    the same map through addtransf, and the linearity check """
    rng = np.random.default_rng(4)
    x = np.linspace(0, 3, 50)
    A = rng.standard_normal((6, 50)); B = rng.standard_normal((6, 20))

    def second_difference(u):
        return u[2:] - 2 * u[1:-1] + u[:-2]

    def run(mod):
        gp = mod.GP(mod.ExpQuad(scale=0.7), checkpos=False).addx(x, 'a').addx(x[:20] + 0.01, 'b')
        gp = gp.addlintransf(second_difference, ['a'], 'd2').addlintransf(lambda u, v: A @ u + B @ v, ['a', 'b'], 'mix')
        gp = gp.addlintransf(lambda u: np.sum(u), ['a'], 'total')
        return gp, gp.prior(['d2', 'mix', 'total', 'a'], raw=True)
    gd, pd = run(lgp)
    go, po = run(oracle)
    for key in po:
        np.testing.assert_allclose(np.asarray(pd[key]), po[key], rtol=0, atol=1e-12 * max(1., np.abs(po[key]).max()))
    ref = lgp.GP(lgp.ExpQuad(scale=0.7), checkpos=False).addx(x, 'a').addx(x[:20] + 0.01, 'b').addtransf({'a': A, 'b': B}, 'mix')
    np.testing.assert_allclose(np.asarray(pd['mix', 'mix']), np.asarray(ref.prior('mix', raw=True)), rtol=0, atol=1e-12)
    y = rng.standard_normal(48)
    md, cd = gd.predfromdata({'d2': y}, 'total', {('d2', 'd2'): 0.1 * np.eye(48)}, raw=True)
    mo, co = go.predfromdata({'d2': y}, 'total', {('d2', 'd2'): 0.1 * np.eye(48)}, raw=True)
    assert abs(md - mo) <= 1e-8 * abs(mo) + 1e-10 and abs(cd - co) <= 1e-8 * abs(co)
    with pytest.raises(RuntimeError):
        gd.addlintransf(lambda u: u + 1, ['a'], 'affine')
    with pytest.raises(RuntimeError):
        gd.addlintransf(lambda u: u ** 2, ['a'], 'square')
    gd.addlintransf(lambda u: u + 1, ['a'], 'affine', checklin=False)
    with pytest.raises(KeyError):
        gd.addlintransf(lambda u: u, ['zzz'], 'q')


# ---- derivatives of the other derivable kernels (VERDICT r03 "missing" 2): the device evaluates the core in truncated Taylor
# arithmetic (csrc/gram_jet.hip), the oracle differentiates its expression symbolically and evaluates in extended precision
# (oracle/symdiff.py); nothing is shared between the two.

def _jet_cases(mod):
    return {
        'periodic': mod.Periodic(outerscale=0.8, scale=1.3, loc=0.2) * 1.4 + 0.3,
        'cos': mod.Cos(scale=0.7) * 0.9,
        'sinc': mod.Sinc(scale=1.1),
        'celerite': mod.Celerite(gamma=0.6, B=0.6, scale=1.5) * 2.,
        'harmonic_q03': mod.Harmonic(Q=0.3), 'harmonic_q08': mod.Harmonic(Q=0.8, scale=0.9), 'harmonic_q1': mod.Harmonic(Q=1),
        'harmonic_q17': mod.Harmonic(Q=1.7, scale=1.2),
        'circular': mod.Circular(tau=5, c=0.4, scale=2.),
        'wendland_k1': mod.Wendland(k=1, alpha=1.2, scale=2.5), 'wendland_k2': mod.Wendland(k=2, alpha=1.5, scale=2.5) * 1.1,
        'decaying': mod.Decaying(alpha=1.5, scale=1.3),
        'wienerintegral': mod.WienerIntegral(scale=0.8),
        'cauchy2': mod.Cauchy(alpha=2, beta=3., scale=1.2) * 1.5,
        'gammaexp2': mod.GammaExp(gamma=2, scale=0.9),
        'constant': mod.Constant() * 2.,
        'fracbrownian11': mod.FracBrownian(H=1, K=1),
        'stationaryfracbrownian1': mod.StationaryFracBrownian(H=1),
    }


# kernel -> how often the PROCESS is derivable (the reference's `derivable=`): orders above it must be refused
_ONCE = {'celerite', 'harmonic_q03', 'harmonic_q08', 'harmonic_q1', 'harmonic_q17', 'circular', 'wendland_k1', 'wienerintegral'}


@pytest.mark.parametrize('name', sorted(_jet_cases(oracle)))
def test_derived_gram_of_the_other_kernels_vs_oracle(lgp, name):
    rng = np.random.default_rng(sum(map(ord, name)))
    x = rng.uniform(0.1, 3, 23)
    y = rng.uniform(0.1, 3, 31)
    y[:6] = x[:6]                                              # coincident points: the diagonal of a derivative block
    kd, ko = _jet_cases(lgp)[name], _jet_cases(oracle)[name]
    # order 0 through the derivative core too: it restates the cores of csrc/gram.hip in Taylor arithmetic
    orders = [(1, 0), (0, 1), (1, 1)] + ([] if name in _ONCE else [(2, 0), (2, 1), (1, 2), (2, 2)])
    for a, b in orders:
        got = kd.gram(x, y, deriv=(a, b)).to_host()
        want = ko.diff(a, b)(x[:, None], y[None, :])
        scale = max(np.abs(want).max(), 1e-3)
        # (coincident points: the oracle takes the limit from an offset of 1e-10, 1e-12 at removable singularities)
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-9 * scale, err_msg=f'{name} {(a, b)}')
        far = np.abs(x[:, None] - y[None, :]) > 1e-6
        np.testing.assert_allclose(got[far], want[far], rtol=0, atol=1e-12 * scale, err_msg=f'{name} {(a, b)} away from coincidences')
    if name in _ONCE:
        with pytest.raises(ValueError):
            kd.gram(x, y, deriv=(2, 0))                        # _Kernel/_ops.py:85-90: more derivatives than the process has


@pytest.mark.parametrize('p', [1, 2, 3])
def test_maternp_derivatives_along_named_fields_vs_oracle(lgp, p):
    """ Maternp over several fields, derived along named ones (one per side): `lgp_gram_deriv` with the host-derived Laurent
    polynomials R_k against the oracle's symbolic route, coincident points included """
    rng = np.random.default_rng(40 + p)
    dt = [('a', float), ('b', float), ('c', float)]
    x = np.empty(17, dt); y = np.empty(13, dt)
    for f in 'abc':
        x[f] = rng.uniform(0, 2, 17); y[f] = rng.uniform(0, 2, 13)
    y[:4] = x[:4]
    kd, ko = lgp.Maternp(p=p, scale=1.3) * 1.2, oracle.Maternp(p=p, scale=1.3) * 1.2
    specs = [(('a', 1), ('b', 1)), (('a', 1), ('a', 1)), (('c', 1), None)]
    if p >= 2:
        specs += [(('a', 2), ('a', 2)), (('b', 2), ('c', 1)), (None, ('b', 2))]
    for sx, sy in specs:
        names = x.dtype.names
        ox = tuple((sx[1] if sx and nm == sx[0] else 0) for nm in names)
        oy = tuple((sy[1] if sy and nm == sy[0] else 0) for nm in names)
        got = kd.gram(x, y, deriv=(ox, oy)).to_host()
        want = ko.diff({sx[0]: sx[1]} if sx else 0, {sy[0]: sy[1]} if sy else 0)(x[:, None], y[None, :])
        scale = max(np.abs(want).max(), 1e-3)
        np.testing.assert_allclose(got, want, rtol=0, atol=3e-9 * scale, err_msg=f'p={p} {sx} {sy}')
    with pytest.raises(ValueError):
        kd.gram(x, y, deriv=((p + 1, 0, 0), (0, 0, 0)))             # derivable p times per side (_kernels/_matern.py:26-27)


@pytest.mark.parametrize('nu', [1.2, 2.3, 3.75])
def test_matern_of_real_order_derivatives_vs_oracle(lgp, nu):
    """ Matern(nu), ceil(nu) - 1 times derivable (_kernels/_matern.py:51-76): the derivative core evaluates K of the orders nu - k on
    the device (csrc/bessel_k_dev.h) and uses the exact limits at coincident points; the oracle differentiates K_nu symbolically and
    evaluates in extended precision.  One- and three-dimensional points. """
    rng = np.random.default_rng(int(100 * nu))
    x = rng.uniform(0.1, 3, 7); y = rng.uniform(0.1, 3, 6)            # (the oracle evaluates K_nu in 120-400 digits)
    y[:2] = x[:2]
    y[4] = x[4] + 1e-7; y[5] = x[5] - 3e-5                       # nearly coincident: large negative orders times small powers
    kd, ko = lgp.Matern(nu=nu, scale=1.3, loc=0.1) * 1.2, oracle.Matern(nu=nu, scale=1.3, loc=0.1) * 1.2
    deg = int(np.ceil(nu) - 1)
    orders = [(a, b) for a in range(min(deg, 2) + 1) for b in range(min(deg, 2) + 1) if a + b]
    for a, b in orders:
        got = kd.gram(x, y, deriv=(a, b)).to_host()
        want = ko.diff(a, b)(x[:, None], y[None, :])
        scale = max(np.abs(want).max(), 1e-3)
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-9 * scale, err_msg=f'nu={nu} {(a, b)}')
        far = np.abs(x[:, None] - y[None, :]) > 1e-3
        np.testing.assert_allclose(got[far], want[far], rtol=0, atol=1e-11 * scale, err_msg=f'nu={nu} {(a, b)} away from coincidences')
    with pytest.raises(ValueError):
        kd.gram(x, y, deriv=(deg + 1, 0))
    # along named fields of three-dimensional points
    dt = [('a', float), ('b', float), ('c', float)]
    X = np.empty(6, dt); Y = np.empty(5, dt)
    for f in 'abc':
        X[f] = rng.uniform(0, 2, 6); Y[f] = rng.uniform(0, 2, 5)
    Y[:2] = X[:2]
    for sx, sy in [(('a', 1), ('b', 1)), (('c', 1), ('c', 1))] + ([(('b', 2), ('b', 2))] if deg >= 2 else []):
        ox = tuple((sx[1] if nm == sx[0] else 0) for nm in X.dtype.names)
        oy = tuple((sy[1] if nm == sy[0] else 0) for nm in X.dtype.names)
        got = kd.gram(X, Y, deriv=(ox, oy)).to_host()
        want = ko.diff({sx[0]: sx[1]}, {sy[0]: sy[1]})(X[:, None], Y[None, :])
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-9 * max(np.abs(want).max(), 1e-3), err_msg=f'nu={nu} {sx} {sy}')


@pytest.mark.parametrize('nu', [2.5, 4.3, 6.])
def test_bessel_kernel_derivatives_vs_oracle(lgp, nu):
    """ Bessel(nu), int(nu // 2) times derivable (_kernels/_matern.py:92-119): G^(k)(r^2) = (-s^2/4)^k / ((nu+1)..(nu+k)) k_{nu+k}(s r), the
    J of the five orders evaluated on the device; the oracle differentiates J_nu symbolically """
    rng = np.random.default_rng(int(10 * nu))
    x = rng.uniform(0.1, 3, 13); y = rng.uniform(0.1, 3, 11)
    y[:4] = x[:4]
    y[4] = x[4] + 1e-7
    x[5] = 9.5; y[5] = 0.2                                       # (argument s r ~ 40: the recurrence / the asymptotic region)
    kd, ko = lgp.Bessel(nu=nu, scale=1.3) * 1.2, oracle.Bessel(nu=nu, scale=1.3) * 1.2
    deg = int(nu // 2)
    orders = [(a, b) for a in range(min(deg, 2) + 1) for b in range(min(deg, 2) + 1) if a + b]
    for a, b in orders:
        got = kd.gram(x, y, deriv=(a, b)).to_host()
        want = ko.diff(a, b)(x[:, None], y[None, :])
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-11 * max(np.abs(want).max(), 1e-3), err_msg=f'nu={nu} {(a, b)}')
    with pytest.raises(ValueError):
        kd.gram(x, y, deriv=(deg + 1, 0))
    s2 = (2 + nu / 2) ** 2 / 1.3 ** 2
    got = kd.gram(x, y, deriv=(1, 1)).to_host()
    assert abs(got[0, 0] - 1.2 * 2 * s2 / (4 * (nu + 1))) <= 1e-12 * got[0, 0]          # the exact value at a coincidence
    dt = [('a', float), ('b', float)]
    X = np.empty(9, dt); Y = np.empty(7, dt)
    for f in 'ab':
        X[f] = rng.uniform(0, 2, 9); Y[f] = rng.uniform(0, 2, 7)
    Y[:3] = X[:3]
    got = kd.gram(X, Y, deriv=((1, 0), (0, 1))).to_host()
    want = ko.diff({'a': 1}, {'b': 1})(X[:, None], Y[None, :])
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-11 * np.abs(want).max())


@pytest.mark.parametrize('n', [4, 5, 6])
def test_color_kernel_derivatives_vs_oracle(lgp, n):
    """ Color(n) = (n - 1) Re E_n(-i |delta|), n // 2 - 1 times derivable (_kernels/_spectral.py:70-111): d^m/dx^m E_n(-i x) =
    i^m E_{n-m}(-i x), real and imaginary parts of the exponential integral of the lower orders on the device (series below x = 2,
    continued fraction above); the oracle differentiates expint symbolically """
    rng = np.random.default_rng(n)
    x = rng.uniform(-4, 4, 14); y = rng.uniform(-4, 4, 12)
    y[:4] = x[:4]
    x[4] = 11.; y[5] = -9.                                      # (arguments in the continued-fraction region)
    kd, ko = lgp.Color(n=n, scale=1.2) * 1.1, oracle.Color(n=n, scale=1.2) * 1.1
    deg = n // 2 - 1
    orders = [(a, b) for a in range(min(deg, 2) + 1) for b in range(min(deg, 2) + 1) if a + b]
    for a, b in orders:
        got = kd.gram(x, y, deriv=(a, b)).to_host()
        want = ko.diff(a, b)(x[:, None], y[None, :])
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-9 * max(np.abs(want).max(), 1e-3), err_msg=f'n={n} {(a, b)}')
        far = np.abs(x[:, None] - y[None, :]) > 1e-6
        np.testing.assert_allclose(got[far], want[far], rtol=0, atol=1e-12 * max(np.abs(want).max(), 1e-3), err_msg=f'n={n} {(a, b)} away from coincidences')
    got = kd.gram(x, y, deriv=(1, 1)).to_host()
    assert abs(got[0, 0] - 1.1 * (n - 1) / (n - 3) / 1.2 ** 2) <= 1e-12 * got[0, 0]       # -k''(0) = (n - 1) E_{n-2}(0)
    with pytest.raises(ValueError):
        kd.gram(x, y, deriv=(deg + 1, 0))


@pytest.mark.parametrize('dw', [1., 0.2, 3.5])
def test_pink_kernel_derivatives_vs_oracle(lgp, dw):
    """ Pink(dw) = (Ci((1 + dw) x) - Ci(x)) / log1p(dw), derivable once (_kernels/_spectral.py:41-66): k' and k'' in closed form, by
    their series where those cancel """
    rng = np.random.default_rng(int(10 * dw))
    x = rng.uniform(-4, 4, 14); y = rng.uniform(-4, 4, 12)
    y[:4] = x[:4]
    y[4] = x[4] + 3e-3; y[5] = x[5] - 1e-5                      # (the series region)
    kd, ko = lgp.Pink(dw=dw, scale=1.2) * 1.1, oracle.Pink(dw=dw, scale=1.2) * 1.1
    for a, b in [(1, 0), (0, 1), (1, 1)]:
        got = kd.gram(x, y, deriv=(a, b)).to_host()
        want = ko.diff(a, b)(x[:, None], y[None, :])
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-9 * max(np.abs(want).max(), 1e-3), err_msg=f'dw={dw} {(a, b)}')
    got = kd.gram(x, y, deriv=(1, 1)).to_host()
    # at a coincidence the reference differentiates the branch its core takes there, cos(delta (1 + dw / 2)) (_spectral.py:62-66)
    assert abs(got[0, 0] - 1.1 * (1 + dw / 2) ** 2 / 1.2 ** 2) <= 1e-12 * got[0, 0]
    with pytest.raises(ValueError):
        kd.gram(x, y, deriv=(2, 0))


@pytest.mark.parametrize('nu', [1.5, 2.4, 3., 3.0004, 5.2])
def test_zeta_kernel_derivatives_vs_oracle(lgp, nu):
    """ Zeta(nu) = Re Li_s(e^(2 pi i delta)) / zeta(s), s = 1 + 2 nu, ceil(nu) - 1 times derivable (_kernels/_zeta.py:32-70): the expansion the
    value core sums, differentiated term by term in theta on the device -- the power series with its power / logarithmic term
    (plain, at integer nu, next to one) and the Fourier series (s >= 10); the oracle differentiates the polylogarithm itself """
    from scipy import special
    rng = np.random.default_rng(int(100 * nu))
    x = rng.uniform(-1.5, 1.5, 5); y = rng.uniform(-1.5, 1.5, 5)         # (the oracle's polylogarithms take a second each)
    y[:3] = x[:3]
    y[3] = x[3] + 1.3                                           # (a whole period apart -- the scale is 1.3 --: a coincidence too)
    y[4] = x[4] + 0.65                                          # (the reflection point)
    kd, ko = lgp.Zeta(nu=nu, scale=1.3) * 1.1, oracle.Zeta(nu=nu, scale=1.3) * 1.1
    deg = int(np.ceil(nu) - 1)
    orders = [(a, b) for a in range(min(deg, 2) + 1) for b in range(min(deg, 2) + 1) if a + b]
    cols = [0, 3, 4]            # the oracle on three columns of y: a coincidence, the period, the reflection point -- and generic pairs in each
    apart = (np.abs(((x[:, None] - y[None, :]) / 1.3 + 0.5) % 1 - 0.5) > 1e-6)[:, cols]
    for a, b in orders:
        got = kd.gram(x, y, deriv=(a, b)).to_host()[:, cols]
        want = ko.diff(a, b)(x[:, None], y[None, cols])
        scale = max(np.abs(want).max(), 1e-3)
        np.testing.assert_allclose(got[apart], want[apart], rtol=0, atol=2e-10 * scale, err_msg=f'nu={nu} {(a, b)}')
    s = 1 + 2 * nu
    got = kd.gram(x, y, deriv=(1, 1)).to_host()
    exact = 1.1 * (2 * np.pi) ** 2 * special.zeta(s - 2, 1) / special.zeta(s, 1) / 1.3 ** 2       # sum_k (2 pi k)^2 k^-s
    assert abs(got[0, 0] - exact) <= 1e-11 * exact and abs(got[3, 3] - exact) <= 1e-8 * exact
    if deg >= 2:
        got = kd.gram(x, y, deriv=(2, 2)).to_host()
        exact = 1.1 * (2 * np.pi) ** 4 * special.zeta(s - 4, 1) / special.zeta(s, 1) / 1.3 ** 4
        assert abs(got[1, 1] - exact) <= 1e-10 * exact
    with pytest.raises(ValueError):
        kd.gram(x, y, deriv=(deg + 1, 0))


def test_nnkernel_derivatives_vs_oracle(lgp):
    """ NNKernel (_kernels/_basic.py:122-154), a function of the dot products x.y, x.x, y.y over the fields: the whole expression in
    Taylor arithmetic on the device (LGP_KERNEL_NNKERNEL), sympy on the same three dot products in the oracle; one-dimensional
    points and named fields, the same and different fields on the two sides """
    rng = np.random.default_rng(31)
    x = rng.uniform(-2, 2, 13); y = rng.uniform(-2, 2, 11); y[:3] = x[:3]
    kd, ko = lgp.NNKernel(sigma0=0.7, scale=1.3, loc=0.1) * 1.2, oracle.NNKernel(sigma0=0.7, scale=1.3, loc=0.1) * 1.2
    for a, b in [(1, 0), (0, 1), (1, 1), (2, 0), (2, 1), (2, 2)]:
        got = kd.gram(x, y, deriv=(a, b)).to_host()
        want = ko.diff(a, b)(x[:, None], y[None, :])
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12 * np.abs(want).max(), err_msg=f'NNKernel {(a, b)}')
    dt = [('a', float), ('b', float), ('c', float)]
    X = np.empty(9, dt); Y = np.empty(7, dt)
    for f in 'abc':
        X[f] = rng.uniform(-1, 1, 9); Y[f] = rng.uniform(-1, 1, 7)
    Y[:3] = X[:3]
    for sx, sy in [(('a', 1), ('b', 1)), (('c', 2), ('c', 1)), (('b', 2), None), (None, ('a', 1)), (('a', 2), ('c', 2))]:
        ox = tuple((sx[1] if sx and nm == sx[0] else 0) for nm in X.dtype.names)
        oy = tuple((sy[1] if sy and nm == sy[0] else 0) for nm in X.dtype.names)
        got = kd.gram(X, Y, deriv=(ox, oy)).to_host()
        want = ko.diff({sx[0]: sx[1]} if sx else 0, {sy[0]: sy[1]} if sy else 0)(X[:, None], Y[None, :])
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12 * np.abs(want).max(), err_msg=f'NNKernel {sx} {sy}')
    # the posterior of a process observed through its slope: device against oracle
    xs = np.linspace(-1.5, 1.5, 9)
    yv = np.tanh(2 * x)
    nf = len(x)
    gc = {('f', 'f'): 1e-4 * np.eye(nf), ('df', 'df'): 1e-4 * np.eye(5), ('f', 'df'): np.zeros((nf, 5)), ('df', 'f'): np.zeros((5, nf))}
    for mod, kk in ((lgp, kd), (oracle, ko)):
        gp = mod.GP(kk, checkpos=False, checksym=False).addx(x, 'f').addx(x[:5], 'df', deriv=1).addx(xs, 'out')
        m, c = gp.predfromdata({'f': yv, 'df': 2 / np.cosh(2 * x[:5]) ** 2}, 'out', gc, raw=True)
        if mod is lgp:
            md, cd = m, c
    assert np.linalg.norm(md - m) <= 1e-7 * np.linalg.norm(m) and np.linalg.norm(cd - c, 2) <= 1e-7 * max(np.linalg.norm(c, 2), 1e-6)


def test_taylor_and_causal_expquad_derivatives_vs_oracle(lgp):
    """ Taylor (I0(2 sqrt(x y)), J0 where x y < 0: the series of h^(m)(w) below |w| = 30, the library's I_m / J_m beyond) and
    CausalExpQuad(alpha=0), the two remaining kernels of _kernels/_basic.py that the reference marks derivable """
    rng = np.random.default_rng(77)
    x = np.concatenate([rng.uniform(-3, 3, 14), [7., -8., 9.5, 0.]])
    y = np.concatenate([rng.uniform(-3, 3, 9), [9., 6., -7.5, 0.]])
    y[:3] = x[:3]
    kd, ko = lgp.Taylor(scale=1.1) * 0.9, oracle.Taylor(scale=1.1) * 0.9
    for a, b in [(1, 0), (0, 1), (1, 1), (2, 0), (2, 1), (1, 2), (2, 2)]:
        got = kd.gram(x, y, deriv=(a, b)).to_host()
        want = ko.diff(a, b)(x[:, None], y[None, :])
        np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-12 * np.abs(want).max(), err_msg=f'Taylor {(a, b)}')
    x = rng.uniform(0, 3, 15); y = rng.uniform(0, 3, 12); y[:4] = x[:4]
    kd, ko = lgp.CausalExpQuad(alpha=0, scale=0.8) * 1.3, oracle.CausalExpQuad(alpha=0, scale=0.8) * 1.3
    for a, b in [(1, 0), (1, 1), (2, 1), (2, 2)]:
        got = kd.gram(x, y, deriv=(a, b)).to_host()
        want = ko.diff(a, b)(x[:, None], y[None, :])
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9 * np.abs(want).max(), err_msg=f'CausalExpQuad {(a, b)}')
    with pytest.raises(ValueError):
        lgp.CausalExpQuad(alpha=0.5).gram(x, y, deriv=(1, 0))


def test_matern_of_real_order_derivatives_meet_the_half_integer_closed_form(lgp):
    """ nu -> p + 1/2: the derivative core of real order next to Maternp's own (Laurent polynomials, another code path), nu = 2.5
    approached from both sides; and its exact values at a coincidence, 3 nu^2 / ((nu - 1)(nu - 2)) for order (2, 2) """
    x = np.array([0.3, 0.9, 1.4, 2.2]); y = np.array([0.3, 0.5, 1.4, 3.1])
    ref = lgp.Maternp(p=2).gram(x, y, deriv=(2, 2)).to_host()
    for nu in (2.5 - 1e-7, 2.5 + 1e-7):
        got = lgp.Matern(nu=nu).gram(x, y, deriv=(2, 2)).to_host()
        np.testing.assert_allclose(got, ref, rtol=0, atol=3e-5 * np.abs(ref).max())           # (d/dnu of the values times 1e-7)
        assert abs(got[0, 0] - 3 * nu ** 2 / ((nu - 1) * (nu - 2))) <= 1e-12 * got[0, 0]
    for nu, a, b, exact in [(1.3, 1, 1, 1.3 / 0.3), (3.3, 1, 1, 3.3 / 2.3), (3.3, 2, 0, -3.3 / 2.3)]:
        got = lgp.Matern(nu=nu).gram(x, y, deriv=(a, b)).to_host()
        assert abs(got[0, 0] - exact) <= 1e-12 * abs(exact) and abs(got[2, 2] - exact) <= 1e-12 * abs(exact)


def test_posterior_with_derivative_data_of_a_periodic_process(lgp):
    """ values and slopes of a periodic function observed together (GP.addx(..., deriv=1) with the Periodic kernel), the function
    and its slope predicted: device against oracle """
    rng = np.random.default_rng(8)
    x = np.sort(rng.uniform(0, 2 * np.pi, 25))
    xs = np.linspace(0, 2 * np.pi, 18)
    y, dy = np.sin(x) + 0.5 * np.sin(2 * x), np.cos(x) + np.cos(2 * x)

    def run(mod):
        gp = (mod.GP(mod.Periodic(outerscale=1.2) * 1.5, checkpos=False, checksym=False)
              .addx(x, 'f').addx(x, 'df', deriv=1).addx(xs, 'g').addx(xs, 'dg', deriv=1))
        noise = {(a, b): (1e-4 * np.eye(25) if a == b else np.zeros((25, 25))) for a in ('f', 'df') for b in ('f', 'df')}
        return gp.predfromdata({'f': y, 'df': dy}, ['g', 'dg'], noise, raw=True)
    (md, cd), (mo, co) = run(lgp), run(oracle)
    for k in mo:
        np.testing.assert_allclose(md[k], mo[k], rtol=0, atol=1e-7)
    for k in co:
        np.testing.assert_allclose(cd[k], co[k], rtol=0, atol=1e-8)
    assert np.max(np.abs(md['dg'] - (np.cos(xs) + np.cos(2 * xs)))) < 0.05         # it does predict the slope


@pytest.mark.parametrize('a,b', [(1, 0), (1, 1), (2, 0), (2, 1), (2, 2)])
def test_derivatives_of_products_and_sums_of_kernels(lgp, a, b):
    """ `_Kernel/_ops.py:115-211` differentiates whatever `_Kernel/_alg.py:32-82` built.  Device: Leibniz' rule over derived blocks
    of the factors (`_Composite._gram_derived`).  Checked two ways: against the same rule spelled out over the ORACLE's derived
    blocks (closed forms / sympy), and -- independently of any product rule -- against central differences of the oracle's
    composite kernel itself """
    import math
    rng = np.random.default_rng(7 + 10 * a + b)
    x = rng.uniform(0.2, 3, 19)
    y = rng.uniform(0.2, 3, 23)
    f = lambda mod: (mod.ExpQuad(scale=1.3), mod.Cauchy(alpha=2, beta=2.5, scale=0.9), mod.Maternp(p=3, scale=1.7))
    (e, c, mt), (eo, co, mo) = f(lgp), f(oracle)
    X, Y = x[:, None], y[None, :]
    # product of two, times a scalar, plus a constant (which must drop out)
    got = (0.7 * (e * c) + 0.4).gram(x, y, deriv=(a, b)).to_host()
    want = 0.7 * sum(math.comb(a, i) * math.comb(b, j) * eo.diff(i, j)(X, Y) * co.diff(a - i, b - j)(X, Y) for i in range(a + 1) for j in range(b + 1))
    scale = np.abs(want).max()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-11 * scale)
    # central differences of the composite's VALUES (oracle), order (a, b): no product rule involved
    ko = 0.7 * (eo * co) + 0.4
    h = 2e-3

    def stencil(order):
        return {0: [(0, 1.)], 1: [(-1, -0.5 / h), (1, 0.5 / h)], 2: [(-1, 1 / h ** 2), (0, -2 / h ** 2), (1, 1 / h ** 2)]}[order]
    fd = sum(wa * wb * ko((x + ia * h)[:, None], (y + ib * h)[None, :]) for ia, wa in stencil(a) for ib, wb in stencil(b))
    np.testing.assert_allclose(got, fd, rtol=0, atol=3e-5 * scale)
    # a sum plus a constant, and a product with a sum inside
    got = ((e + mt) + 0.3).gram(x, y, deriv=(a, b)).to_host()
    want = eo.diff(a, b)(X, Y) + mo.diff(a, b)(X, Y)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-11 * np.abs(want).max())
    got = (e * (c + mt)).gram(x, y, deriv=(a, b)).to_host()
    want = sum(math.comb(a, i) * math.comb(b, j) * eo.diff(i, j)(X, Y) * (co.diff(a - i, b - j)(X, Y) + mo.diff(a - i, b - j)(X, Y))
               for i in range(a + 1) for j in range(b + 1))
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-11 * np.abs(want).max())


def test_posterior_with_derivative_data_of_a_product_process(lgp):
    """ values and slopes of a process with a product kernel observed together (the oracle assembles the same blocks by Leibniz) """
    rng = np.random.default_rng(11)
    x = np.sort(rng.uniform(0, 6, 30))
    xs = np.linspace(0.5, 5.5, 9)
    gp = (lgp.GP(lgp.ExpQuad(scale=1.5) * lgp.Periodic(outerscale=2.), checkpos=False, checksym=False)
          .addx(x, 'f').addx(x, 'df', deriv=1).addx(xs, 'fs').addx(xs, 'dfs', deriv=1))
    fx, dfx = np.sin(x) * np.cos(x / 3), np.cos(x) * np.cos(x / 3) - np.sin(x) * np.sin(x / 3) / 3
    z = np.zeros((30, 30))
    noise = {('f', 'f'): 1e-4 * np.eye(30), ('df', 'df'): 1e-4 * np.eye(30), ('f', 'df'): z, ('df', 'f'): z}
    m, c = gp.predfromdata({'f': fx, 'df': dfx}, ['fs', 'dfs'], noise, raw=True)
    # the slope of the posterior mean of the values is the posterior mean of the slopes: central differences on shifted points
    h = 1e-4
    gp2 = (lgp.GP(lgp.ExpQuad(scale=1.5) * lgp.Periodic(outerscale=2.), checkpos=False, checksym=False)
           .addx(x, 'f').addx(x, 'df', deriv=1).addx(xs + h, 'p').addx(xs - h, 'q'))
    m2, _ = gp2.predfromdata({'f': fx, 'df': dfx}, ['p', 'q'], noise, raw=True)
    np.testing.assert_allclose(m['dfs'], (m2['p'] - m2['q']) / (2 * h), rtol=0, atol=1e-6)
    assert np.abs(m['fs'] - np.sin(xs) * np.cos(xs / 3)).max() < 0.05 and np.abs(m['dfs']).max() > 0.3


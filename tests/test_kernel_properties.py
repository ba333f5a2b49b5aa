"""
The generic kernel checks of the reference (``tests/kernels/test_kernels.py:98-420``: positive semidefinite and symmetric Gram
matrices on scalar and structured points, symmetry of off-diagonal blocks, implicit (array-valued) fields, loc / scale as a change
of variable, stationarity, unit variance, continuity in zero, derivatives) run over every kernel of the device package.
"""

import numpy as np
import pytest
from numpy.lib import recfunctions
from scipy import linalg

from . import util

@pytest.fixture(scope='module', params=[pytest.param('device', marks=pytest.mark.gpu), 'oracle'])
def lgp(request):
    """ the device package (GPU) and the numpy oracle (CPU): the same checks pin both """
    if request.param == 'oracle':
        import oracle
        return oracle
    import lsqfitgp_amd
    lsqfitgp_amd.Context.default()
    return lsqfitgp_amd


# name -> (constructor keyword sets, maxdim (None: any), stationary, domain of the inputs, derivable orders)
KERNELS = {
    'ExpQuad': ([{}], None, True, (-5, 5), 2),
    'Maternp': ([dict(p=0), dict(p=1), dict(p=2), dict(p=3), dict(p=7)], None, True, (-5, 5), 'p'),
    'Matern': ([dict(nu=v + 0.5) for v in range(5)] + [dict(nu=v + 0.49) for v in range(5)] + [dict(nu=v + 0.51) for v in range(5)]      # ref :483-490
               + [dict(nu=v) for v in range(5)] + [dict(nu=v - 0.01) for v in range(1, 5)] + [dict(nu=v + 0.01) for v in range(5)], None, True, (-5, 5), 0),
    'GammaExp': ([dict(gamma=0.3), dict(gamma=1), dict(gamma=1.7), dict(gamma=2)], None, True, (-5, 5), 0),
    'Cauchy': ([dict(alpha=0.5, beta=1), dict(alpha=1, beta=2), dict(alpha=2, beta=2), dict(alpha=2, beta=0.3)], None, True, (-5, 5), 0),
    'Wendland': ([dict(k=0, alpha=1), dict(k=1, alpha=2), dict(k=2, alpha=2.5), dict(k=3, alpha=2)], 'alpha', True, (-5, 5), 0),
    'CausalExpQuad': ([dict(alpha=0), dict(alpha=1), dict(alpha=4)], None, True, (-5, 5), 0),
    'Log': ([{}], None, True, (-5, 5), 0),
    'Constant': ([{}], None, True, (-5, 5), 0),
    'White': ([{}], None, True, (-5, 5), 0),
    'Linear': ([{}], None, False, (-5, 5), 0),
    'Expon': ([{}], 1, True, (-5, 5), 0),
    'HoleEffect': ([{}], 1, True, (-5, 5), 0),
    'Periodic': ([dict(outerscale=1), dict(outerscale=0.3)], 1, True, (-5, 5), 0),
    'Cos': ([{}], 1, True, (-5, 5), 0),
    'Sinc': ([{}], 1, True, (-5, 5), 0),
    'Celerite': ([dict(gamma=1, B=0), dict(gamma=0.5, B=0.5), dict(gamma=2, B=-1)], 1, True, (-5, 5), 0),
    'Harmonic': ([dict(Q=0.2), dict(Q=0.8), dict(Q=1), dict(Q=4)], 1, True, (-5, 5), 0),
    'Circular': ([dict(tau=4, c=0.5), dict(tau=7, c=0.2)], 1, True, (-5, 5), 0),
    'StationaryFracBrownian': ([dict(H=0.2), dict(H=0.5), dict(H=0.9), dict(H=1)], 1, True, (-5, 5), 0),
    'Bessel': ([{}] + [dict(nu=nu) for nu in range(5)] + [dict(nu=nu - 0.01) for nu in range(1, 5)] + [dict(nu=nu + 0.01) for nu in range(5)]
               + [dict(nu=nu + 0.5) for nu in range(5)], 'bessel', True, (-5, 5), 0),                       # ref :697-701
    'Color': ([dict(n=n) for n in (2, 3, 4, 5, 6, 20)], 1, True, (-5, 5), 0),            # ref :796-798
    'Pink': ([dict(dw=1), dict(dw=0.2), dict(dw=30)], 1, True, (-5, 5), 0),
    'Zeta': ([dict(nu=nu) for nu in (0, 0.1, 1, 1.5, 4.9, 1000)], 1, True, (-5, 5), 0),     # ref :593-595
    'Wiener': ([{}], 1, False, (0, 5), 0),
    'WienerIntegral': ([{}], 1, False, (0, 5), 0),
    'OrnsteinUhlenbeck': ([{}], 1, False, (0, 5), 0),
    'BrownianBridge': ([{}], 1, False, (0, 1), 0),
    'FracBrownian': ([dict(H=0.1), dict(H=0.5), dict(H=1), dict(H=0.3, K=0.4), dict(H=1, K=0.7)], 1, False, (-5, 5), 0),
    'Decaying': ([dict(alpha=1), dict(alpha=0.3), dict(alpha=3)], 1, False, (0, 5), 0),
    'Taylor': ([{}], 1, False, (-2, 2), 0),
    'Rescaling': ([{}], None, False, (-5, 5), 0),
    'NNKernel': ([dict(sigma0=1), dict(sigma0=0.3)], None, False, (-5, 5), 0),
    'Gibbs': ([{}], None, True, (-5, 5), 0),
}
CASES = [(name, kw) for name, spec in KERNELS.items() for kw in spec[0]]
IDS = [f'{n}-{"-".join(f"{k}{v}" for k, v in kw.items())}' for n, kw in CASES]


def maxdim(name, kw):
    md = KERNELS[name][1]
    if md == 'alpha':
        return int(np.floor(2 * kw['alpha'] - 1))
    if md == 'bessel':
        return 2 * (int(np.floor(kw.get('nu', 0))) + 1)
    return md


def scalar_x(rng, name, size=100):
    lo, hi = KERNELS[name][3]
    return rng.uniform(lo, hi, size)


def nd_x(rng, name, ndim, size=100):
    lo, hi = KERNELS[name][3]
    out = np.empty(size, [(f'f{i}', float) for i in range(ndim)])
    for f in out.dtype.names:
        out[f] = rng.uniform(lo, hi, size)
    return out


def check_sym_and_psd(cov, eps=np.finfo(float).eps):
    util.assert_allclose(cov, cov.T, rtol=1e-5, atol=1e-7)            # ref :160-163
    eigv = linalg.eigvalsh(cov)
    assert np.min(eigv) >= -len(cov) * eps * max(np.max(eigv), 1e-300) * 4


@pytest.mark.parametrize('name,kw', CASES, ids=IDS)
def test_positive_and_symmetric(lgp, rng, name, kw):
    """ ref :165-187 `test_positive_scalar_0`, `test_positive_nd_0`, and the scalar operations on top """
    k = getattr(lgp, name)(**kw)
    x = scalar_x(rng, name)
    cov = k(x[None, :], x[:, None])
    check_sym_and_psd(cov)
    check_sym_and_psd((k * 2.5 + 0.3)(x[None, :], x[:, None]))
    md = maxdim(name, kw)
    if md is None or md >= 3:
        xs = nd_x(rng, name, 3)
        check_sym_and_psd(k(xs[None, :], xs[:, None]))
    # the Gram entry point agrees with the broadcasting call
    if hasattr(k, 'gram'):
        util.assert_allclose(k.gram(x, x).to_host(), cov, rtol=1e-13, atol=1e-14)


@pytest.mark.parametrize('name,kw', CASES, ids=IDS)
def test_symmetric_offdiagonal(lgp, rng, name, kw):
    """ ref :217-233 """
    k = getattr(lgp, name)(**kw)
    x = scalar_x(rng, name)
    a, b = x[:50, None], x[None, 50:]
    util.assert_allclose(k(a, b), k(b, a), atol=1e-10)
    if hasattr(k, 'gram'):
        util.assert_allclose(k.gram(x[:50], x[50:]).to_host(), k.gram(x[50:], x[:50]).to_host().T, atol=1e-10)


@pytest.mark.parametrize('name,kw', CASES, ids=IDS)
def test_implicit_fields_and_loc_scale(lgp, rng, name, kw):
    """ ref :235-262: an array-valued field is the same as named fields; loc / scale are a change of variable """
    k = getattr(lgp, name)(**kw)
    md = maxdim(name, kw)
    loc, scale = -2., 3.
    x = scalar_x(rng, name, 60)
    kls = getattr(lgp, name)(**kw, scale=scale, loc=loc)
    xt = x * scale + loc                                # then (xt - loc) / scale = x up to rounding
    util.assert_allclose(kls(xt[:, None], xt[None, :]), k(x[:, None], x[None, :]), rtol=1e-11, atol=1e-12)
    if md is not None and md < 3:
        xs = nd_x(rng, name, 3, 10)
        with pytest.raises((ValueError, AssertionError)):
            k(xs[:, None], xs[None, :])
        return
    xs = nd_x(rng, name, 3, 60)
    dense = recfunctions.structured_to_unstructured(xs)
    ximp = np.empty(len(xs), [('f0', float, (3,))])
    ximp['f0'] = dense
    util.assert_allclose(k(xs[:, None], xs[None, :]), k(ximp[:, None], ximp[None, :]), atol=1e-15, rtol=1e-14)
    xt = np.empty_like(xs)
    for f in xs.dtype.names:
        xt[f] = xs[f] * scale + loc
    util.assert_allclose(kls(xt[:, None], xt[None, :]), k(xs[:, None], xs[None, :]), rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize('name,kw', [c for c in CASES if KERNELS[c[0]][2]], ids=[i for i, c in zip(IDS, CASES) if KERNELS[c[0]][2]])
def test_stationary_normalized_continuous(lgp, rng, name, kw):
    """ ref :265-308: the variance of a stationary kernel is 1 everywhere; it is continuous in zero unless it is known not to be """
    k = getattr(lgp, name)(**kw)
    x = scalar_x(rng, name)
    var = k(x, x)
    util.assert_allclose(var, 1, rtol=1e-14, atol=1e-15)
    discontinuous = ((name == 'Cauchy' and kw.get('alpha', 2) < 1) or (name == 'GammaExp' and kw.get('gamma', 1) < 1) or name == 'White' or (name == 'Matern' and kw['nu'] < 0.5) or (name == 'Zeta' and kw['nu'] < 0.5)
                     or (name == 'StationaryFracBrownian' and kw.get('H', 0.5) < 0.5))
    if not discontinuous:
        util.assert_allclose(k(np.zeros(1), np.full(1, 1e-15)), k(np.zeros(1), np.zeros(1)), rtol=1e-10)


@pytest.mark.parametrize('name,kw,order', [('ExpQuad', {}, 1), ('ExpQuad', {}, 2), ('Maternp', dict(p=1), 1), ('Maternp', dict(p=2), 1),
                                           ('Maternp', dict(p=2), 2), ('Maternp', dict(p=3), 2)])
def test_derivatives_positive_and_symmetric(lgp, rng, name, kw, order):
    """ ref :310-420: the process derived `order` times has a positive semidefinite covariance; derivatives taken on either
    side mirror each other """
    k = getattr(lgp, name)(**kw, scale=1.3)
    if not hasattr(k, 'gram'):
        pytest.skip('the oracle\'s derivatives have tests of their own (tests/test_deriv_known_answer.py)')
    x = scalar_x(rng, name, 80)
    cov = k.gram(x, x, deriv=(order, order)).to_host()
    check_sym_and_psd(cov, eps=1e-12)
    left = k.gram(x[:40], x[40:], deriv=(order, 0)).to_host()
    right = k.gram(x[40:], x[:40], deriv=(0, order)).to_host()
    util.assert_allclose(left, right.T, atol=1e-10)
    if name == 'ExpQuad':
        xs = nd_x(rng, name, 2, 60)
        cov = k.gram(xs, xs, deriv=((order, 0), (order, 0))).to_host()
        check_sym_and_psd(cov, eps=1e-12)


def test_every_kernel_is_listed(lgp):
    """ ref :113-114 `test_public`, the other way round: nothing exported escapes these checks """
    if not hasattr(lgp, 'Context'):
        assert all(hasattr(lgp, n) for n in KERNELS)
        return
    exported = {n for n in dir(lgp) if isinstance(getattr(lgp, n), type) and issubclass(getattr(lgp, n), lgp.Kernel) and n not in ('Kernel', 'BART', 'Categorical')}
    assert exported == set(KERNELS) | {n for n, _ in DISCRETE}                  # (the kernels of integers and of strings: checked below)


# ---- kernels whose inputs are not real numbers: the discrete-time MA / AR (integers) and BagOfWords (strings) -------------------

_rng_ma = np.random.default_rng([2023, 8, 25, 0, 25])                         # ref tests/kernels/test_kernels.py:754
DISCRETE = ([('MA', dict(w=w)) for w in ([], [0], [1], [1, 1], [1, -1], [2, 1], [1, 2, 3, 4, 5], _rng_ma.standard_normal(30))]      # ref :756-761
            + [('AR', dict(phi=phi, maxlag=100)) for phi in ([], [0], [-0.5], [0.5], [0.9], [-0.9], [0.5, 0], [0, 0.5], 3 * [0] + [0.5])]   # ref :769-770
            + [('AR', dict(gamma=g, maxlag=100)) for g in ([0], [1], [1, 0], [1, 0.5], [1, 0.5, 0.25], [1, -0.9])]                  # ref :771-772
            + [('AR', dict(slnr=r, lnc=c)) for r, c in (([], []), ([1 / 10], []), ([1 / 2], []), ([1 / 10, 1 / 2], []), ([1 / 10, 1 / 10], []),
                                                       ([1 / 10, 1 / 10, -1 / 2], []), ([], [1 / 10 + 1j]), ([], [1 / 2 + 1j]),
                                                       ([], [1 / 10 + 1j, 1 / 2 + 2j]), ([], [1 / 10 + 1j, 1 / 10 + 2j]),
                                                       ([1 / 10], [1 / 2 + 1j]), ([-1 / 10, 1 / 3], [1 / 10 + 1j, 1 / 2 + 0.5j]))]      # ref :773-790
            + [('BagOfWords', {})])
DISCRETE_IDS = [f'{n}-{i}' for i, (n, _) in enumerate(DISCRETE)]

LIPSUM = """Duis mollis 12 est non 7 commodo luctus, nisi erat 12 porttitor ligula 3 eget lacinia 44 odio sem nec elit 7 7 Curabitur
blandit # tempus porttitor $ Sed posuere 100 consectetur est at 3 lobortis % Cras mattis & consectetur purus + sit amet 12"""


def discrete_x(rng, name, size=100):
    if name == 'BagOfWords':
        words = np.array(LIPSUM.split())
        return np.array([' '.join(rng.choice(words, 10)) for _ in range(30)])
    return rng.integers(0, 100, size)                                           # ref :763-765


@pytest.mark.parametrize('name,kw', DISCRETE, ids=DISCRETE_IDS)
def test_discrete_positive_symmetric_stationary(lgp, rng, name, kw):
    """ ref `Base` / `Stationary` checks over the kernels of integers and strings: positive semidefinite and symmetric, the
    off-diagonal blocks transposes of each other, shift invariance of the discrete-time kernels, normalisation """
    k = getattr(lgp, name)(**kw)
    x = discrete_x(rng, name)
    cov = k(x[None, :], x[:, None])
    assert cov.shape == (len(x), len(x)) and np.all(np.isfinite(cov))
    scale = max(np.max(np.abs(cov)), 1e-300)
    check_sym_and_psd(cov / scale, eps=(200 if name == 'AR' else 1) * np.finfo(float).eps)
    check_sym_and_psd((k * 2.5 + 0.3)(x[None, :], x[:, None]) / scale, eps=(200 if name == 'AR' else 1) * np.finfo(float).eps)
    h = len(x) // 2
    util.assert_allclose(k(x[:h, None], x[None, h:]), k(x[None, h:], x[:h, None]), atol=1e-10 * scale)
    if hasattr(k, 'gram'):
        util.assert_allclose(k.gram(x, x).to_host(), cov, rtol=1e-13, atol=1e-14 * scale)
        util.assert_allclose(k.gram(x[:h], x[h:]).to_host(), k.gram(x[h:], x[:h]).to_host().T, atol=1e-10 * scale)
    if name != 'BagOfWords':
        util.assert_allclose(k((x + 1000)[None, :], (x + 1000)[:, None]), cov, atol=1e-13 * scale)      # stationary (ref :296-301)
        kn = getattr(lgp, name)(**kw, norm=True)
        var = kn(x, x)
        if np.all(np.isfinite(var)) and scale > 1e-300 and not (name == 'MA' and not np.any(kw['w'])) and not (name == 'AR' and 'gamma' in kw and kw['gamma'][0] == 0):
            util.assert_allclose(var, 1, rtol=1e-13)
        with pytest.raises((TypeError, AssertionError, IndexError)):
            k(x.astype(float)[None, :] + 0.5, x.astype(float)[:, None])                                      # the inputs must be integers
    # structured points with one field
    xs = np.empty(len(x), [('t', x.dtype)])
    xs['t'] = x
    util.assert_allclose(k(xs[None, :], xs[:, None]), cov, atol=1e-13 * scale)


def test_ar_parametrisations_agree(lgp, rng):
    """ the three parametrisations of AR describe the same process (``_kernels/_arma.py:141-160``), and AR(1) / MA(1) have the
    textbook autocovariances """
    x = rng.integers(0, 60, 80)
    phi = lgp.AR.phi_from_roots(np.sort(np.array([0.2, -0.4])), np.array([0.3 + 1.1j]))
    assert phi.shape == (4,)
    gamma = lgp.AR.gamma_from_phi(phi)
    util.assert_allclose(lgp.AR.phi_from_gamma(gamma), phi, rtol=1e-10, atol=1e-12)
    a = lgp.AR(slnr=[0.2, -0.4], lnc=[0.3 + 1.1j])(x[:, None], x[None, :])
    b = lgp.AR(phi=phi, maxlag=60)(x[:, None], x[None, :])
    c = lgp.AR(gamma=gamma, maxlag=60)(x[:, None], x[None, :])
    util.assert_allclose(b, a, rtol=1e-10, atol=1e-12)
    util.assert_allclose(c, a, rtol=1e-10, atol=1e-12)
    lag = np.abs(x[:, None] - x[None, :])
    util.assert_allclose(lgp.AR(phi=[0.7], maxlag=60)(x[:, None], x[None, :]), 0.7 ** lag / (1 - 0.49), rtol=1e-12)
    util.assert_allclose(lgp.MA(w=[1, 0.5])(x[:, None], x[None, :]), np.where(lag == 0, 1.25, np.where(lag == 1, 0.5, 0.)), rtol=1e-15)
    # the autocovariance against a long moving-average representation of the same process (an independent route)
    psi = np.zeros(4000)
    psi[0] = 1
    for i in range(1, len(psi)):
        psi[i] = sum(phi[k] * psi[i - 1 - k] for k in range(min(len(phi), i)))
    brute = np.array([psi[:len(psi) - m] @ psi[m:] for m in range(61)])
    util.assert_allclose(a, brute[lag], rtol=1e-9)
    # lags beyond maxlag are not numbers; bad parameter sets are refused
    far = lgp.AR(phi=[0.5], maxlag=5)(np.array([0]), np.array([9]))
    assert np.isnan(far).all()
    for bad in (dict(phi=[0.5]), dict(phi=[0.5], gamma=[1, 0.5], maxlag=3), dict(slnr=[0.1]), dict()):
        k = lgp.AR(**bad)                                             # (raised when the kernel is evaluated, as in the reference)
        with pytest.raises(ValueError):
            k(np.array([1, 2, 3]), np.array([3, 2, 1]))


def test_bag_of_words_counts(lgp):
    """ ``_kernels/_basic.py:272-296`` with the delimiters the reference really has: its pattern contains the range ':'..'–', so
    letters separate words too and only digits (and # $ % & * +) make them """
    x = np.array(['12 7 12', 'a12b7', '7;7;7 +', '', 'word words'])
    k = lgp.BagOfWords()
    cov = k(x[:, None], x[None, :])
    expect = np.array([[5., 3., 3., 0., 0.],          # {12: 2, 7: 1}
                       [3., 2., 3., 0., 0.],          # {12: 1, 7: 1}
                       [3., 3., 10., 0., 0.],         # {7: 3, +: 1}
                       [0., 0., 0., 0., 0.],
                       [0., 0., 0., 0., 0.]])
    util.assert_allclose(cov, expect)

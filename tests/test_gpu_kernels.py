"""
GPU parity tests of the Gram kernels (C-ABI: lgp_gram_isotropic, lgp_pairs_isotropic,
lgp_gram_bart, lgp_pairs_bart) through the host kernel objects: the reference's
kernel pins (tests/kernels/test_kernels.py, tests/kernels/test_bart.py) and
entry-wise comparison with the CPU oracle and the committed golden fixtures.
"""

import os

import ctypes

import numpy as np
import pytest

import oracle
from . import util
from .test_oracle_kernels import KernelPins

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def lgp():
    import lsqfitgp_amd
    lsqfitgp_amd.Context.default()
    return lsqfitgp_amd


def _factories():
    import lsqfitgp_amd as lgp
    return {
        'ExpQuad': lambda **kw: lgp.ExpQuad(**kw),
        'Maternp0': lambda **kw: lgp.Maternp(p=0, **kw),
        'Maternp1': lambda **kw: lgp.Maternp(p=1, **kw),
        'Maternp2': lambda **kw: lgp.Maternp(p=2, **kw),
        'Maternp5': lambda **kw: lgp.Maternp(p=5, **kw),
        'GammaExp1.5': lambda **kw: lgp.GammaExp(gamma=1.5, **kw),
        'Cauchy1-3': lambda **kw: lgp.Cauchy(alpha=1, beta=3, **kw),
    }


class TestDeviceKernelPins(KernelPins):
    """ the reference's generic kernel tests, run on the device kernels """

    @pytest.fixture(params=['ExpQuad', 'Maternp0', 'Maternp1', 'Maternp2', 'Maternp5', 'GammaExp1.5', 'Cauchy1-3'])
    def make(self, request):
        return _factories()[request.param]


PAIRS = [
    ('ExpQuad', {}, {}), ('ExpQuad', dict(scale=2.5, loc=0.3), {}), ('Maternp', dict(p=2, scale=1.3), dict(p=2, scale=1.3)),
    ('Maternp', dict(p=0), dict(p=0)), ('Maternp', dict(p=7, loc=-1.), dict(p=7, loc=-1.)),
]


POW_KERNELS = [('GammaExp', {}), ('GammaExp', dict(gamma=0.7, scale=2.)), ('GammaExp', dict(gamma=2, loc=0.5)), ('GammaExp', dict(gamma=1.9)),
               ('Cauchy', {}), ('Cauchy', dict(alpha=2, beta=0.5, scale=0.7)), ('Cauchy', dict(alpha=1.3, beta=4.)), ('Cauchy', dict(alpha=0.5, beta=30., loc=-2.))]


@pytest.mark.parametrize('name,kw', POW_KERNELS, ids=[f'{n}-{k}' for n, k in POW_KERNELS])
@pytest.mark.parametrize('flt', [np.float64, np.float32])
def test_gammaexp_cauchy_vs_oracle(lgp, name, kw, flt, rng):
    """ GammaExp / Cauchy (_kernels/_basic.py:94-115,315-344) against the oracle: the device `pow`/`exp` against libm's,
    through exp(-pow(.)) and pow(1 + pow(.)/beta, .) -- a few ulp of the exponent t = -log k, i.e. (8 + 4 t) eps relative;
    exact where the reference branches on the parameter (gamma == 2, alpha == 2 act on r2 itself); k(x, x) = 1 exactly """
    n, m = 257, 130
    x = rng.uniform(-4, 4, (n, 2)).astype(flt); y = rng.uniform(-4, 4, (m, 2)).astype(flt)
    dt = [('a', flt), ('b', flt)]
    xs = np.empty(n, dt); ys = np.empty(m, dt)
    xs['a'], xs['b'], ys['a'], ys['b'] = x[:, 0], x[:, 1], y[:, 0], y[:, 1]
    kd = getattr(lgp, name)(**kw); ko = getattr(oracle, name)(**kw)
    G = kd.gram(xs, ys).to_host()
    R = ko(xs[:, None], ys[None, :])
    assert G.dtype == flt and R.dtype == flt and G.shape == (n, m)
    eps = np.finfo(flt).eps

    amp = kw.get('beta', 2) / kw.get('alpha', 2) if name == 'Cauchy' else 0       # (1 + u)^(-beta/alpha): ulps of 1 + u times beta/alpha

    def close(a, b):
        # a few ulp in the exponent t = -log k of exp(-t) / pow(., .): relative error (8 + 4 t + 2 beta/alpha) eps
        b64 = np.asarray(b, float)
        tol = (8 + 4 * np.abs(np.log(np.maximum(b64, 1e-300))) + 2 * amp) * eps * b64 + np.finfo(flt).tiny     # (+ tiny: float32 results below the normal range are flushed to zero on the device)
        assert np.all(np.abs(np.asarray(a, float) - b64) <= tol)

    close(G, R)
    D = kd.gram(xs, xs).to_host()
    np.testing.assert_array_equal(np.diag(D), np.ones(n, flt))
    np.testing.assert_array_equal(D, D.T)
    Gp = kd.pairs(xs[:m], ys).to_host()
    close(Gp, ko(xs[:m], ys))
    # algops and the broadcasting call contract go through the same core
    np.testing.assert_allclose((kd * 2.5 + 1)(xs[:5, None], ys[None, :7]), (ko * 2.5 + 1)(xs[:5, None], ys[None, :7]), rtol=(64 + 2 * amp) * eps)


def test_gammaexp_cauchy_arguments(lgp):
    for bad in (dict(gamma=0), dict(gamma=2.5), dict(gamma=-1)):
        with pytest.raises(AssertionError):
            lgp.GammaExp(**bad)
    for bad in (dict(alpha=0), dict(alpha=2.1), dict(beta=0), dict(beta=-3)):
        with pytest.raises(AssertionError):
            lgp.Cauchy(**bad)
    from lsqfitgp_amd import _lib
    lib = _lib.load(); ctx = lgp.Context.default()
    a = lgp.DeviceArray.zeros((1, 4)); o = lgp.DeviceArray.zeros((4, 4))
    par = (ctypes.c_double * 2)(3.0, 1.5)          # gamma out of range is refused by the library as well
    assert lib.lgp_gram_isotropic_f64(ctx.handle, _lib.LGP_KERNEL_GAMMAEXP, 0, 2, par, 1, a.ptr, a.ld, 4, a.ptr, a.ld, 4, 0, 0., 0, 1., 0, None, None, 0, o.ptr, o.ld) < 0
    assert lib.lgp_gram_isotropic_f64(ctx.handle, _lib.LGP_KERNEL_EXPQUAD, 0, 2, par, 1, a.ptr, a.ld, 4, a.ptr, a.ld, 4, 0, 0., 0, 1., 0, None, None, 0, o.ptr, o.ld) < 0


@pytest.mark.parametrize('name,kw,_', PAIRS)
@pytest.mark.parametrize('n,m', [(1, 1), (3, 700), (33, 511), (512, 513), (777, 333), (1500, 2)])
def test_gram_ulp_1d(lgp, name, kw, _, n, m, rng):
    """ every entry within 2 ulp of the oracle (libm exp vs device exp); ragged sizes cover all tile edges """
    x = rng.uniform(-5, 5, n); y = rng.uniform(-5, 5, m)
    kd = getattr(lgp, name)(**kw); ko = getattr(oracle, name)(**kw)
    G = kd.gram(x, y).to_host()
    R = ko(x[:, None], y[None, :])
    assert G.shape == (n, m) and G.dtype == np.float64
    assert util.ulp_distance(G, R) <= 2
    Gp = kd.pairs(x[:min(n, m)], y[:min(n, m)]).to_host()
    assert util.ulp_distance(Gp, ko(x[:min(n, m)], y[:min(n, m)])) <= 2


@pytest.mark.parametrize('d', [2, 3, 5])
def test_gram_ulp_nd(lgp, d, rng):
    dt = [(f'f{i}', float) for i in range(d)]
    x = np.empty(300, dt); y = np.empty(129, dt)
    for f, _t in dt:
        x[f] = rng.uniform(0, 4, 300); y[f] = rng.uniform(0, 4, 129)
    for kd, ko in ((lgp.Maternp(p=2, scale=0.8), oracle.Maternp(p=2, scale=0.8)), (lgp.ExpQuad() * 2.5 + 1, oracle.ExpQuad() * 2.5 + 1)):
        G = kd.gram(x, y).to_host()
        R = ko(x[:, None], y[None, :])
        assert util.ulp_distance(G, R) <= 3
    Gs = lgp.ExpQuad().gram(x, x).to_host()
    np.testing.assert_array_equal(Gs, Gs.T)            # exactly symmetric
    np.testing.assert_array_equal(np.diag(Gs), 1)


def test_gram_f32_and_int(lgp, rng):
    x32 = rng.uniform(-3, 3, 200).astype(np.float32)
    G = lgp.ExpQuad().gram(x32, x32)
    assert G.dtype == np.float32
    R = oracle.ExpQuad()(x32[:, None], x32[None, :])
    assert R.dtype == np.float32
    assert util.ulp_distance(G.to_host(), R) <= 2
    xi = np.arange(20)                                  # integer inputs compute in float64 (x64 mode)
    Gi = lgp.ExpQuad(scale=3).gram(xi, xi)
    assert Gi.dtype == np.float64
    assert util.ulp_distance(Gi.to_host(), oracle.ExpQuad(scale=3)(xi[:, None], xi[None, :])) <= 2


def test_gram_empty(lgp):
    G = lgp.ExpQuad().gram(np.zeros(0), np.arange(3.))
    assert G.shape == (0, 3)
    assert lgp.ExpQuad()(np.zeros((0, 1)), np.arange(3.)[None, :]).shape == (0, 3)


def test_kernel_argument_checks(lgp):
    with pytest.raises(AssertionError):
        lgp.ExpQuad(scale=-1)
    with pytest.raises(AssertionError):
        lgp.ExpQuad(loc=np.inf)
    with pytest.raises(AssertionError):
        lgp.Maternp(p=1.5)
    with pytest.raises(AssertionError):
        lgp.Maternp()


def test_golden_c3(lgp):
    """ Matern-5/2 3-D Gram entries of the committed C3-small fixture inputs """
    g = np.load(os.path.join(HERE, 'golden', 'c3_matern52_n1024.npz'))
    dt = [('f0', float), ('f1', float), ('f2', float)]
    x = np.empty(1024, dt); x['f0'], x['f1'], x['f2'] = g['x0'], g['x1'], g['x2']
    G = lgp.Maternp(p=2, scale=1.).gram(x, x).to_host()
    R = oracle.Maternp(p=2, scale=1.)(x[:, None], x[None, :])
    assert util.ulp_distance(G, R) <= 3


# ---- BART -------------------------------------------------------------------------------

def _struct(ix):
    p = ix.shape[1]
    return np.array([tuple(r) for r in ix], dtype=[(f'f{i}', ix.dtype) for i in range(p)])


BART_CONFIGS = [
    dict(maxd=0), dict(maxd=1), dict(maxd=2), dict(maxd=2, gamma=0.3), dict(maxd=2, reset=[1]), dict(maxd=4, reset=2),
    dict(maxd=10, reset=[2, 4, 6, 8]), dict(maxd=3, reset=[1, 2]), dict(maxd=4, reset=[1, 2, 3]), dict(maxd=2, intercept=False),
    dict(maxd=2, gamma='auto'), dict(maxd=1, gamma='auto'), dict(pnt=[0.9, 0.4, 0.3]), dict(maxd=5, reset=[1, 3]),
]


@pytest.mark.parametrize('cfg', BART_CONFIGS, ids=[str(c) for c in BART_CONFIGS])
@pytest.mark.parametrize('flt', [np.float64, np.float32])
def test_bart_gram_vs_oracle(lgp, cfg, flt, rng):
    n, m, p = 70, 300, 6
    X = rng.integers(0, 9, (n + m, p)).astype(float)
    X[:, 0] = rng.integers(0, 2, n + m)                 # a binary covariate (one split)
    X[:, 1] = 3.                                        # a constant covariate (no splits: n_i = 0)
    splits = lgp.BART.splits_from_coord(X)
    ix = lgp.BART.indices_from_coord(X, splits)
    w = np.array([1, 2, 0, 1, 3, 1], flt)               # one zero-weight axis
    kw = dict(cfg)
    if 'pnt' in kw:
        kw['pnt'] = np.array(kw['pnt'], flt)
    else:
        kw['alpha'] = flt(0.95); kw['beta'] = flt(2)
    if not isinstance(kw.get('gamma', 1), str):
        kw['gamma'] = flt(kw.get('gamma', 1))
    kd = lgp.BART(splits=(splits[0], None), indices=True, weights=w, **kw)
    ko = oracle.BART(splits=(splits[0], None), indices=True, weights=w, **kw)
    G = kd.gram(_struct(ix[:n]), _struct(ix[n:]))
    assert G.dtype == flt
    R = ko(_struct(ix[:n])[:, None], _struct(ix[n:])[None, :])
    assert R.dtype == flt
    tol = 64 if flt == np.float64 else 64
    assert util.ulp_distance(G.to_host(), R) <= tol, np.max(np.abs(G.to_host() - R))
    # from coordinates == from indices, 0 ulp (ref tests/kernels/test_bart.py:355-367)
    kc = lgp.BART(splits=splits, weights=w, **kw)
    Gc = kc.gram(_struct(X[:n]), _struct(X[n:])).to_host()
    np.testing.assert_array_equal(Gc, G.to_host())
    S = kd.gram(_struct(ix[:n]), _struct(ix[:n])).to_host()
    np.testing.assert_array_equal(np.diag(S), 1)
    assert util.ulp_distance(S, S.T) <= 64              # swap invariance (ref :135-147)


def test_bart_golden_c5(lgp):
    g = np.load(os.path.join(HERE, 'golden', 'c5_bart_n256.npz'))
    ix = g['ix']; p = ix.shape[1]
    for name, flt in (('f64', np.float64), ('f32', np.float32)):
        kw = dict(alpha=flt(0.95), beta=flt(2), gamma=flt(1), weights=np.ones(p, flt))
        k = lgp.BART(splits=(g['nsplits'], None), indices=True, maxd=10, reset=[2, 4, 6, 8], **kw)
        G = k.gram(_struct(ix), _struct(ix)).to_host()
        assert util.ulp_distance(G, g['K_' + name]) <= 64
        k2 = lgp.BART(splits=(g['nsplits'], None), indices=True, maxd=2, **kw)
        assert util.ulp_distance(k2.gram(_struct(ix), _struct(ix)).to_host(), g['K2_' + name]) <= 64
        eig = np.linalg.eigvalsh(G.astype(float))
        assert eig[0] >= -len(G) * np.finfo(flt).eps * eig[-1]       # PSD (ref test_kernels.py:163-166)


def test_bart_correlation_classmethod(lgp, rng):
    """ the vectorised BART.correlation on the device against the oracle (both input conventions) """
    p = 5
    sb, sbw, sa = rng.integers(0, 6, (3, 40, p))
    w = rng.integers(1, 10, p).astype(float)
    for md, reset in [(0, None), (1, None), (2, None), (2, [1]), (4, [2]), (3, [1, 2])]:
        for gamma in (0., 0.4, 1.):
            c1 = lgp.BART.correlation(sb, sbw, sa, alpha=0.9, beta=1.6, gamma=gamma, maxd=md, reset=reset, weights=w)
            c2 = oracle.BART.correlation(sb, sbw, sa, alpha=0.9, beta=1.6, gamma=gamma, maxd=md, reset=reset, weights=w)
            assert c1.shape == (40,)
            util.assert_allclose(c1, c2, rtol=1e-14, atol=4e-15)
            n = sb + sbw + sa
            c3 = lgp.BART.correlation(n, sb, sb + sbw, alpha=0.9, beta=1.6, gamma=gamma, maxd=md, reset=reset, weights=w, altinput=True)
            util.assert_allclose(c3, c2, rtol=1e-14, atol=4e-15)
    # reference known answers (tests/kernels/test_bart.py:283-301,313-330,369-372)
    sss = [1, 2], [3, 1], [5, 8]
    c1 = lgp.BART.correlation(*sss, alpha=0.9, beta=1.6, maxd=2)
    c2 = lgp.BART.correlation(*sss, pnt=[0.9, 0.9 / 2 ** 1.6, 0.9 / 3 ** 1.6])
    util.assert_allclose(c1, c2, rtol=1e-15)
    ci = lgp.BART.correlation(*sss, alpha=0.9, intercept=False)
    util.assert_allclose(lgp.BART.correlation(*sss, alpha=0.9), ci * 0.9 + 0.1, rtol=1e-15)
    assert lgp.BART.correlation(*sss, pnt=np.array([0.9, 0.4, 0.3], 'f')).dtype == np.float32
    assert lgp.BART.correlation(*[np.array(s, np.int32) for s in sss], pnt=np.array([0.9, 0.4, 0.3])).dtype == np.float64
    empty = np.array([], int)
    assert lgp.BART.correlation(empty, empty, empty) == 1
    with pytest.raises(KeyError):
        lgp.BART.correlation([0], [0], [0], gamma='ciao')
    # a depth-4 slice: the explicit recursion (slow path) on the device
    util.assert_allclose(lgp.BART.correlation(*sss, maxd=4), oracle.BART.correlation(*sss, maxd=4), rtol=1e-14)
    with pytest.raises(NotImplementedError):
        lgp.BART.correlation(*sss, maxd=12)             # a slice deeper than 10 levels: use `reset`


def test_bart_correlation_vectorised_parameters(lgp, rng):
    """ alpha / beta / gamma / pnt with leading axes of their own, broadcast against the leading axes of the counts
    (``_kernels/_bart.py:373-401``): the device groups the pairs by distinct parameters; the oracle broadcasts like the reference """
    p = 4
    sb, sbw, sa = rng.integers(0, 5, (3, 6, 1, p))
    alpha = np.array([0.95, 0.7, 0.5])
    beta = np.array([[2.], [0.7]]) * np.ones(3)                  # (2, 3)
    c1 = lgp.BART.correlation(sb, sbw, sa, alpha=alpha, beta=2., maxd=2)
    c2 = oracle.BART.correlation(sb, sbw, sa, alpha=alpha, beta=2., maxd=2)
    assert c1.shape == c2.shape == (6, 3)
    util.assert_allclose(c1, c2, rtol=1e-14, atol=4e-15)
    sb2, sbw2, sa2 = rng.integers(0, 5, (3, 5, 1, 1, p))
    gamma = np.array([0., 0.3, 1.])
    c1 = lgp.BART.correlation(sb2, sbw2, sa2, alpha=alpha, beta=beta, gamma=gamma, maxd=3, reset=[1])
    c2 = oracle.BART.correlation(sb2, sbw2, sa2, alpha=alpha, beta=beta, gamma=gamma, maxd=3, reset=[1])
    assert c1.shape == c2.shape == (5, 2, 3)
    util.assert_allclose(c1, c2, rtol=1e-14, atol=4e-15)
    pnt = np.array([[0.9, 0.5, 0.2], [0.8, 0.1, 0.05]])
    c1 = lgp.BART.correlation(sb[:, 0][:2], sbw[:, 0][:2], sa[:, 0][:2], pnt=pnt, gamma='auto')
    c2 = oracle.BART.correlation(sb[:, 0][:2], sbw[:, 0][:2], sa[:, 0][:2], pnt=pnt, gamma='auto')
    assert c1.shape == c2.shape == (2,)
    util.assert_allclose(c1, c2, rtol=1e-14, atol=4e-15)
    assert lgp.BART.correlation(sb, sbw, sa, alpha=alpha.astype('f'), beta=np.float32(2), maxd=1).dtype == np.float32


def test_bart_kernel_object_with_vectorised_parameters(lgp, rng):
    """ alpha / beta / gamma / pnt as arrays on the kernel OBJECT: in the reference they broadcast against the shape of the kernel's
    output (``_kernels/_bart.py:301-303``: "fully vectorized", :373-401), trailing axes first -- one parameter per column of a Gram
    matrix, or per row with a trailing unit axis.  Checked against the scalar-parameter kernels column by column / row by row, and
    against the oracle's kernel object where it broadcasts the same way """
    n, m, p = 9, 7, 4
    X = rng.integers(0, 6, (n + m, p)).astype(float)
    splits = lgp.BART.splits_from_coord(X)
    x, y = _struct(X[:n]), _struct(X[n:])
    alpha = rng.uniform(0.3, 0.95, m)
    k = lgp.BART(splits=splits, alpha=alpha, beta=1.5, maxd=2)
    G = k.gram(x, y).to_host()
    assert G.shape == (n, m) and G.dtype == np.float64
    for j in range(m):
        ref = lgp.BART(splits=splits, alpha=alpha[j], beta=1.5, maxd=2).gram(x, y).to_host()
        np.testing.assert_array_equal(G[:, j], ref[:, j])
    R = oracle.BART(splits=splits, alpha=alpha, beta=1.5, maxd=2)(x[:, None], y[None, :])
    util.assert_allclose(G, R, rtol=1e-14, atol=4e-15)
    # one beta per row, one gamma per column, through a product with a constant and with reset
    beta = rng.uniform(0.5, 3., (n, 1))
    gamma = rng.uniform(0., 1., m)
    k = 3. * lgp.BART(splits=splits, alpha=0.9, beta=beta, gamma=gamma, maxd=4, reset=[2])
    G = k.gram(x, y).to_host()
    for i in range(0, n, 4):
        for j in range(0, m, 3):
            ref = (3. * lgp.BART(splits=splits, alpha=0.9, beta=beta[i, 0], gamma=gamma[j], maxd=4, reset=[2])).gram(x, y).to_host()
            assert G[i, j] == ref[i, j]
    # pnt with a leading axis; element-wise evaluation (`pairs`): one parameter set per pair
    pnt = np.stack([np.array([0.9, 0.5, 0.2]) * f for f in rng.uniform(0.5, 1., n)])
    k = lgp.BART(splits=splits, pnt=pnt, gamma='auto')
    v = k.pairs(x, _struct(X[n - 2:2 * n - 2])).to_host()
    for i in range(n):
        ref = lgp.BART(splits=splits, pnt=pnt[i], gamma='auto').pairs(x[i:i + 1], _struct(X[n - 2 + i:n - 1 + i])).to_host()
        assert v[i] == ref[0]
    # float32 parameters give a float32 kernel; a parameter shape that would ADD axes to the output is refused, so are tangents
    assert lgp.BART(splits=splits, alpha=alpha.astype('f'), beta=np.float32(2)).gram(x, y).dtype == np.float32
    with pytest.raises(NotImplementedError):
        lgp.BART(splits=splits, alpha=np.full((2, 1, 1), 0.9)).gram(x, y)
    with pytest.raises(NotImplementedError):
        lgp.BART(splits=splits, alpha=alpha).gram_jvp(x, y)


def test_bart_corr_one_cases(lgp, rng):
    """ correlation = 1 when the points coincide, all weights vanish or there are no covariates (ref :105-133) """
    p = 4
    ix = rng.integers(0, 5, (30, p)); n = ix + rng.integers(0, 3, (30, p))
    for md in (0, 1, 2):
        c = lgp.BART.correlation(n, ix, ix, maxd=md, gamma=0.2, altinput=True)
        np.testing.assert_array_equal(c, 1)
        c = lgp.BART.correlation(n, ix, np.minimum(ix + 1, n), maxd=md, gamma=0.2, weights=np.zeros(p), altinput=True)
        np.testing.assert_array_equal(c, 1)


# ---- derivative functors (the tangents jax.jacfwd produces in the reference's fit) -------------------

def _fd(f, h):
    return (f(+h) - f(-h)) / (2 * h)


@pytest.mark.parametrize('make', [
    lambda lgp_, s: lgp_.ExpQuad(scale=s),
    lambda lgp_, s: lgp_.Maternp(p=2, scale=s),
    lambda lgp_, s: lgp_.Maternp(p=0, scale=s),
    lambda lgp_, s: lgp_.Maternp(p=5, scale=s) * 3.5,
    lambda lgp_, s: lgp_.GammaExp(gamma=1.3, scale=s),
    lambda lgp_, s: lgp_.GammaExp(gamma=2, scale=s) * 0.5 + 1.0,
    lambda lgp_, s: lgp_.Cauchy(alpha=1.5, beta=2.5, scale=s),
    lambda lgp_, s: lgp_.Cauchy(alpha=2, beta=1, scale=s),
])
@pytest.mark.parametrize('d', [1, 3, 5])
def test_gram_dscale_vs_finite_differences_of_the_oracle(lgp, make, d, rng):
    """ lgp_gram_isotropic_dscale: K bit-identical to the plain Gram kernel, dK/dscale against central differences of the
    ORACLE's kernel in the scale (same kernel classes on both sides) """
    n, m, s0 = 70, 45, 1.7
    if d == 1:
        x, y = rng.uniform(-3, 3, n), rng.uniform(-3, 3, m)
    else:
        dt = [(f'f{i}', float) for i in range(d)]
        x, y = np.empty(n, dt), np.empty(m, dt)
        for nm in x.dtype.names:
            x[nm] = rng.uniform(-3, 3, n)
            y[nm] = rng.uniform(-3, 3, m)
    k = make(lgp, s0)
    K, dK = k.gram_dscale(x, y)
    np.testing.assert_array_equal(K.to_host(), k.gram(x, y).to_host())
    fd = _fd(lambda h: make(oracle, s0 + h)(x[:, None], y[None, :]), 1e-5)
    got = dK.to_host()
    assert np.max(np.abs(got - fd)) <= 2e-9 * max(1.0, np.max(np.abs(fd)))
    # symmetric block: the diagonal of the derivative is zero
    K2, dK2 = k.gram_dscale(x, x)
    assert np.all(np.diag(dK2.to_host()) == 0)


def test_gram_dscale_float32(lgp, rng):
    x = rng.uniform(-3, 3, 50).astype(np.float32)
    k = lgp.Maternp(p=2, scale=np.float32(1.3))
    K, dK = k.gram_dscale(x, x)
    assert K.dtype == np.float32 and dK.dtype == np.float32
    fd = _fd(lambda h: oracle.Maternp(p=2, scale=1.3 + h)(x.astype(float)[:, None], x.astype(float)[None, :]), 1e-5)
    assert np.max(np.abs(dK.to_host() - fd)) <= 2e-5


@pytest.mark.parametrize('flt', [np.float64, np.float32])
@pytest.mark.parametrize('kw', [dict(maxd=2), dict(maxd=1), dict(maxd=0), dict(maxd=4, reset=2), dict(maxd=10, reset=[2, 4, 6, 8]),
                                dict(maxd=6, reset=[1, 2, 4]), dict(maxd=2, intercept=False), dict(maxd=3, reset=[1], gamma=0.7)])
def test_bart_gram_jvp_vs_finite_differences_of_the_oracle(lgp, kw, flt, rng):
    """ lgp_gram_bart_jvp: value identical to lgp_gram_bart, tangents with respect to alpha and beta against central
    differences of the oracle's float64 kernel (every recursion-slice layout the device path supports) """
    n, p = 60, 4
    X = np.concatenate([rng.integers(0, 2, (n, 1)), rng.integers(0, 25, (n, p - 1))], axis=1).astype(float)
    splits = oracle.BART.splits_from_coord(X)
    ix = oracle.BART.indices_from_coord(X, splits).astype(np.int32)
    pts = np.array([tuple(r) for r in ix], dtype=[(f'f{i}', np.int32) for i in range(p)])
    a0, b0 = 0.9, 1.6
    kw = dict(kw)
    kw.setdefault('gamma', flt(1))
    if not isinstance(kw['gamma'], np.generic):
        kw['gamma'] = flt(kw['gamma'])

    def mk(mod, a, b, f):
        return mod.BART(splits=(splits[0], None), indices=True, alpha=f(a), beta=f(b), weights=np.ones(p, f), **kw) * f(2.5)

    k = mk(lgp, a0, b0, flt)
    K, (dKa, dKb) = k.gram_jvp(pts, pts)
    np.testing.assert_array_equal(K.to_host(), k.gram(pts, pts).to_host())
    kw64 = dict(kw, gamma=np.float64(kw['gamma']))

    def ko(a, b):
        return oracle.BART(splits=(splits[0], None), indices=True, alpha=np.float64(a), beta=np.float64(b), weights=np.ones(p), **kw64)(pts[:, None], pts[None, :]) * 2.5

    fa = _fd(lambda h: ko(a0 + h, b0), 1e-6)
    fb = _fd(lambda h: ko(a0, b0 + h), 1e-6)
    tol = 1e-8 if flt == np.float64 else 2e-5
    assert np.max(np.abs(dKa.to_host() - fa)) <= tol * max(1.0, np.max(np.abs(fa)))
    assert np.max(np.abs(dKb.to_host() - fb)) <= tol * max(1.0, np.max(np.abs(fb)))
    assert np.all(np.diag(dKa.to_host()) == 0) and np.all(np.diag(dKb.to_host()) == 0)      # k(x, x) = const


@pytest.mark.parametrize('maxd,reset', [(3, None), (4, None), (5, None), (5, [1]), (6, [3]), (4, [1])])
@pytest.mark.parametrize('flt', [np.float64, np.float32])
def test_bart_deep_recursion(lgp, maxd, reset, flt, rng):
    """ recursion slices deeper than two levels (reference _kernels/_bart.py:759-806: explicit recursion over the splits that do not
    separate the two points) on the device -- depth-first search with an explicit stack, the closed form of the last three
    levels at the bottom -- against the oracle's restatement of the same lines, at the reference's own tolerance for shortcuts
    against the full recursion (tests/kernels/test_bart.py:226-235: 32 ulp); Gram form and vectorised `correlation` """
    p = 3
    sb, sbw, sa = rng.integers(0, 3, (3, 12, p))
    w = np.array([1., 2., 0.5])
    kw = dict(alpha=flt(0.9), beta=flt(1.3), gamma=flt(0.7), maxd=maxd, reset=reset, weights=w.astype(flt))
    c1 = lgp.BART.correlation(sb, sbw, sa, **kw)
    c2 = oracle.BART.correlation(sb, sbw, sa, **kw)
    assert c1.dtype == flt and c1.shape == (12,)
    assert util.ulp_distance(c1, c2) <= 32, (c1, c2)
    if maxd <= 4 and reset is None:       # ... and the recursion all the way down (debug=True of the reference), float64 oracle
        c3 = oracle.BART.correlation(sb, sbw, sa, debug=True, **{**kw, 'alpha': 0.9, 'beta': 1.3, 'gamma': 0.7, 'weights': w})
        util.assert_allclose(c1, c3, rtol=2e-5 if flt == np.float32 else 1e-13)
    # Gram form from coordinates: a kernel object with deep slices
    X = rng.integers(0, 4, (14, p)).astype(float)
    splits = lgp.BART.splits_from_coord(X)
    kd = lgp.BART(splits=splits, **kw)
    ko = oracle.BART(splits=splits, **kw)
    G = kd.gram(_struct(X[:6]), _struct(X[6:])).to_host()
    R = ko(_struct(X[:6])[:, None], _struct(X[6:])[None, :])
    assert G.dtype == flt and util.ulp_distance(G, R) <= 32
    S = kd.gram(_struct(X), _struct(X)).to_host()
    np.testing.assert_array_equal(np.diag(S), 1)
    assert util.ulp_distance(S, S.T) <= 64
    with pytest.raises(NotImplementedError):
        kd.gram_jvp(_struct(X), _struct(X))


@pytest.mark.parametrize('make,d', [
    (lambda m: m.Wendland(k=0, alpha=1, scale=2.), 1), (lambda m: m.Wendland(k=1, alpha=2, scale=3.), 2), (lambda m: m.Wendland(k=2, alpha=2.5, scale=4.), 3),
    (lambda m: m.Wendland(k=3, alpha=2, scale=2.5, loc=0.3), 2), (lambda m: m.CausalExpQuad(alpha=1, scale=1.5), 1),
    (lambda m: m.CausalExpQuad(alpha=0, scale=1.1), 3), (lambda m: m.CausalExpQuad(alpha=3.7), 2), (lambda m: m.Log(scale=0.7), 1), (lambda m: 2.5 * m.Log() + 1, 3),
], ids=['wendland0', 'wendland1', 'wendland2', 'wendland3', 'causal1', 'causal0', 'causal3.7', 'log', 'log3d'])
@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_posabs_kernels_vs_oracle(lgp, make, d, dtype, rng):
    """ the isotropic kernels with input='posabs' (per field (|x - y| + eps)^2, _Kernel/_isotropic.py:67-78): Wendland
    (_kernels/_wendland.py:32-84, compact support), CausalExpQuad (_kernels/_basic.py:346-360), Log (:381-392) """
    n, m = 150, 270
    dt = [(f'f{i}', dtype) for i in range(d)]
    x = np.empty(n, dt); y = np.empty(m, dt)
    for f in x.dtype.names:
        x[f] = rng.uniform(-2, 2, n).astype(dtype); y[f] = rng.uniform(-2, 2, m).astype(dtype)
    y[:20] = x[:20]
    if d == 1:
        x, y = x['f0'], y['f0']
    G = make(lgp).gram(x, y).to_host()
    R = make(oracle)(x[:, None], y[None, :])
    assert G.dtype == dtype
    tol = 2e-14 if dtype == np.float64 else 2e-5        # (erfc / log1p / pow of the device library vs libm)
    np.testing.assert_allclose(G, R, rtol=0, atol=tol * np.abs(R).max())
    # the elementwise form and symmetry / unit diagonal of the correlation kernels
    P = make(lgp)(x[:m if m < n else n], y[:n if n < m else m])
    np.testing.assert_allclose(P, np.diagonal(R)[:len(P)], rtol=0, atol=tol * np.abs(R).max())


def test_wendland_dimension_limit(lgp):
    x = np.zeros(4, [('a', float), ('b', float)])
    with pytest.raises(ValueError):
        lgp.Wendland(k=0, alpha=1).gram(x, x)             # maxdim = floor(2 alpha - 1) = 1
    with pytest.raises(NotImplementedError):
        lgp.Wendland(k=4)


@pytest.mark.parametrize('n', [512, 513, 1000, 2048 + 37])
@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_one_triangle_gram_is_the_full_gram(n, dtype):
    """ the same array on both sides: tiles on / left of the diagonal evaluated, the rest mirrored (`symmetric` of the C-ABI,
    reference `halfmatrix` _elements.py:567-573) -- bit for bit the matrix of the full evaluation, for every plain core """
    import lsqfitgp_amd as lgp
    rng = np.random.default_rng(n)
    for d in (1, 3, 5):
        x = np.empty(n, [(f'f{i}', dtype) for i in range(d)]) if d > 1 else rng.uniform(0, 20, n).astype(dtype)
        if d > 1:
            for i in range(d):
                x[f'f{i}'] = rng.uniform(0, 5, n)
        kernels = [lgp.ExpQuad(scale=1.3), lgp.Maternp(p=2, scale=0.7) * 1.5 + 0.1, lgp.Maternp(p=1), lgp.Maternp(p=3, scale=2.), lgp.Maternp(p=0),
                   lgp.GammaExp(gamma=1.2), lgp.Cauchy(alpha=1.5, beta=2.5, scale=3.)]
        if d <= 3:
            kernels.append(lgp.Wendland(k=1, alpha=2, scale=4.))
        for k in kernels:
            half = k.gram(x, x).to_host()                # xs is ys: one triangle
            full = k.gram(x, x.copy()).to_host()         # two arrays: every tile
            np.testing.assert_array_equal(half, full)
            np.testing.assert_array_equal(half, half.T)


def test_one_triangle_gram_forced_for_every_core_and_dtype():
    """ by default the one-triangle grid is used where it pays (fp64, cores heavier than ExpQuad); LGP_GRAM_HALF=1 forces it
    everywhere (read once per process, hence the child process): still the full grid's matrix bit for bit """
    import os, subprocess, sys
    code = '''
import numpy as np, lsqfitgp_amd as lgp
rng = np.random.default_rng(5)
for dtype in (np.float64, np.float32):
    for n in (1024, 1500):
        for d in (1, 3):
            x = np.empty(n, [(f"f{i}", dtype) for i in range(d)]) if d > 1 else rng.uniform(0, 20, n).astype(dtype)
            if d > 1:
                for i in range(d):
                    x[f"f{i}"] = rng.uniform(0, 5, n)
            for k in (lgp.ExpQuad(scale=1.3), lgp.Maternp(p=2) * 2. + 1., lgp.GammaExp(gamma=1.2)):
                half = k.gram(x, x).to_host()
                full = k.gram(x, x.copy()).to_host()
                assert np.array_equal(half, full) and np.array_equal(half, half.T), (dtype, n, d, type(k).__name__)
print("ok")
'''
    env = dict(os.environ, LGP_GRAM_HALF='1')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, '-c', code], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.strip().endswith('ok'), res.stdout + res.stderr


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('make', [
    lambda m: m.Constant(), lambda m: 2. * m.Constant(scale=3.) + 1., lambda m: m.White(), lambda m: m.White(scale=0.3, loc=1.) * 4.,
    lambda m: m.Expon(scale=2.), lambda m: m.HoleEffect(scale=0.5), lambda m: m.HoleEffect(loc=0.2) * 3. + 0.5,
    lambda m: m.Periodic(outerscale=0.7, scale=1.5), lambda m: m.Periodic(), lambda m: m.Cos(scale=0.8, loc=3.), lambda m: m.Sinc(scale=0.3), lambda m: m.Sinc(),
])
def test_simple_stationary_kernels_vs_oracle(lgp, make, dtype, rng):
    """ Constant, White (any fields) and the stationary kernels of one variable Expon, HoleEffect, Periodic, Cos, Sinc
    (``_kernels/_basic.py``, ``_spectral.py``) as cores of the generic Gram kernel """
    n, m = 170, 300
    x = rng.uniform(-4, 4, n).astype(dtype); y = rng.uniform(-4, 4, m).astype(dtype)
    y[:25] = x[:25]
    y[25:40] = x[25:40] + dtype(0.01)                     # (the Taylor branch of sinc)
    G = make(lgp).gram(x, y).to_host()
    R = np.asarray(make(oracle)(x[:, None], y[None, :]), float)
    assert G.dtype == dtype
    tol = 4e-15 if dtype == np.float64 else 4e-6           # (sin / cos / exp of the device library against libm, arguments up to 27)
    np.testing.assert_allclose(G, R, rtol=0, atol=tol * max(1., np.abs(R).max()))
    P = make(lgp)(x[:m if m < n else n], y[:n if n < m else m])
    np.testing.assert_allclose(P, np.diagonal(R)[:len(P)], rtol=0, atol=tol * max(1., np.abs(R).max()))


def test_white_and_constant_on_structured_points_and_dimension_limits(lgp):
    a = np.zeros(6, [('u', float), ('v', float)]); a['u'] = [0, 0, 1, 1, 2, 2]; a['v'] = [0, 1, 0, 1, 1e-300, 0]
    np.testing.assert_array_equal(lgp.White().gram(a, a).to_host(), np.asarray(oracle.White()(a[:, None], a[None, :]), float))
    np.testing.assert_array_equal(lgp.Constant().gram(a, a).to_host(), np.ones((6, 6)))
    for cls in (lgp.Expon, lgp.HoleEffect, lgp.Periodic, lgp.Cos, lgp.Sinc):
        with pytest.raises(ValueError):
            cls().gram(a, a)
    with pytest.raises(AssertionError):
        lgp.Periodic(outerscale=0)
    # a GP on a periodic kernel: the posterior repeats with the period
    x = np.linspace(0, 6, 40)
    gp = lgp.GP(lgp.Periodic(scale=1 / (2 * np.pi), outerscale=0.5), checkpos=False).addx(x, 'd').addx(x + 1., 'shifted')
    m1, _ = gp.predfromdata({'d': np.sin(2 * np.pi * x)}, 'shifted', raw=True)
    np.testing.assert_allclose(m1, np.sin(2 * np.pi * x), atol=1e-6)


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('make', [
    lambda m: m.Wiener(), lambda m: m.Wiener(scale=0.5) * 2. + 1., lambda m: m.OrnsteinUhlenbeck(scale=0.5), lambda m: m.BrownianBridge(scale=2.),
    lambda m: m.Decaying(alpha=1.5), lambda m: m.Decaying(alpha=0), lambda m: m.FracBrownian(), lambda m: m.FracBrownian(H=0.8, K=0.6, loc=0.5),
    lambda m: m.WienerIntegral(scale=0.7), lambda m: m.StationaryFracBrownian(H=0.3, scale=0.4), lambda m: m.StationaryFracBrownian(), lambda m: m.Linear(scale=2., loc=0.3),
])
def test_one_variable_kernels_vs_oracle(lgp, make, dtype, rng):
    """ the kernels of x and y themselves (``_kernels/_randomwalk.py``, Decaying, Linear): the generic Gram kernel hands the
    core the two transformed coordinates """
    n, m = 170, 300
    x = rng.uniform(0, 1, n).astype(dtype); y = rng.uniform(0, 1, m).astype(dtype)
    y[:25] = x[:25]
    G = make(lgp).gram(x, y).to_host()
    R = np.asarray(make(oracle)(x[:, None], y[None, :]), float)
    assert G.dtype == dtype
    tol = 4e-15 if dtype == np.float64 else 4e-6
    np.testing.assert_allclose(G, R, rtol=0, atol=tol * max(1., np.abs(R).max()))
    P = make(lgp)(x[:m if m < n else n], y[:n if n < m else m])
    np.testing.assert_allclose(P, np.diagonal(R)[:len(P)], rtol=0, atol=tol * max(1., np.abs(R).max()))


def test_linear_over_fields_and_domains(lgp, rng):
    for d in (2, 3, 6):
        a = np.empty(120, [(f'f{i}', float) for i in range(d)]); b = np.empty(90, a.dtype)
        for f in a.dtype.names:
            a[f] = rng.standard_normal(120); b[f] = rng.standard_normal(90)
        k = lgp.Linear(scale=1.5) * 2.
        np.testing.assert_allclose(k.gram(a, b).to_host(), (oracle.Linear(scale=1.5) * 2.)(a[:, None], b[None, :]), rtol=0, atol=1e-14)
    x = np.linspace(-1, 1, 8)
    for cls in (lgp.Wiener, lgp.OrnsteinUhlenbeck, lgp.BrownianBridge, lgp.Decaying, lgp.WienerIntegral):
        with pytest.raises(AssertionError):
            cls().gram(x, x)
        with pytest.raises(ValueError):
            cls().gram(np.zeros(3, [('a', float), ('b', float)]), np.zeros(3, [('a', float), ('b', float)]))
    with pytest.raises(AssertionError):
        lgp.BrownianBridge().gram(np.array([0.5, 1.5]), np.array([0.5]))
    lgp.FracBrownian(H=0.3).gram(x, x)                   # defined on the whole line
    # a Wiener process pinned at three points: the posterior mean interpolates linearly between them
    gp = lgp.GP(lgp.Wiener(), checkpos=False).addx(np.array([1., 2., 4.]), 'd').addx(np.array([1.5, 3.]), 'm')
    m, _ = gp.predfromdata({'d': np.array([1., 3., -1.])}, 'm', raw=True)
    np.testing.assert_allclose(m, [2., 1.], atol=1e-9)


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('make', [
    lambda m: m.Celerite(), lambda m: m.Celerite(gamma=0.7, B=0.7, scale=0.5), lambda m: m.Celerite(gamma=2., B=-1.5) * 2.,
    lambda m: m.Harmonic(Q=0.2), lambda m: m.Harmonic(Q=0.7, scale=0.5), lambda m: m.Harmonic(), lambda m: m.Harmonic(Q=5., scale=0.3),
    lambda m: m.Circular(), lambda m: m.Circular(tau=6, c=0.3, scale=2.), lambda m: m.Circular(scale=0.25, loc=1.) + 1.,
])
def test_oscillator_and_circular_kernels_vs_oracle(lgp, make, dtype, rng):
    """ Celerite, Harmonic (four branches in Q) and Circular (``_kernels/_celerite.py``, ``_wendland.py:88-113``) """
    n, m = 170, 300
    x = rng.uniform(-4, 4, n).astype(dtype); y = rng.uniform(-4, 4, m).astype(dtype)
    y[:25] = x[:25]
    G = make(lgp).gram(x, y).to_host()
    R = np.asarray(make(oracle)(x[:, None], y[None, :]), float)
    tol = 1e-14 if dtype == np.float64 else 2e-5           # (Circular: a rounding of delta mod 1 times tau / c)
    np.testing.assert_allclose(G, R, rtol=0, atol=tol * max(1., np.abs(R).max()))


def test_harmonic_limits(lgp, rng):
    """ Q = 1 with scale = sqrt(1/3) is Matern-3/2 in one dimension; B = gamma makes Celerite the Harmonic kernel with
    eta Q = 1 / B (the remarks in the reference's docstrings) """
    x = rng.uniform(0, 3, 200)
    H = lgp.Harmonic(scale=np.sqrt(1 / 3)).gram(x, x).to_host()
    M = lgp.Maternp(p=1).gram(x, x).to_host()
    np.testing.assert_allclose(H, M, rtol=0, atol=1e-14)
    Q = 3.
    eta = np.sqrt(1 - 1 / Q ** 2)
    B = 1 / (eta * Q)
    np.testing.assert_allclose(lgp.Celerite(gamma=B, B=B, scale=1 / eta).gram(x, x).to_host(), lgp.Harmonic(Q=Q).gram(x, x).to_host(), rtol=0, atol=2e-14)


def test_sums_and_products_of_kernels(lgp, rng):
    """ k1 + k2 and k1 * k2 (``_Kernel/_alg.py:32-82``): every operand evaluated by its own Gram kernel on the same points, the
    matrices combined on the device; scalar operations on the composite; mixed dtypes; nesting """
    n, m = 210, 130
    x = rng.uniform(0, 6, n); y = rng.uniform(0, 6, m)
    make = lambda M: (M.ExpQuad(scale=2.) * M.Periodic(outerscale=0.5, scale=0.3) + M.Constant() * 0.3 + M.Maternp(p=1, scale=0.7) * M.Cos(scale=0.5)) * 2. + 0.1
    G = make(lgp).gram(x, y).to_host()
    R = make(oracle)(x[:, None], y[None, :])
    np.testing.assert_allclose(G, R, rtol=0, atol=1e-14 * np.abs(R).max())
    P = make(lgp)(x[:m], y)
    np.testing.assert_allclose(P, np.diagonal(R), rtol=0, atol=1e-14 * np.abs(R).max())
    # structured points, a float32 operand next to float64 ones
    s = np.empty(n, [('a', float), ('b', float)]); t = np.empty(m, s.dtype)
    for f in 'ab':
        s[f] = rng.uniform(0, 3, n); t[f] = rng.uniform(0, 3, m)
    make2 = lambda M: M.ExpQuad(scale=1.5) * M.Linear() + M.Maternp(p=2) + M.White()
    np.testing.assert_allclose(make2(lgp).gram(s, t).to_host(), make2(oracle)(s[:, None], t[None, :]), rtol=0, atol=2e-14)
    K = make2(lgp).gram(s, s).to_host()
    np.testing.assert_array_equal(K, K.T)
    # a GP on a sum of kernels: the prior covariance is the sum of the priors
    gp = lgp.GP(lgp.ExpQuad(scale=2.) + lgp.Cos() * 0.5, checkpos=False).addx(x, 'x')
    np.testing.assert_allclose(gp.prior('x', raw=True), (oracle.ExpQuad(scale=2.) + oracle.Cos() * 0.5)(x[:, None], x[None, :]), rtol=0, atol=1e-14)
    # derivative observations pass through a sum term by term, through a product by Leibniz' rule (tests/test_gpu_deriv.py holds the
    # latter to the oracle and to central differences)
    d1 = (lgp.ExpQuad(scale=2.) + lgp.ExpQuad(scale=0.5) * 3.).gram(x, y, deriv=(1, 1)).to_host()
    want = lgp.ExpQuad(scale=2.).gram(x, y, deriv=(1, 1)).to_host() + 3. * lgp.ExpQuad(scale=0.5).gram(x, y, deriv=(1, 1)).to_host()
    np.testing.assert_allclose(d1, want, rtol=0, atol=1e-14 * np.abs(want).max())
    e, c = lgp.ExpQuad(), lgp.Cos()
    d2 = (e * c).gram(x, y, deriv=(1, 0)).to_host()
    want = e.gram(x, y, deriv=(1, 0)).to_host() * c.gram(x, y).to_host() + e.gram(x, y).to_host() * c.gram(x, y, deriv=(1, 0)).to_host()
    np.testing.assert_allclose(d2, want, rtol=0, atol=1e-14 * np.abs(want).max())


def test_tangents_of_sums_and_products(lgp, rng):
    """ the hyperparameter tangents of composite kernels: linearity and the product rule, against central differences """
    D = lgp.Dual
    x = rng.uniform(0, 6, 150); y = rng.uniform(0, 6, 110)

    def build(s1, s2, a, c):
        return (lgp.ExpQuad(scale=s1) * lgp.Maternp(p=1, scale=s2) * a + lgp.ExpQuad(scale=s2) + c) * a

    p0 = dict(s1=1.7, s2=0.9, a=1.3, c=0.2)
    K, tans = build(**{k: D(v, {k: 1.}) for k, v in p0.items()}).gram_tangents(x, y)
    np.testing.assert_array_equal(K.to_host(), build(**p0).gram(x, y).to_host())
    h = 1e-6
    for key in p0:
        up, dn = dict(p0), dict(p0)
        up[key] += h; dn[key] -= h
        fd = (build(**up).gram(x, y).to_host() - build(**dn).gram(x, y).to_host()) / (2 * h)
        np.testing.assert_allclose(tans[key].to_host(), fd, rtol=3e-7, atol=3e-9, err_msg=key)


def test_taylor_rescaling_categorical(lgp, rng):
    """ Taylor (I0 / J0 of 2 sqrt|xy|, ``_kernels/_basic.py:393-408``), Rescaling (a user function of the points evaluated on the
    host, ``:231-252``) and Categorical (a covariance table, ``:210-229``) against the oracle; Rescaling as a factor of a product """
    x = rng.uniform(-2, 2, 150); y = rng.uniform(-2, 2, 90)
    y[:10] = x[:10]
    for dtype, tol in ((np.float64, 1e-13), (np.float32, 1e-5)):
        G = lgp.Taylor(scale=0.7).gram(x.astype(dtype), y.astype(dtype)).to_host()
        R = oracle.Taylor(scale=0.7)(x.astype(dtype)[:, None], y.astype(dtype)[None, :])
        np.testing.assert_allclose(G, R, rtol=tol, atol=tol)
    std = lambda v: 1 + 0.5 * np.sin(v)
    make = lambda M: M.ExpQuad(scale=1.5) * M.Rescaling(stdfun=std, loc=0.2) * 2. + M.Rescaling() * 0.1
    np.testing.assert_allclose(make(lgp).gram(x, y).to_host(), make(oracle)(x[:, None], y[None, :]), rtol=0, atol=1e-14)
    np.testing.assert_allclose(make(lgp)(x[:90], y), np.diagonal(make(oracle)(x[:, None], y[None, :])), rtol=0, atol=1e-14)
    s = np.empty(40, [('a', float), ('b', float)]); s['a'] = rng.uniform(0, 1, 40); s['b'] = rng.uniform(0, 1, 40)
    f2 = lambda p: p['a'] + 2 * p['b']
    np.testing.assert_allclose(lgp.Rescaling(stdfun=f2, scale=2.).gram(s, s).to_host(), oracle.Rescaling(stdfun=f2, scale=2.)(s[:, None], s[None, :]), rtol=1e-14)
    with pytest.raises(ValueError):
        lgp.Rescaling(stdfun=lambda v: v[:3]).gram(x, y)
    a = rng.standard_normal((5, 5)); cov = a @ a.T
    i = rng.integers(0, 5, 200); j = rng.integers(0, 5, 120)
    np.testing.assert_allclose(lgp.Categorical(cov=cov).gram(i, j).to_host(), cov[i[:, None], j[None, :]], rtol=1e-15)
    np.testing.assert_allclose((lgp.Categorical(cov=cov) * 2.)(i[:120], j), 2 * cov[i[:120], j], rtol=1e-15)
    with pytest.raises(AssertionError):
        lgp.Categorical(cov=cov).gram(x, y)                    # not integers
    with pytest.raises(IndexError):
        lgp.Categorical(cov=cov).gram(np.array([7]), j)


def test_nnkernel_vs_oracle(lgp, rng):
    """ NNKernel (``_kernels/_basic.py:122-154``) as a composite of Linear, a host-evaluated outer product and the device arcsine """
    x = rng.uniform(-3, 3, 160); y = rng.uniform(-3, 3, 110)
    for kw in (dict(), dict(sigma0=0.4, scale=2., loc=0.5)):
        np.testing.assert_allclose(lgp.NNKernel(**kw).gram(x, y).to_host(), oracle.NNKernel(**kw)(x[:, None], y[None, :]), rtol=0, atol=2e-15)
        np.testing.assert_allclose((lgp.NNKernel(**kw) * 3. + 1.)(x[:110], y), 3 * np.diagonal(oracle.NNKernel(**kw)(x[:, None], y[None, :])) + 1, rtol=0, atol=1e-14)
    s = np.empty(90, [('a', float), ('b', float, (2,))])
    s['a'] = rng.standard_normal(90); s['b'] = rng.standard_normal((90, 2))
    np.testing.assert_allclose(lgp.NNKernel(sigma0=2.).gram(s, s).to_host(), oracle.NNKernel(sigma0=2.)(s[:, None], s[None, :]), rtol=0, atol=2e-15)


def test_gibbs_vs_oracle(lgp, rng):
    """ Gibbs (``_kernels/_basic.py:156-185``): the per-point scales evaluated on the host ride as one more coordinate row """
    x = rng.uniform(-3, 3, 170); y = rng.uniform(-3, 3, 120)
    y[:15] = x[:15]
    sf = lambda v: 0.5 + 0.3 * np.abs(v)
    for kw in (dict(), dict(scalefun=sf), dict(scalefun=sf, scale=2., loc=0.5)):
        G = lgp.Gibbs(**kw).gram(x, y).to_host()
        np.testing.assert_allclose(G, oracle.Gibbs(**kw)(x[:, None], y[None, :]), rtol=0, atol=2e-15)
        np.testing.assert_allclose((lgp.Gibbs(**kw) * 2.)(x[:120], y), 2 * np.diagonal(oracle.Gibbs(**kw)(x[:, None], y[None, :])), rtol=0, atol=4e-15)
    np.testing.assert_allclose(lgp.Gibbs().gram(x, y).to_host(), lgp.ExpQuad().gram(x, y).to_host(), rtol=0, atol=2e-16)       # s = 1: the Gaussian kernel
    s = np.empty(140, [('a', float), ('b', float), ('c', float), ('d', float)])
    for f in s.dtype.names:
        s[f] = rng.uniform(-1, 1, 140)
    sf4 = lambda p: 0.7 + p['a'] ** 2
    K = lgp.Gibbs(scalefun=sf4).gram(s, s).to_host()
    np.testing.assert_allclose(K, oracle.Gibbs(scalefun=sf4)(s[:, None], s[None, :]), rtol=0, atol=2e-15)
    np.testing.assert_array_equal(K, K.T)
    with pytest.raises(AssertionError):
        lgp.Gibbs(scalefun=lambda v: v).gram(x, y)          # not positive
    # as an operand of a product, and under a GP
    gp = lgp.GP(lgp.Gibbs(scalefun=sf) * lgp.Constant() * 1.5, checkpos=False).addx(x, 'x')
    np.testing.assert_allclose(gp.prior('x', raw=True), 1.5 * oracle.Gibbs(scalefun=sf)(x[:, None], x[None, :]), rtol=0, atol=4e-15)


def test_dim_selects_fields(lgp, rng):
    """ ``dim=`` (``_Kernel/_ops.py:233-260``): a kernel acts on one field or a list of fields of structured points -- the way
    separable kernels over mixed inputs are written, with products and sums on top; ``maxdim`` counts the selected fields """
    n, m = 140, 90
    dt = [('t', float), ('c', int), ('v', float, (2,)), ('w', float)]
    x = np.zeros(n, dt); y = np.zeros(m, dt)
    for a, k in ((x, n), (y, m)):
        a['t'] = rng.uniform(0, 5, k); a['c'] = rng.integers(0, 3, k); a['v'] = rng.uniform(-1, 1, (k, 2)); a['w'] = rng.uniform(0, 1, k)
    y[:10] = x[:10]
    b = rng.standard_normal((3, 3)); cov = b @ b.T
    make = lambda M: (M.ExpQuad(dim='t', scale=2.) * M.Categorical(cov=cov, dim='c') + M.Maternp(p=1, dim='v') * M.Expon(dim='w', scale=0.5)
                      + M.GammaExp(gamma=1.3, dim=['t', 'w'], scale=1.5) * 0.3)
    G = make(lgp).gram(x, y).to_host()
    R = make(oracle)(x[:, None], y[None, :])
    np.testing.assert_allclose(G, R, rtol=0, atol=1e-14 * np.abs(R).max())
    np.testing.assert_allclose(make(lgp)(x[:m], y), np.diagonal(R), rtol=0, atol=1e-14 * np.abs(R).max())
    # one field of many satisfies maxdim = 1; all of them do not
    lgp.Cos(dim='w').gram(x, y)
    with pytest.raises(ValueError):
        lgp.Cos().gram(x, y)
    with pytest.raises(ValueError):
        lgp.ExpQuad(dim='t').gram(x['t'], y['t'])              # not structured
    with pytest.raises(TypeError):
        lgp.ExpQuad(dim=3)
    # under a GP
    gp = lgp.GP(lgp.ExpQuad(dim='t', scale=2.) * lgp.Categorical(cov=cov, dim='c'), checkpos=False).addx(x, 'x')
    np.testing.assert_allclose(gp.prior('x', raw=True), (oracle.ExpQuad(dim='t', scale=2.) * oracle.Categorical(cov=cov, dim='c'))(x[:, None], x[None, :]), rtol=0, atol=1e-14)


def test_matern_half_integer_orders(lgp, rng):
    """ `Matern(nu=p + 1/2)` is `Maternp(p)` (reference ``tests/kernels/test_kernels.py:501-512``) and agrees with the Bessel-function
    form of the oracle """
    x = rng.uniform(-3, 3, 120); y = rng.uniform(-3, 3, 80)
    for p in range(4):
        G = lgp.Matern(nu=p + 0.5, scale=1.3).gram(x, y).to_host()
        np.testing.assert_array_equal(G, lgp.Maternp(p=p, scale=1.3).gram(x, y).to_host())
        np.testing.assert_allclose(G, oracle.Matern(nu=p + 0.5, scale=1.3)(x[:, None], y[None, :]), rtol=1e-12, atol=1e-15)


def _fields(a):
    """ (n, d) coordinates as a structured array with one array field """
    s = np.empty(len(a), dtype=[('v', float, (a.shape[1],))])
    s['v'] = a
    return s


def _matern_core(nu, r2):
    """ ``_kernels/_matern.py:73-76`` with scipy's K_nu """
    from scipy import special
    x = np.sqrt(2 * nu * r2)
    with np.errstate(all='ignore'):
        v = 2 ** (1 - nu) / special.gamma(nu) * x ** nu * special.kv(nu, x)
    return np.where(x == 0, 1., v)


@pytest.mark.parametrize('nu', [0.05, 0.3, 0.49999, 0.50001, 0.75, 1.0, 1.2, 1.999, 2.0, 2.3, 3.7, 7.1, 10.0, 33.3, 100.0])
def test_matern_real_order_against_scipy_bessel(lgp, rng, nu):
    """ ``LGP_KERNEL_MATERN``: the device's K_nu (Temme series below x = 2, continued fraction above, upward recurrence) against the
    oracle's ``scipy.special.kv`` form of ``_kernels/_matern.py:56-76``: 1e-12 relative -- distances from 1e-9 (series, one term)
    through the switch at x = 2 to the far tail, and exact zeros """
    x = np.concatenate([[0.], rng.uniform(-3, 3, 150), 10. ** rng.uniform(-9, 1.5, 150)])
    y = np.concatenate([[0., 1e-300], rng.uniform(-3, 3, 100)])
    for scale in (1., 0.37):
        G = lgp.Matern(nu=nu, scale=scale).gram(x, y).to_host()
        ref = oracle.Matern(nu=nu, scale=scale)(x[:, None], y[None, :])
        # the oracle's product 2^(1-nu)/Gamma(nu) x^nu K_nu(x) overflows -- or passes through subnormal numbers and keeps a few
        # digits -- at small x for the larger orders: there, the leading series sum_k (-x^2/4)^k / (k! (nu-1) ... (nu-k))
        # (the x^(2 nu) series is far below the last bit)
        lost = ~np.isfinite(ref)
        xx = np.sqrt(2 * nu) * np.abs(x[:, None] - y[None, :]) / scale
        if nu > 30:
            lost |= xx < 0.3
            q = xx ** 2 / 4
            series = 1 - q / (nu - 1) + q ** 2 / (2 * (nu - 1) * (nu - 2)) - q ** 3 / (6 * (nu - 1) * (nu - 2) * (nu - 3))
            assert xx[lost].max() < 0.3
        else:
            assert not lost.any()
            series = ref
        np.testing.assert_allclose(G, np.where(lost, series, ref), rtol=1e-12, atol=1e-200)
    assert G[0, 0] == 1. and np.all(np.isfinite(G))
    # a distance at which K_nu itself overflows: the limit (the oracle's product overflow * underflow is not a number there)
    tiny = lgp.Matern(nu=nu).gram(np.array([0.]), np.array([1e-300, 1e-160])).to_host()
    if nu >= 2:
        np.testing.assert_allclose(tiny, 1., rtol=0, atol=1e-15)


def test_matern_real_order_forms(lgp, rng):
    """ the other entry points of the same core: several fields (compile-time and run-time field counts), the one-triangle grid
    with its mirrored stores against the full grid, the pairs kernel (diagonal), fp32, nu = 0 (``_matern.py:74``: white in the
    distance), output operations, and the GP on top """
    for d in (1, 2, 3, 5):
        x = rng.standard_normal((700, d)); y = rng.standard_normal((90, d))
        k = lgp.Matern(nu=1.7, scale=1.9) * 2.5 + 0.25
        r2 = ((x[:, None, :] - y[None, :, :]) ** 2).sum(-1)
        ref = 2.5 * _matern_core(1.7, r2 / 1.9 ** 2) + 0.25
        xs_, ys_ = _fields(x), _fields(y)
        np.testing.assert_allclose(k.gram(xs_, ys_).to_host(), ref, rtol=1e-12)
        S = k.gram(xs_, xs_).to_host()                              # n >= 512: lower tiles + mirrored stores
        np.testing.assert_array_equal(S, S.T)
        rs = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)
        np.testing.assert_allclose(S, 2.5 * _matern_core(1.7, rs / 1.9 ** 2) + 0.25, rtol=1e-12)
        np.testing.assert_allclose(k.pairs(xs_[:90], ys_).to_host(), np.diag(ref[:90]), rtol=1e-12)
    x = rng.uniform(-3, 3, 300).astype(np.float32)
    G32 = lgp.Matern(nu=0.8).gram(x, x[:40])
    assert G32.dtype == np.float32
    np.testing.assert_allclose(G32.to_host(), oracle.Matern(nu=0.8)(x[:, None].astype(float), x[None, :40].astype(float)), rtol=0, atol=3e-6)
    z = np.array([0., 1., 1., 2.5])
    np.testing.assert_array_equal(lgp.Matern(nu=0).gram(z, z).to_host(), (z[:, None] == z[None, :]).astype(float))
    with pytest.raises(NotImplementedError):
        lgp.Matern(nu=150.3)
    with pytest.raises(AssertionError):
        lgp.Matern(nu=-0.5)
    xs = np.linspace(0, 4, 60); xt = np.linspace(0.1, 3.9, 7); yv = np.sin(xs)
    post = {}
    for mod in (lgp, oracle):
        gp = mod.GP(mod.Matern(nu=2.2, scale=0.8), checkpos=False, checksym=False).addx(xs, 'a').addx(xt, 'b')
        post[mod] = gp.predfromdata({'a': yv}, 'b', raw=True), gp.marginal_likelihood({'a': yv})
    np.testing.assert_allclose(post[lgp][0][0], post[oracle][0][0], rtol=0, atol=1e-8)
    np.testing.assert_allclose(post[lgp][0][1], post[oracle][0][1], rtol=0, atol=1e-8)
    np.testing.assert_allclose(post[lgp][1], post[oracle][1], rtol=1e-9)


@pytest.mark.parametrize('nu', [0.3, 1.0, 1.2, 2.7, 6.4])
def test_matern_real_order_scale_tangent(lgp, rng, nu):
    """ dK/dscale from the same pass (-c (x/2)^nu K_{nu-1}(x) dx/dscale, K_{nu-1} a by-product of the recurrence) against central
    differences of the oracle, and through `Dual` hyperparameters """
    x = rng.uniform(-3, 3, (130, 2)); y = rng.uniform(-3, 3, (70, 2))
    s, h = 1.3, 1e-5
    r2 = ((x[:, None, :] - y[None, :, :]) ** 2).sum(-1)
    x, y = _fields(x), _fields(y)
    K, dK = lgp.Matern(nu=nu, scale=s).gram_dscale(x, y)
    core = lambda r2: _matern_core(nu, r2)
    fd = (core(r2 / (s + h) ** 2) - core(r2 / (s - h) ** 2)) / (2 * h)
    np.testing.assert_allclose(K.to_host(), core(r2 / s ** 2), rtol=1e-12)
    np.testing.assert_allclose(dK.to_host(), fd, rtol=0, atol=2e-9)
    k = lgp.Matern(nu=nu, scale=lgp.Dual(s, {'s': 1.})) * lgp.Dual(2., {'a': 1.})
    G, tan = k.gram_tangents(x, y)
    np.testing.assert_allclose(tan['s'].to_host(), 2 * fd, rtol=0, atol=4e-9)
    np.testing.assert_allclose(tan['a'].to_host(), core(r2 / s ** 2), rtol=1e-12)


def test_discrete_time_kernels_against_oracle(lgp, rng):
    """ MA and AR (``lgp_gram_lagtable``: a device look-up over the integer lag): full grid, the one-triangle grid (n >= 512), the
    pairs form, output operations, large absolute times, and a GP with them """
    t = rng.integers(0, 400, 700)
    s = rng.integers(-50, 500, 90)
    cases = [(lgp.MA(w=[1, 2, 3, 4, 5]), oracle.MA(w=[1, 2, 3, 4, 5])),
             (lgp.MA(w=rng.standard_normal(30), norm=True), None),
             (lgp.AR(phi=[0.5, -0.3, 0.1], maxlag=600), oracle.AR(phi=[0.5, -0.3, 0.1], maxlag=600)),
             (lgp.AR(gamma=[1, 0.5, 0.25], maxlag=300, norm=True), oracle.AR(gamma=[1, 0.5, 0.25], maxlag=300, norm=True)),
             (lgp.AR(slnr=[1 / 10, 1 / 10, -1 / 2], lnc=[1 / 10 + 1j, 1 / 2 + 2j]), oracle.AR(slnr=[1 / 10, 1 / 10, -1 / 2], lnc=[1 / 10 + 1j, 1 / 2 + 2j]))]
    for kd, ko in cases:
        if ko is None:
            continue
        ref = ko(t[:, None], s[None, :])
        scale = np.nanmax(np.abs(ref))
        got = kd.gram(t, s).to_host()
        np.testing.assert_array_equal(np.isnan(got), np.isnan(ref))
        np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(ref), rtol=0, atol=1e-12 * scale)
        S = kd.gram(t, t).to_host()
        np.testing.assert_array_equal(S, S.T)
        np.testing.assert_allclose(S, ko(t[:, None], t[None, :]), rtol=0, atol=1e-12 * scale)
        np.testing.assert_allclose((kd * 3. + 0.5).pairs(t[:90], s).to_host(), 3. * np.diag(ref[:90]) + 0.5, rtol=0, atol=1e-12 * scale, equal_nan=True)
        big = 2 ** 40
        np.testing.assert_array_equal(kd.gram(t + big, s + big).to_host(), got)
    x = np.arange(200)
    y = np.cumsum(rng.standard_normal(200)) * 0.1
    post = {}
    for mod in (lgp, oracle):
        gp = mod.GP(mod.AR(phi=[0.9], maxlag=300) * 0.1 + mod.MA(w=[1., 0.4]) * 0.01, checkpos=False, checksym=False).addx(x[:150], 'a').addx(x[150:], 'b')
        post[mod] = gp.predfromdata({'a': y[:150]}, 'b', raw=True), gp.marginal_likelihood({'a': y[:150]})
    np.testing.assert_allclose(post[lgp][0][0], post[oracle][0][0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(post[lgp][0][1], post[oracle][0][1], rtol=0, atol=1e-9)
    np.testing.assert_allclose(post[lgp][1], post[oracle][1], rtol=1e-10)
    with pytest.raises(TypeError):
        lgp.MA(w=[1, 1], scale=2.)
    with pytest.raises(TypeError):
        lgp.MA(w=[1, 1]).gram(np.linspace(0, 1, 5), np.linspace(0, 1, 5))


def test_bag_of_words_against_oracle(lgp, rng):
    """ ``BagOfWords``: counts on the host, the product of the count matrices on the device """
    vocab = np.array(['12', '7', '3', '100', '44', '#', '$', '+', 'lorem', 'ipsum', 'a1b2'])
    x = np.array([' '.join(rng.choice(vocab, 12)) for _ in range(60)])
    y = np.array([', '.join(rng.choice(vocab, 5)) for _ in range(25)])
    k = lgp.BagOfWords() * 0.5
    np.testing.assert_allclose(k.gram(x, y).to_host(), 0.5 * oracle.BagOfWords()(x[:, None], y[None, :]), rtol=1e-15)
    np.testing.assert_allclose(k(x[:25], y), 0.5 * oracle.BagOfWords()(x[:25], y), rtol=1e-15)
    gp = lgp.GP(lgp.BagOfWords() * 2., checkpos=False).addx(x, 'x').addx(y, 'y')
    np.testing.assert_allclose(gp.prior('x', raw=True), 2 * oracle.BagOfWords()(x[:, None], x[None, :]), rtol=1e-14)
    np.testing.assert_allclose(gp.prior(raw=True)['x', 'y'], 2 * oracle.BagOfWords()(x[:, None], y[None, :]), rtol=1e-14)


def _expn_imag_re_quad(n, x):
    """ Re E_n(-ix) = int_1^inf cos(x w) / w^n dw by QUADPACK's Fourier integrator (an independent route, good to ~1e-10) """
    from scipy import integrate
    return integrate.quad(lambda w: w ** -float(n), 1, np.inf, weight='cos', wvar=float(x), limlst=500, limit=500)[0]


@pytest.mark.parametrize('n', [2, 3, 4, 5, 6, 20, 51])
def test_color_against_closed_forms_and_quadrature(lgp, rng, n):
    """ ``Color(n)`` = (n - 1) Re E_n(-i D): against E_2, E_3 from scipy's complex E_1 by the recurrence n E_{n+1} = e^-z - z E_n
    (1e-14), against the oracle's restatement of the reference's two-branch algorithm where that is accurate (well below its knee,
    or far above it), and everywhere against numerical quadrature of the defining integral at 3e-10 """
    from scipy import special
    x0 = np.zeros(1)
    d = np.concatenate([[0., 1e-12, 1e-6], np.linspace(0.01, 4, 60), [1.99, 2.0, 2.01], np.geomspace(4, 3000, 40)])
    got = lgp.Color(n=n).gram(x0, d).to_host()[0]
    assert got[0] == 1. and np.all(np.abs(got) <= 1 + 1e-14)
    if n in (2, 3):
        z = -1j * d[1:]
        e = np.exp(-z) - z * special.exp1(z)
        if n == 3:
            e = (np.exp(-z) - z * e) / 2
        # (the upward recurrence cancels |z| e^-z against z E_n: its own error grows like eps |z|)
        assert np.all(np.abs(got[1:] - (n - 1) * e.real) <= 2e-14 + 3e-16 * d[1:])
    ref = oracle.Color(n=n)(x0[:, None], d[None, :])[0]
    knee = (special.gamma(n + 20) / np.finfo(float).eps) ** (1 / (n + 19))
    trust = (d < min(3., knee / 4)) | (d > 4 * knee)                    # (the reference's own error estimate, _expint.py:37-51)
    np.testing.assert_allclose(got[trust], ref[trust], rtol=0, atol=1e-11)
    if n <= 20:
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9 if n <= 6 else 3e-2)   # (n = 20: the reference is 1.4e-2 off just below its knee; n = 51: 2e2)
    pick = np.arange(len(d))[3::4]
    pick = pick[d[pick] >= 0.1]                                         # (QUADPACK's Fourier integrator gives up at slow oscillations)
    quad = np.array([(n - 1) * _expn_imag_re_quad(n, v) for v in d[pick]])
    np.testing.assert_allclose(got[pick], quad, rtol=0, atol=3e-10 * n)
    # a GP on top, scale and output operations
    x = rng.uniform(-5, 5, 600)
    G = (lgp.Color(n=n, scale=0.7) * 2.).gram(x, x).to_host()
    np.testing.assert_array_equal(G, G.T)
    if n <= 20:
        np.testing.assert_allclose(G, 2 * oracle.Color(n=n, scale=0.7)(x[:, None], x[None, :]), rtol=0, atol=2e-9 if n <= 6 else 6e-2)


@pytest.mark.parametrize('dw', [1e-9, 0.01, 1., 30., 1e4])
def test_pink_against_scipy_sici(lgp, rng, dw):
    """ ``Pink(dw)``: (Ci(D (1 + dw)) - Ci(D)) / log1p(dw) with scipy's Ci; the cosine branch for D dw < sqrt(eps) """
    x0 = np.zeros(1)
    d = np.concatenate([[0., 1e-12, 1e-7], np.linspace(0.01, 4, 60), np.geomspace(4, 3000, 40)])
    got = lgp.Pink(dw=dw).gram(x0, d).to_host()[0]
    ref = oracle.Pink(dw=dw)(x0[:, None], d[None, :])[0]
    assert got[0] == 1.
    # (Ci at large arguments: the difference of two values each good to 1e-16 absolute, divided by log1p(dw))
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-14 / min(1., np.log1p(dw)) + 1e-14)
    x = rng.uniform(-5, 5, 600)
    G = (lgp.Pink(dw=dw, scale=1.3) * 0.5 + 1.).gram(x, x).to_host()
    np.testing.assert_array_equal(G, G.T)
    np.testing.assert_allclose(G, 0.5 * oracle.Pink(dw=dw, scale=1.3)(x[:, None], x[None, :]) + 1., rtol=0, atol=1e-14 / min(1., np.log1p(dw)) + 1e-14)
    np.testing.assert_allclose((lgp.Pink(dw=dw) * 3.).pairs(x[:50], x[50:100]).to_host(), 3 * oracle.Pink(dw=dw)(x[:50], x[50:100]), rtol=0,
                               atol=3e-14 / min(1., np.log1p(dw)) + 3e-14)


@pytest.mark.parametrize('nu', [0, 0.01, 0.5, 0.99, 1, 2.5, 3.99, 4.5, 7.3, 12., 30.])
def test_bessel_kernel_against_scipy_jv(lgp, rng, nu):
    """ ``Bessel(nu)`` = Gamma(nu + 1) (x/2)^-nu J_nu(x), x = (2 + nu/2) r: the device's J_nu (series, Miller's recurrence, Hankel's
    expansion) against the oracle's scipy ``jv`` over all three regions and their boundaries, 5e-14 absolute """
    s = 2 + nu / 2
    xs = np.concatenate([[0., 1e-200, 1e-9], np.linspace(0.01, 3, 20), [3., 3.0001, np.sqrt(8 * (nu + 1)), np.sqrt(8 * (nu + 1)) + 1e-3],
                         np.linspace(3.01, 40, 120), [max(30., nu * nu), max(30., nu * nu) + 1e-3], np.geomspace(40, 3000, 50)])
    r = xs / s
    got = lgp.Bessel(nu=nu).gram(np.zeros(1), r).to_host()[0]
    ref = oracle.Bessel(nu=nu)(np.zeros((1, 1)), r[None, :])[0]
    assert got[0] == 1. and np.all(np.abs(got) <= 1 + 1e-13)
    from scipy import special
    lost = (xs > 0) & ((xs / 2) ** nu / special.gamma(nu + 1) < 1e-290)      # J_nu itself underflows there in the oracle's product: the limit, 1
    np.testing.assert_allclose(got, np.where(lost, 1., ref), rtol=0, atol=5e-14)
    # several fields up to maxdim, one more refused; one-triangle grid; pairs; output operations; scale
    md = 2 * (int(np.floor(nu)) + 1)
    d = min(md, 3)
    pts = np.empty(600, [('v', float, (d,))])
    pts['v'] = rng.uniform(-3, 3, (600, d))
    k = lgp.Bessel(nu=nu, scale=1.7) * 2. + 0.5
    ko = oracle.Bessel(nu=nu, scale=1.7)
    S = k.gram(pts, pts).to_host()
    np.testing.assert_array_equal(S, S.T)
    np.testing.assert_allclose(S, 2 * ko(pts[:, None], pts[None, :]) + 0.5, rtol=0, atol=2e-13)
    np.testing.assert_allclose(k.pairs(pts[:100], pts[100:200]).to_host(), 2 * ko(pts[:100], pts[100:200]) + 0.5, rtol=0, atol=2e-13)
    if md < 8:
        over = np.empty(10, [('v', float, (md + 1,))])
        over['v'] = rng.uniform(-3, 3, (10, md + 1))
        with pytest.raises((ValueError, lgp.LibraryError)):
            lgp.Bessel(nu=nu).gram(over, over)


def _polylog_re_quad(s, theta):
    """ Re Li_s(e^{i theta}) = 1/Gamma(s) int_0^inf t^(s-1) Re[z / (e^t - z)] dt, z = e^{i theta}: the Bose-Einstein integral by QUADPACK """
    from scipy import integrate, special
    c = np.cos(theta)
    f = lambda t: t ** (s - 1) * (np.exp(t) * c - 1) / (np.exp(2 * t) - 2 * np.exp(t) * c + 1)
    pts = sorted({min(theta, 0.999), 0.5 * min(theta, 0.999), min(2 * theta, 0.9995)})
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        v = (integrate.quad(f, 0, 1., limit=800, points=pts, epsabs=1e-14, epsrel=1e-13)[0] + integrate.quad(f, 1., 40., limit=400, epsabs=1e-14, epsrel=1e-13)[0]
             + integrate.quad(f, 40., 300., limit=400, epsabs=1e-14, epsrel=1e-13)[0])
    return v / special.gamma(s)


@pytest.mark.parametrize('nu', [0, 0.1, 0.3, 0.5, 0.99, 0.9995, 1, 1.0002, 1.01, 1.5, 2, 2.5, 3.7, 4.4, 4.5, 4.9, 7, 1000])
def test_zeta_kernel(lgp, rng, nu):
    """ ``Zeta(nu)``: the device's sum of the host-computed expansion against (i) the oracle's restatement of the reference's Hurwitz
    series form (and its closed-form limit at integer nu), (ii) the Bernoulli polynomials at half-integer nu, (iii) the
    Bose-Einstein integral of the polylogarithm by quadrature, (iv) the Fourier series itself where it converges fast """
    from scipy import special
    s = 1 + 2 * nu
    d = np.concatenate([[0., 1., -3., 1e-9, 1e-6], np.linspace(0.001, 0.5, 40), rng.uniform(-4, 4, 80)])
    got = lgp.Zeta(nu=nu).gram(np.zeros(1), d).to_host()[0]
    # (the reference reduces the SIGNED difference modulo 1, which costs a tiny negative difference its digits: the oracle gets |d|)
    ref = oracle.Zeta(nu=nu)(np.abs(d)[None, :], np.zeros((1, 1)))[0]
    frac = abs(s - (2 * round((s - 1) / 2) + 1))                       # distance of s from an odd integer: the oracle's form loses eps / distance there
    tol = 2e-14 if frac == 0 or frac > 0.5 else 2e-14 + 1e-13 / frac
    np.testing.assert_allclose(got, ref, rtol=0, atol=tol)
    assert np.all(np.abs(got[:3] - 1.) <= 1e-14) and np.all(np.abs(got) <= 1 + 1e-14)
    if nu == 0.5:
        x = np.abs(d) % 1
        np.testing.assert_allclose(got, 6 * (x * x - x + 1 / 6), rtol=0, atol=2e-14)                     # B_2
    if nu == 1.5:
        x = np.abs(d) % 1
        np.testing.assert_allclose(got, -30 * (x ** 4 - 2 * x ** 3 + x * x - 1 / 30), rtol=0, atol=2e-14)   # B_4
    if 0 < nu < 10:
        pick = [5, 8, 15, 25, 44]
        x = np.abs(d[pick]) % 1
        quad = np.array([_polylog_re_quad(s, 2 * np.pi * min(v, 1 - v)) for v in x]) / special.zeta(s, 1)
        np.testing.assert_allclose(got[pick], quad, rtol=0, atol=2e-11)
    if nu >= 1:
        k = np.arange(1, 400001.)
        with np.errstate(over='ignore'):
            direct = np.array([np.sum(np.cos(2 * np.pi * k * v) / k ** s) for v in d[::7]]) / special.zeta(s, 1)
        np.testing.assert_allclose(got[::7], direct, rtol=0, atol=5e-12 if nu < 1.5 else 1e-13)       # (the truncated sum's own tail: N^-(s-1) / (s-1))
    # the other entry points
    x = rng.uniform(-3, 3, 600)
    k = lgp.Zeta(nu=nu, scale=1.5, loc=0.2) * 2. + 0.25
    ko = oracle.Zeta(nu=nu, scale=1.5, loc=0.2)
    S = k.gram(x, x).to_host()
    np.testing.assert_array_equal(S, S.T)
    np.testing.assert_allclose(S, 2 * ko(x[:, None], x[None, :]) + 0.25, rtol=0, atol=4 * tol + 1e-13)
    np.testing.assert_allclose(k.pairs(x[:100], x[100:200]).to_host(), 2 * ko(x[:100], x[100:200]) + 0.25, rtol=0, atol=4 * tol + 1e-13)
    if nu in (0.5, 1):
        xs = np.linspace(0, 2, 80); y = np.sin(2 * np.pi * xs)
        post = {}
        for mod in (lgp, oracle):
            gp = mod.GP(mod.Zeta(nu=nu) + mod.White() * 1e-4, checkpos=False, checksym=False).addx(xs, 'a').addx(xs[:7] + 0.013, 'b')
            post[mod] = gp.predfromdata({'a': y}, 'b', raw=True), gp.marginal_likelihood({'a': y})
        np.testing.assert_allclose(post[lgp][0][0], post[oracle][0][0], rtol=0, atol=1e-8)
        np.testing.assert_allclose(post[lgp][0][1], post[oracle][0][1], rtol=0, atol=1e-8)
        np.testing.assert_allclose(post[lgp][1], post[oracle][1], rtol=1e-8)


@pytest.mark.parametrize('make', [lambda m, s: m.Color(n=2, scale=s), lambda m, s: m.Color(n=3, scale=s), lambda m, s: m.Color(n=6, scale=s),
                                  lambda m, s: m.Pink(dw=1., scale=s), lambda m, s: m.Pink(dw=1e-10, scale=s), lambda m, s: m.Pink(dw=40., scale=s),
                                  lambda m, s: m.Bessel(nu=0, scale=s), lambda m, s: m.Bessel(nu=1.3, scale=s), lambda m, s: m.Bessel(nu=4.5, scale=s)],
                         ids=['Color2', 'Color3', 'Color6', 'Pink1', 'Pink1e-10', 'Pink40', 'Bessel0', 'Bessel1.3', 'Bessel4.5'])
def test_scale_tangents_of_the_special_function_kernels(lgp, rng, make):
    """ dK/dscale of Color (-(n-1) Im E_{n-1}(-ix)), Pink (Ci' = cos x / x) and Bessel (k_nu' = -x / (2 (nu+1)) k_{nu+1}) in the same
    pass as the matrix, against central differences of the oracle over the series / continued-fraction / recurrence regions, and
    through `Dual` hyperparameters """
    s, h = 0.8, 1e-6
    x = np.concatenate([rng.uniform(-3, 3, 100), [0., 1e-9], 10. ** rng.uniform(-3, 1.5, 60)])      # (beyond, the differences' own error grows like h^2 x^3)
    y = np.concatenate([[0.], rng.uniform(-3, 3, 40)])
    K, dK = make(lgp, s).gram_dscale(x, y)
    ref = make(oracle, s)(x[:, None], y[None, :])
    # (differences of the device's own values, good to the last bits; the oracle's Color is 1e-9 off near its knee, which a
    #  difference quotient magnifies; what is left is the quotient's truncation error, h^2 / 6 times a third derivative ~ x^3)
    fd = (make(lgp, s + h).gram(x, y).to_host() - make(lgp, s - h).gram(x, y).to_host()) / (2 * h)
    np.testing.assert_allclose(K.to_host(), ref, rtol=0, atol=1e-9)
    np.testing.assert_allclose(dK.to_host(), fd, rtol=0, atol=1e-6 * max(1., np.abs(fd).max()))
    G, tan = (make(lgp, lgp.Dual(s, {'s': 1.})) * lgp.Dual(2., {'a': 1.})).gram_tangents(x, y)
    np.testing.assert_allclose(tan['s'].to_host(), 2 * fd, rtol=0, atol=2e-6 * max(1., np.abs(fd).max()))
    np.testing.assert_allclose(tan['a'].to_host(), ref, rtol=0, atol=1e-9)


class TestReferenceSpecialValues:
    """ the device's Zeta and Color cores against golden values THE REFERENCE HOLDS (tests/golden/reference_special/: mpmath values
    cached by its own tests, ``tests/test_special.py:190-226,415-424``); CPU counterpart: tests/test_reference_special.py """

    def test_zeta_kernel_against_the_reference_values(self, lgp):
        from . import reference_special as R
        eps = np.finfo(float).eps
        worst = 0.
        for cid, x, s, z in R.periodic_zeta_cases():
            for si, zi in zip(s, z):
                want = zi.real
                zs = want[0]                                   # x[0] = -1: F(integer, s) = zeta(s)
                nu = (float(si) - 1) / 2
                got = lgp.Zeta(nu=nu).gram(np.zeros(1), x).to_host()[0] * zs
                err = np.max(np.abs(got - want)) / (eps * np.max(np.abs(want)))
                worst = max(worst, err)
                assert err < 150, (cid, si, err)              # the reference's own tolerance (``tests/test_special.py:222-226``)
        assert worst < 60                                      # (measured on the CPU mirror of the same sum: 35)

    def test_color_kernel_against_the_reference_values(self, lgp):
        from . import reference_special as R
        n, x, sol = R.expn_case()
        for k, row in zip(n, sol):
            got = lgp.Color(n=int(k)).gram(np.zeros(1), x).to_host()[0]
            want = (k - 1) * row.real                          # Color(n)(d) = (n - 1) Re E_n(-i d)
            # the reference holds its own evaluation to 1e-7 of the largest entry; the device's series / continued fraction to 1e-12
            assert np.max(np.abs(got - want)) <= 1e-12 * max(np.max(np.abs(want)), 1.), (k, np.max(np.abs(got - want)))

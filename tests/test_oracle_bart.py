"""
Pins the CPU oracle's BART restatement (oracle/kernels.py) with the reference's
own property / known-answer tests, ``tests/kernels/test_bart.py`` (ported test
by test; parameter grids thinned where the pure-Python deep recursion would
take minutes).  No GPU involved.
"""

import numpy as np
import pytest
from pytest import mark

import oracle
from oracle.kernels import BART
from . import util

rng = np.random.default_rng(202307302223)

plist = [1, 5]

smark = mark.parametrize('sb,sbw,sa,w', sum([[
    (*rng.integers(0, 4, (3, p)), rng.integers(1, 10, p)),
    (*np.zeros((3, p), int), rng.integers(1, 10, p)),
    (np.zeros(p, int), np.pad([1], (0, p - 1)), np.zeros(p, int), rng.integers(1, 10, p)),
    (rng.integers(0, 10, p), (np.arange(p) == rng.integers(p)).astype(int), rng.integers(0, 10, p), rng.integers(1, 10, p)),
] for p in plist], []))

amark = mark.parametrize('a', [
    pytest.param(np.array([0, 1])[:, None], id='a01'),
    pytest.param(np.linspace(0.01, 0.99, 3)[:, None], id='aother'),
])

bmark = mark.parametrize('b', [
    pytest.param(np.array([0, np.inf]), id='b0inf'),
    pytest.param(np.linspace(1, 10, 3), id='bother'),
])

umark = mark.parametrize('u', [np.array([0, 1])[:, None, None]])

mdmark = mark.parametrize('md', range(4))


@mdmark
@bmark
@amark
@smark
def test_lower_lt_upper(sb, sbw, sa, w, a, b, md):
    """ 0 <= lower <= interp/stricter upper <= upper <= 1 (ref :48-63) """
    lw = BART.correlation(sb, sbw, sa, alpha=a, beta=b, gamma=0, maxd=md, weights=w)
    au = BART.correlation(sb, sbw, sa, alpha=a, beta=b, gamma='auto' if 1 <= md <= 3 else 0, maxd=md, weights=w)
    up = BART.correlation(sb, sbw, sa, alpha=a, beta=b, gamma=1, maxd=md, weights=w)
    np.testing.assert_array_max_ulp(np.zeros_like(lw), np.minimum(0, lw))
    np.testing.assert_array_max_ulp(lw, np.minimum(lw, au))
    np.testing.assert_array_max_ulp(au, np.minimum(au, up))
    np.testing.assert_array_max_ulp(up, np.minimum(up, 1))
    if md <= 2:
        vg = BART.correlation(sb, sbw, sa, alpha=a, beta=b, gamma=1, maxd=md * 2, reset=[md], weights=w)
        np.testing.assert_array_max_ulp(lw, np.minimum(lw, vg))
        np.testing.assert_array_max_ulp(vg, np.minimum(vg, up))


@bmark
@amark
@smark
def test_lower_upper_incr_maxd(sb, sbw, sa, w, a, b):
    """ lower/upper increases/decreases as maxd is increased (ref :65-80) """
    for md in range(4):
        lw = BART.correlation(sb, sbw, sa, alpha=a, beta=b, gamma=0, maxd=md, weights=w)
        up = BART.correlation(sb, sbw, sa, alpha=a, beta=b, gamma=1, maxd=md, weights=w)
        if md:
            np.testing.assert_array_max_ulp(plw, np.minimum(lw, plw), 2)
            np.testing.assert_array_max_ulp(pup, np.maximum(up, pup), 2)
        plw = lw
        pup = up


@mdmark
@umark
@bmark
@amark
@smark
def test_incr_beta(sb, sbw, sa, w, a, b, u, md):
    """ increases as beta is increased (ref :82-91) """
    c = BART.correlation(sb, sbw, sa, alpha=a, beta=b, gamma=u, maxd=md, weights=w)
    ci = BART.correlation(sb, sbw, sa, alpha=a, beta=b + 1e-3, gamma=u, maxd=md, weights=w)
    np.testing.assert_array_max_ulp(ci, np.maximum(c, ci))


@mdmark
@umark
@bmark
@amark
@smark
def test_incr_alpha(sb, sbw, sa, w, a, b, u, md):
    """ decreases as alpha is increased (ref :93-103) """
    da = 1e-3
    a = np.minimum(1 - da, a)
    c = BART.correlation(sb, sbw, sa, alpha=a, beta=b, gamma=u, maxd=md, weights=w)
    ci = BART.correlation(sb, sbw, sa, alpha=a + da, beta=b, gamma=u, maxd=md, weights=w)
    np.testing.assert_array_max_ulp(ci, np.minimum(c, ci))


@mdmark
@umark
@bmark
@amark
@mark.parametrize('sb,sbw,sa,w', sum([[
    # n^0 = 0
    (rng.integers(0, 10, p), np.zeros(p, int), rng.integers(0, 10, p), rng.integers(1, 10, p)),
    (*np.zeros((3, p), int), rng.integers(1, 10, p)),
] for p in plist], []) + sum([[
    # w = 0
    (*rng.integers(0, 4, (3, p)), np.zeros(p)),
] for p in plist], []) + sum([[
    # wi = 0 or ni = 0
    np.concatenate([rng.integers(0, 4, (3, p)), rng.integers(1, 10, (1, p))]) *
    (rng.integers(0, 2, p).astype(bool) ^ np.array([0, 0, 0, 1], bool)[:, None]),
] for p in plist], []) + [
    # p = 0
    (*np.empty((3, 0), int), np.empty(0)),
])
def test_corr_1(sb, sbw, sa, w, a, b, u, md):
    """ correlation = 1 if n^0 = 0, w = 0, (wi = 0 or ni = 0), p = 0 (ref :105-133) """
    c = BART.correlation(sb, sbw, sa, alpha=a, beta=b, gamma=u, maxd=md, weights=w)
    np.testing.assert_array_max_ulp(c, np.ones_like(c))


@mdmark
@umark
@bmark
@amark
@smark
def test_swap_ab(sb, sbw, sa, w, a, b, u, md, rng):
    """ invariant under swapping of nplus and nminus (ref :135-147) """
    swap = rng.integers(0, 2, sb.size)
    s1 = np.where(swap, sa, sb)
    s2 = np.where(swap, sb, sa)
    c = BART.correlation(sb, sbw, sa, alpha=a, beta=b, gamma=u, maxd=md, weights=w)
    cs = BART.correlation(s1, sbw, s2, alpha=a, beta=b, gamma=u, maxd=md, weights=w)
    np.testing.assert_array_max_ulp(c, cs, 16)


@mdmark
@umark
@bmark
@amark
@smark
def test_perm_dims(sb, sbw, sa, w, a, b, u, md, rng):
    """ invariant under reordering of the dimensions (ref :149-159) """
    perm = rng.permutation(sb.size)
    c = BART.correlation(sb, sbw, sa, alpha=a, beta=b, gamma=u, maxd=md, weights=w)
    cp = BART.correlation(sb[perm], sbw[perm], sa[perm], alpha=a, beta=b, gamma=u, maxd=md, weights=w[perm])
    np.testing.assert_array_max_ulp(c, cp, 64)


@mark.parametrize('md', range(3))
@umark
@bmark
@amark
@mark.parametrize('sb,sbw,sa,w', sum([[
    (*rng.integers(0, 10, (3, p)), rng.integers(1, 10, p)),
] for p in plist], []))
def test_incr_n0(sb, sbw, sa, w, a, b, u, md, rng):
    """ correlation decreases as n0 increases at fixed ntot (ref :161-189) """
    sb, sbw, sa = sb.copy(), sbw.copy(), sa.copy()
    c = BART.correlation(sb, sbw, sa, alpha=a, beta=b, gamma=u, maxd=md, weights=w)
    ntot = sb + sbw + sa
    which = rng.permuted(np.eye(sb.size)[0]).astype(bool)
    dn = np.where(which & (sb + sa).astype(bool), 1, 0)
    lr = rng.integers(0, 2, sb.size).astype(bool)
    lr ^= lr & ~sb.astype(bool) | ~lr & ~sa.astype(bool)
    sb -= np.where(lr, dn, 0)
    sbw += dn
    sa -= np.where(~lr, dn, 0)
    assert np.all(ntot == sb + sbw + sa)
    assert np.all(sb >= 0) and np.all(sa >= 0)
    ci = BART.correlation(sb, sbw, sa, alpha=a, beta=b, gamma=u, maxd=md, weights=w)
    np.testing.assert_array_max_ulp(ci, np.minimum(c, ci))


def _values(m):
    vals = m.args[1]
    return sum((getattr(v, 'values', (v,)) for v in vals), ())


@mark.parametrize('sb,sbw,sa,w,a,b,md', [
    # n^0 = 0
    (rng.integers(0, 10, p), np.zeros(p, int), rng.integers(0, 10, p), rng.integers(1, 10, p), a, b, d)
    for p in plist for a in _values(amark) for b in _values(bmark) for d in range(0, 4)
] + [
    # alpha = 0
    (rng.integers(0, 10, p), rng.integers(1, 10, p), rng.integers(0, 10, p), rng.integers(1, 10, p), 0, b, d)
    for p in plist for b in _values(bmark) for d in range(0, 3)
] + [
    # beta = inf
    (rng.integers(0, 10, p), rng.integers(1, 10, p), rng.integers(0, 10, p), rng.integers(1, 10, p), a, np.inf, d)
    for p in plist for a in _values(amark) for d in range(1, 3)
])
def test_lower_eq_upper(sb, sbw, sa, w, a, b, md):
    """ upper = lower where the solution is exact (ref :195-224) """
    lw = BART.correlation(sb, sbw, sa, alpha=a, beta=b, gamma=0, maxd=md, weights=w)
    up = BART.correlation(sb, sbw, sa, alpha=a, beta=b, gamma=1, maxd=md, weights=w)
    np.testing.assert_array_max_ulp(lw, up)


@mark.parametrize('md', range(3))
@umark
@mark.parametrize('b', [np.array([0., 2.])])
@mark.parametrize('a', [np.array([0.3, 0.95])[:, None]])
@mark.parametrize('sb,sbw,sa,w', [
    (*np.random.default_rng(1).integers(0, 3, (3, 1)), np.array([3])),
    (*np.random.default_rng(2).integers(0, 3, (3, 3)), np.array([1, 4, 2])),
])
def test_no_shortcuts(sb, sbw, sa, w, a, b, u, md):
    """ same result if no recursions are avoided (ref :226-235; small counts: full recursion in Python) """
    c = BART.correlation(sb, sbw, sa, alpha=a, beta=b, gamma=u, maxd=md, weights=w)
    cd = BART.correlation(sb, sbw, sa, alpha=a, beta=b, gamma=u, maxd=md, debug=True, weights=w)
    np.testing.assert_array_max_ulp(c, cd, 32)


@mdmark
@umark
@bmark
@amark
@smark
def test_nzero(sb, sbw, sa, w, a, b, u, md):
    """ adding zeros in sb, sbw, sa does not change the result (ref :237-248) """
    kw = dict(alpha=a, beta=b, gamma=u, maxd=md)
    c = BART.correlation(sb, sbw, sa, weights=w, **kw)
    z = lambda x, f=0, p=4: np.concatenate([x, np.full(p, f)])
    c0 = BART.correlation(z(sb), z(sbw), z(sa), weights=z(w, 1), **kw)
    np.testing.assert_array_max_ulp(c, c0, 4)


@mdmark
@umark
@bmark
@amark
@smark
def test_wzero(sb, sbw, sa, w, a, b, u, md, rng):
    """ adding zero weights with random n does not change the result (ref :250-261) """
    kw = dict(alpha=a, beta=b, gamma=u, maxd=md)
    c = BART.correlation(sb, sbw, sa, weights=w, **kw)
    z = lambda x, f=5, p=4: np.concatenate([x, rng.integers(f + 1, size=p)])
    c0 = BART.correlation(z(sb), z(sbw), z(sa), weights=z(w, 0), **kw)
    np.testing.assert_array_max_ulp(c, c0, 4)


def test_structured():
    """ ref :263-266 """
    X = np.arange(10 * 2.).reshape(1, -1, 2).view('d,d')
    splits = BART.splits_from_coord(X)
    cov = oracle.BART(splits=splits)(X, X.T)
    assert cov.shape == np.broadcast_shapes(X.shape, X.T.shape)
    X1 = X.reshape(-1)
    cov = oracle.BART(splits=splits)(X1[:, None], X1[None, :])
    assert cov.shape == (10, 10)
    util.assert_allclose(np.diag(cov), 1)


def test_duplicates():
    """ ref :268-271 """
    x = np.repeat(np.arange(10 * 2.).reshape(-1, 2), 2, axis=0).view('d,d').squeeze(axis=-1)
    length, splits = BART.splits_from_coord(x)
    assert np.all(length == 9)


def test_integer():
    """ ref :273-281 """
    X = np.arange(10 * 2).reshape(1, -1, 2)
    X1 = X.astype('d').view('d,d').squeeze(-1)
    X2 = X.astype('l').view('l,l').squeeze(-1)
    kernel1 = oracle.BART(splits=BART.splits_from_coord(X1))
    kernel2 = oracle.BART(splits=BART.splits_from_coord(X2))
    v1 = kernel1(X1, X1.T)
    v2 = kernel2(X1, X1.T)
    np.testing.assert_array_equal(v1, v2)


def test_pnt():
    """ ref :283-289 """
    sss = [1, 2], [3, 1], [5, 8]
    alpha = 0.9
    beta = 1.6
    c1 = BART.correlation(*sss, alpha=alpha, beta=beta, maxd=2)
    c2 = BART.correlation(*sss, alpha=None, beta=None, pnt=[alpha, alpha / 2 ** beta, alpha / 3 ** beta])
    util.assert_allclose(c1, c2, rtol=1e-15)


def test_wrong_gamma():
    """ ref :291-293 """
    with pytest.raises(KeyError):
        BART.correlation([0], [0], [0], gamma='ciao')


def test_intercept():
    """ ref :295-301 """
    sss = [1, 2], [3, 1], [5, 8]
    alpha = 0.9
    c1 = BART.correlation(*sss, alpha=alpha)
    c2 = BART.correlation(*sss, alpha=alpha, intercept=False)
    c2 = c2 * alpha + (1 - alpha)
    util.assert_allclose(c1, c2, rtol=1e-15)


def test_splits_1d():
    """ ref :303-311 """
    l = [2]
    s = np.array([-1/2, 1/2])
    x = np.array([-1, 0, 1])[:, None]
    k1 = oracle.BART(splits=(l, s))
    k2 = oracle.BART(splits=(l, s[:, None]))
    np.testing.assert_array_equal(k1(x, x.T), k2(x, x.T))


def test_f32():
    """ ref :313-316 """
    sss = [1, 2], [3, 1], [5, 8]
    c = BART.correlation(*sss, pnt=np.array([0.9, 0.4, 0.3], 'f'))
    assert c.dtype == 'f'


def test_i32():
    """ int32 inputs do not force fp32 arithmetic (ref :318-330) """
    sss = [1, 2], [3, 1], [5, 8]
    sss = [np.array(s, np.int32) for s in sss]
    pnt = np.array([0.9, 0.4, 0.3], np.float64)
    c1 = BART.correlation(*sss, pnt=pnt)
    assert c1.dtype == np.float64
    sss = [np.array(s, np.int64) for s in sss]
    c2 = BART.correlation(*sss, pnt=pnt)
    assert c2.dtype == np.float64
    np.testing.assert_array_max_ulp(c1, c2, 0)


@mark.parametrize('md,reset', [
    (0, None),
    (1, None),
    (2, None), (2, [1]),
    (3, None), (3, [1, 2]),
    (4, [2]), (4, [1, 2, 3]),
])
@umark
@bmark
@amark
@smark
def test_altinput(sb, sbw, sa, w, a, b, u, md, reset, rng):
    """ the two independent implementations of the reference (``_correlation_old`` :516-610 and the
    altinput ``_correlation`` :628-806) agree (ref :332-353) """
    c1 = BART.correlation(sb, sbw, sa, alpha=a, beta=b, gamma=u, maxd=md, reset=reset, weights=w)
    n = sb + sbw + sa
    ix = sb
    iy = sb + sbw
    ix, iy = np.broadcast_arrays(ix, iy)
    swap = rng.integers(0, 2, size=ix.shape, dtype=bool)
    ix, iy = np.where(swap, iy, ix), np.where(swap, ix, iy)
    c2 = BART.correlation(n, ix, iy, alpha=a, beta=b, gamma=u, maxd=md, reset=reset, weights=w, altinput=True)
    util.assert_allclose(c1, c2, rtol=1e-15, atol=2e-15)


def test_index_input(rng):
    """ coordinates or directly indices give the same result (ref :355-367) """
    n, p = 30, 10
    asstruct = lambda x: np.asarray(x).view([('f0', x.dtype, p)]).squeeze(-1)
    X = asstruct(rng.standard_normal((n, p)))
    splits = BART.splits_from_coord(X)
    x, y = asstruct(rng.standard_normal((2, n, p)))
    x, y = np.ix_(x, y)
    ix = asstruct(BART.indices_from_coord(x, splits))
    iy = asstruct(BART.indices_from_coord(y, splits))
    c1 = oracle.BART(splits=splits)(x, y)
    c2 = oracle.BART(splits=splits, indices=True)(ix, iy)
    np.testing.assert_array_max_ulp(c1, c2, 0)


def test_empty():
    """ ref :369-372 """
    empty = np.array([], int)
    c = BART.correlation(empty, empty, empty)
    assert c == 1


def test_docstring_closed_form_depth1():
    """ hand-checkable D=1 closed form of ``_kernels/_bart.py:165-178,697-708``:
    1 - P0 + Q (P0 - P0/W sum_i w_i n0_i/n_i), Q = 1 - P1 + gamma P1 """
    n = np.array([4, 7]); ix = np.array([1, 6]); iy = np.array([3, 2]); w = np.array([2., 3.])
    alpha, beta, gamma = 0.95, 2., 0.3
    P0, P1 = alpha, alpha / 2 ** beta
    Q = 1 - P1 + gamma * P1
    expected = 1 - P0 + Q * (P0 - P0 / w.sum() * np.sum(w * np.abs(ix - iy) / n))
    c = BART.correlation(n, ix, iy, alpha=alpha, beta=beta, gamma=gamma, maxd=1, weights=w, altinput=True)
    util.assert_allclose(c, expected, rtol=1e-15)

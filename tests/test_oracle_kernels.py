"""
Pins the oracle's ExpQuad / Maternp / fasthash restatements with the reference's
own pins: ``tests/kernels/test_kernels.py`` (k(x,x)=1, stationarity, PSD +
symmetry, loc/scale, implicit vs explicit fields, Maternp == Bessel-K form) and
the fasthash known-answer table of ``tests/test_jax.py:112-186``.  CPU only.
"""

import ctypes
import json
import os

import numpy as np
import pytest
from scipy import special

import oracle
from oracle import kernels as ok
from . import util

HERE = os.path.dirname(os.path.abspath(__file__))

KERNELS = {
    'ExpQuad': lambda **kw: oracle.ExpQuad(**kw),
    'Maternp0': lambda **kw: oracle.Maternp(p=0, **kw),
    'Maternp1': lambda **kw: oracle.Maternp(p=1, **kw),
    'Maternp2': lambda **kw: oracle.Maternp(p=2, **kw),
    'Maternp5': lambda **kw: oracle.Maternp(p=5, **kw),
    'GammaExp1': lambda **kw: oracle.GammaExp(**kw),
    'GammaExp1.5': lambda **kw: oracle.GammaExp(gamma=1.5, **kw),
    'GammaExp2': lambda **kw: oracle.GammaExp(gamma=2, **kw),
    'Cauchy': lambda **kw: oracle.Cauchy(**kw),
    'Cauchy1-3': lambda **kw: oracle.Cauchy(alpha=1, beta=3, **kw),
}


class KernelPins:
    """ re-used by the GPU parity tests with the device kernel factories """

    factories = KERNELS

    @pytest.fixture(params=sorted(KERNELS))
    def make(self, request):
        return self.factories[request.param]

    def random_x(self, rng, n=100, nd=False):
        if nd:
            x = np.empty(n, dtype=[('a', float), ('b', float), ('c', float)])
            for f in x.dtype.names:
                x[f] = rng.uniform(-5, 5, n)
            return x
        return rng.uniform(-5, 5, n)

    def test_normalized(self, make, rng):
        """ k(x, x) = 1 (ref :303-308) """
        x = self.random_x(rng)
        util.assert_allclose(make()(x, x), 1, rtol=1e-14, atol=1e-15)

    def test_normalized_nd(self, make, rng):
        x = self.random_x(rng, nd=True)
        util.assert_allclose(make()(x, x), 1, rtol=1e-14, atol=1e-15)

    def test_stationary(self, make, rng):
        """ k(x, y) = k(x - y, 0) (ref :296-301) """
        x = self.random_x(rng, 50); y = self.random_x(rng, 50)
        k = make()
        util.assert_allclose(k(x, y), k(x - y, np.zeros(50)), atol=1e-15, rtol=1e-14)

    def test_positive_symmetric(self, make, rng):
        """ PSD bound min eig >= -N eps max eig, exact symmetry (ref :163-187) """
        for nd in (False, True):
            x = self.random_x(rng, nd=nd)
            cov = make()(x[None, :], x[:, None])
            util.assert_allclose(cov, cov.T, rtol=1e-5, atol=1e-7)
            eigv = np.linalg.eigvalsh(cov)
            assert np.min(eigv) >= -len(cov) * np.finfo(float).eps * np.max(eigv)

    def test_symmetric_offdiagonal(self, make, rng):
        """ k(x, y) = k(y, x) (ref :217-231) """
        x = self.random_x(rng, 60); y = self.random_x(rng, 40)
        k = make()
        util.assert_allclose(k(x[:, None], y[None, :]), k(y[:, None], x[None, :]).T, atol=1e-10)

    def test_implicit_fields(self, make, rng):
        """ unstructured 1-D input == structured input with one field (ref :239-245) """
        x = self.random_x(rng, 30)
        xs = np.empty(30, dtype=[('f0', float)]); xs['f0'] = x
        k = make()
        util.assert_allclose(k(x[:, None], x[None, :]), k(xs[:, None], xs[None, :]), rtol=1e-14)

    def test_loc_scale(self, make, rng):
        """ kernel(loc=l, scale=s)(x, y) == kernel()((x-l)/s, (y-l)/s) (ref :247-260) """
        x = self.random_x(rng, 40); y = self.random_x(rng, 40)
        loc, scale = -2.0, 0.3
        c1 = make()((x - loc) / scale, (y - loc) / scale)
        c2 = make(loc=loc, scale=scale)(x, y)
        util.assert_allclose(c1, c2, rtol=1e-12, atol=1e-15)

    def test_scalar_algops(self, make, rng):
        """ kernel * c and kernel + c (ref ``_Kernel/_alg.py:32-82``) """
        x = self.random_x(rng, 20)
        k = make()
        base = k(x[:, None], x[None, :])
        util.assert_allclose((k * 3.5)(x[:, None], x[None, :]), base * 3.5, rtol=1e-15)
        util.assert_allclose((2 * k + 0.25)(x[:, None], x[None, :]), base * 2 + 0.25, rtol=1e-15)
        util.assert_allclose(((k + 1) * 2)(x[:, None], x[None, :]), (base + 1) * 2, rtol=1e-15)

    def test_broadcast_shapes(self, make, rng):
        """ call contract: result shape = broadcast shape (ref ``_crosskernel.py:192-200``) """
        k = make()
        x = self.random_x(rng, 12)
        assert k(x[:, None], x[None, :]).shape == (12, 12)
        assert k(x[None, :], x[:, None]).shape == (12, 12)
        assert k(x, x).shape == (12,)
        assert k(x[:3, None, None], x[None, :4, None]).shape == (3, 4, 1)
        assert k(x[0], x[1]).shape == ()
        util.assert_allclose(k(x[:, None], x[None, :]).T, k(x[None, :], x[:, None]), rtol=1e-15)


class TestOracleKernelPins(KernelPins):
    pass


@pytest.mark.parametrize('p', range(10))
def test_maternp_vs_bessel(p, rng):
    """ Maternp(p) == 2^(1-nu)/Gamma(nu) x^nu K_nu(x), nu = p + 1/2 (ref :500-512) """
    r = rng.uniform(0.01, 10, 200)
    nu = p + 0.5
    x = np.sqrt(2 * nu) * r
    expected = 2 ** (1 - nu) / special.gamma(nu) * x ** nu * special.kv(nu, x)
    result = oracle.Maternp(p=p)(r, np.zeros_like(r))
    util.assert_allclose(result, expected, rtol=1e-9, atol=1e-16)


def test_expquad_formula(rng):
    x = rng.uniform(-3, 3, 50); y = rng.uniform(-3, 3, 50)
    np.testing.assert_array_equal(oracle.ExpQuad()(x, y), np.exp(-1/2 * (x - y) ** 2))


def test_field_order_accumulation(rng):
    """ r^2 accumulates left to right in field order (``_Kernel/_util.py:74-99``) """
    x = np.empty(40, dtype=[('a', float), ('b', float, 2), ('c', float)])
    y = np.empty(40, dtype=x.dtype)
    for arr in (x, y):
        arr['a'] = rng.standard_normal(40); arr['b'] = rng.standard_normal((40, 2)); arr['c'] = rng.standard_normal(40)
    r2 = (x['a'] - y['a']) ** 2
    r2 = r2 + np.sum((x['b'] - y['b']) ** 2, axis=-1)
    r2 = r2 + (x['c'] - y['c']) ** 2
    np.testing.assert_array_equal(oracle.ExpQuad()(x, y), np.exp(-1/2 * r2))


# ---- fasthash known-answer table ------------------------------------------------

@pytest.fixture(scope='module')
def kat():
    with open(os.path.join(HERE, 'golden', 'fasthash_kat.json')) as f:
        return json.load(f)


def test_fasthash_kat(kat):
    """ the reference's own golden vectors (``tests/test_jax.py:112-186``) """
    for inp, h64, h32 in zip(kat['inputs'], kat['hashes64'], kat['hashes32']):
        buf = np.array(inp, np.uint8)
        assert int(ok.fasthash64(buf, kat['seed64'])) == h64
        assert int(ok.fasthash32(buf, kat['seed32'])) == h32


def test_fasthash_vs_reference_c(kat, rng):
    """ the reference's C fixture compiled in place (oracle/Makefile -> oracle/_ref) against the restatement """
    path = os.path.join(os.path.dirname(HERE), 'oracle', '_ref', 'libfasthash_ref.so')
    if not os.path.exists(path):
        pytest.skip('oracle/_ref/libfasthash_ref.so not built (needs /root/reference; run make -C oracle)')
    lib = ctypes.CDLL(path)
    lib.fasthash64.restype = ctypes.c_uint64
    lib.fasthash64.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64]
    lib.fasthash32.restype = ctypes.c_uint32
    lib.fasthash32.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]
    for inp, h64, h32 in zip(kat['inputs'], kat['hashes64'], kat['hashes32']):
        buf = np.array(inp + [0], np.uint8)
        assert lib.fasthash64(buf.ctypes.data, len(inp), kat['seed64']) == h64
        assert lib.fasthash32(buf.ctypes.data, len(inp), kat['seed32']) == h32
    for _ in range(50):
        n = int(rng.integers(0, 200))
        buf = rng.integers(0, 256, n + 1, dtype=np.uint8)
        seed = int(rng.integers(0, 2 ** 63))
        assert lib.fasthash64(buf.ctypes.data, n, seed) == int(ok.fasthash64(buf[:n], seed))


def test_hash_equality_equals_vector_equality(rng):
    """ BART uses the hash only as an equality test of index vectors (``_bart.py:674-681``): for the
    vectors that occur, hash equality == exact equality, which is what the device kernel computes """
    seed = np.uint64(16132933535611723338)
    ix = rng.integers(0, 4, (200, 6)).astype(np.int64)
    hashes = np.array([int(ok.fasthash64(row, seed)) for row in ix], dtype=np.uint64)
    eq_hash = hashes[:, None] == hashes[None, :]
    eq_vec = np.all(ix[:, None, :] == ix[None, :, :], axis=-1)
    np.testing.assert_array_equal(eq_hash, eq_vec)


def test_gammaexp_cauchy_closed_forms(rng):
    """ the definitions of ``_kernels/_basic.py:94-115,315-344`` and the relations their docstrings state:
    GammaExp(gamma=1) is Matern 1/2, GammaExp(gamma=2) is exp(-r^2), Cauchy(alpha=2) is the rational quadratic """
    x = rng.standard_normal(50)
    y = rng.standard_normal(50)
    r = np.abs(x - y)
    np.testing.assert_allclose(oracle.GammaExp()(x, y), np.exp(-r), rtol=1e-15)
    np.testing.assert_allclose(oracle.GammaExp()(x, y), oracle.Maternp(p=0)(x, y), rtol=1e-14)
    np.testing.assert_array_equal(oracle.GammaExp(gamma=2)(x, y), np.exp(-(x - y) ** 2))
    np.testing.assert_allclose(oracle.GammaExp(gamma=0.7)(x, y), np.exp(-r ** 0.7), rtol=1e-14)
    np.testing.assert_allclose(oracle.Cauchy(alpha=2, beta=5)(x, y), (1 + r ** 2 / 5) ** -2.5, rtol=1e-14)
    np.testing.assert_allclose(oracle.Cauchy(alpha=1.3, beta=0.4)(x, y), (1 + r ** 1.3 / 0.4) ** (-0.4 / 1.3), rtol=1e-14)
    big = oracle.Cauchy(alpha=1.5, beta=1e9)(x, y)          # beta -> inf: GammaExp(gamma=alpha, scale=alpha^(1/alpha))
    np.testing.assert_allclose(big, oracle.GammaExp(gamma=1.5, scale=1.5 ** (1 / 1.5))(x, y), rtol=1e-7)
    for bad in (dict(gamma=0), dict(gamma=2.1)):
        with pytest.raises(AssertionError):
            oracle.GammaExp(**bad)(x, y)
    for bad in (dict(alpha=0), dict(alpha=3), dict(beta=0)):
        with pytest.raises(AssertionError):
            oracle.Cauchy(**bad)(x, y)


def test_simple_stationary_kernels_closed_forms(rng):
    """ Constant, White, Expon, HoleEffect, Periodic, Cos, Sinc of the oracle against the formulas of their docstrings
    (``_kernels/_basic.py:34-59,187-208,254-269,299-310``, ``_spectral.py:25-38,113-122``), with scale and loc """
    x = rng.uniform(-4, 4, 60); y = rng.uniform(-4, 4, 45)
    y[:10] = x[:10]
    d = (x[:, None] - y[None, :])
    eps = np.finfo(float).eps
    np.testing.assert_array_equal(oracle.Constant(scale=2.)(x[:, None], y[None, :]), np.ones((60, 45)))
    np.testing.assert_array_equal(oracle.White()(x[:, None], y[None, :]), (d == 0).astype(int))
    np.testing.assert_allclose(oracle.Expon(scale=2.)(x[:, None], y[None, :]), np.exp(-np.abs(d) / 2), rtol=1e-14)
    np.testing.assert_allclose(oracle.HoleEffect(scale=0.5)(x[:, None], y[None, :]), (1 - 2 * np.abs(d)) * np.exp(-2 * np.abs(d)), rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(oracle.Periodic(outerscale=0.7, scale=1.5)(x[:, None], y[None, :]), np.exp(-2 * (np.sin(d / 3) / 0.7) ** 2), rtol=1e-13)
    np.testing.assert_allclose(oracle.Cos(scale=0.8, loc=3.)(x[:, None], y[None, :]), np.cos(d / 0.8), rtol=0, atol=1e-14)
    u = np.abs(d) / 0.3 + eps
    np.testing.assert_allclose(oracle.Sinc(scale=0.3)(x[:, None], y[None, :]), np.sinc(u), rtol=0, atol=3e-14)      # (x/s - y/s against (x - y)/s, arguments up to 27)
    # the Taylor branch of sinc meets the closed form where they switch
    from oracle.kernels import sinc
    t = np.array([0.0999999, 0.1, 0.1000001, 1e-3, 0.])
    np.testing.assert_allclose(sinc(t), np.sinc(t), rtol=5e-16, atol=0)
    # structured points: White multiplies the equalities of the fields
    a = np.zeros(4, [('u', float), ('v', float)]); a['u'] = [0, 0, 1, 1]; a['v'] = [0, 1, 0, 1]
    np.testing.assert_array_equal(oracle.White()(a[:, None], a[None, :]), np.eye(4, dtype=int))
    with pytest.raises(ValueError):
        oracle.Cos()(a[:, None], a[None, :])


def test_random_walk_kernels_closed_forms(rng):
    """ Linear, Wiener, OrnsteinUhlenbeck, BrownianBridge, Decaying, FracBrownian, WienerIntegral, StationaryFracBrownian of
    the oracle against the formulas of ``_kernels/_basic.py:76-91,362-379`` and ``_randomwalk.py`` and their known limits """
    x = rng.uniform(0, 1, 40); y = rng.uniform(0, 1, 30)
    X, Y = x[:, None], y[None, :]
    np.testing.assert_allclose(oracle.Linear(scale=2.)(X, Y), X * Y / 4, rtol=1e-15)
    np.testing.assert_array_equal(oracle.Wiener()(X, Y), np.minimum(X, Y))
    np.testing.assert_allclose(oracle.BrownianBridge()(X, Y), np.minimum(X, Y) - X * Y, rtol=1e-15)
    np.testing.assert_allclose(oracle.OrnsteinUhlenbeck(scale=0.5)(X, Y), np.exp(-2 * np.abs(X - Y)) - np.exp(-2 * (X + Y)), rtol=1e-13)
    np.testing.assert_allclose(oracle.Decaying(alpha=1.5)(X, Y), (1 + X + Y) ** -1.5, rtol=1e-14)
    a, b = np.minimum(X, Y), np.maximum(X, Y)
    np.testing.assert_allclose(oracle.WienerIntegral()(X, Y), a * a * (b - a / 3) / 2, rtol=1e-15)
    # H = 1/2, K = 1: the Wiener kernel; the stationary version at H = 1/2 is the triangle (1 - |d|)_+
    np.testing.assert_allclose(oracle.FracBrownian()(X, Y), np.minimum(X, Y), rtol=1e-14, atol=1e-16)
    np.testing.assert_allclose(oracle.StationaryFracBrownian()(X, Y), np.maximum(0, 1 - np.abs(X - Y)), rtol=1e-14, atol=1e-16)
    np.testing.assert_allclose(oracle.FracBrownian(H=0.8, K=0.6)(X, Y), ((X ** 1.6 + Y ** 1.6) ** 0.6 - np.abs(X - Y) ** 0.96) / 2 ** 0.6, rtol=1e-14)
    # Linear over fields; domains
    s = np.empty(5, [('u', float), ('v', float)]); s['u'] = np.arange(5.); s['v'] = 2.
    np.testing.assert_array_equal(oracle.Linear()(s[:, None], s[None, :]), np.outer(np.arange(5.), np.arange(5.)) + 4)
    with pytest.raises(AssertionError):
        oracle.Wiener()(np.array([-1.])[:, None], y[None, :])
    with pytest.raises(AssertionError):
        oracle.BrownianBridge()(np.array([1.5])[:, None], y[None, :])


def test_oscillator_and_circular_closed_forms(rng):
    """ Celerite, Harmonic, Circular of the oracle against ``_kernels/_celerite.py:34-121`` / ``_wendland.py:88-113``: the
    formulas of the docstrings and the relations they state """
    x = rng.uniform(-3, 3, 50)
    d = x[:, None] - x[None, :]
    a = np.abs(d)
    np.testing.assert_allclose(oracle.Celerite(gamma=0.8, B=-0.3)(x[:, None], x[None, :]), np.exp(-0.8 * a) * (np.cos(d) - 0.3 * np.sin(a)), rtol=1e-13, atol=1e-15)
    for Q in (0.2, 0.49, 0.5, 0.8):                       # both branches below 1 are exp(-tau / Q) (cosh + sinh / (eta Q))
        eta = np.sqrt(1 / Q ** 2 - 1)
        want = np.exp(-a / Q) * (np.cosh(eta * a) + np.sinh(eta * a) / (eta * Q))
        np.testing.assert_allclose(oracle.Harmonic(Q=Q)(x[:, None], x[None, :]), want, rtol=1e-11)
    np.testing.assert_allclose(oracle.Harmonic()(x[:, None], x[None, :]), (1 + a) * np.exp(-a), rtol=1e-14)
    Q = 4.
    eta = np.sqrt(1 - 1 / Q ** 2)
    np.testing.assert_allclose(oracle.Harmonic(Q=Q)(x[:, None], x[None, :]), np.exp(-a / Q) * (np.cos(eta * a) + np.sin(eta * a) / (eta * Q)), rtol=1e-12, atol=1e-15)
    # Circular: period 1, W_c of the geodesic distance
    K = oracle.Circular(tau=5, c=0.4)(x[:, None], x[None, :])
    np.testing.assert_allclose(K, oracle.Circular(tau=5, c=0.4)((x + 3.)[:, None], x[None, :]), rtol=0, atol=1e-12)
    t = np.arccos(np.cos(2 * np.pi * d)) / (2 * np.pi)
    np.testing.assert_allclose(K, (1 + 5 * t / 0.4) * np.maximum(1 - t / 0.4, 0) ** 5, rtol=0, atol=1e-12)

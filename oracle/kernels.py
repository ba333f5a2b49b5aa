"""
CPU restatement (numpy/scipy) of the three kernels on the hot path and of the
thin slice of kernel algebra the GP evaluates them through -- test
infrastructure only, see ``oracle/__init__.py``.  parity unpinned vs JAX-CPU.

Follows (reference paths):
``src/lsqfitgp/_Kernel/_crosskernel.py:152-200`` (call contract, ctor xtransf chain),
``_Kernel/_ops.py:292-326`` (loc, scale), ``_Kernel/_alg.py:32-82`` (scalar add/mul),
``_Kernel/_isotropic.py:61-81`` + ``_Kernel/_util.py:74-99`` (r^2 over fields),
``_kernels/_basic.py:61-75`` (ExpQuad), ``:94-115`` (GammaExp), ``:315-344`` (Cauchy), ``_kernels/_matern.py:29-49`` +
``_special/_bessel.py:101-110`` (Maternp), ``_kernels/_bart.py`` (BART),
``_jaxext/_fasthash.py:56-102`` (fasthash64/32).
"""

import collections
import numbers
import re

import numpy
from numpy.lib import recfunctions
from scipy import linalg, special


# --------------------------------------------------------------------------
# structured-array helpers (``_Kernel/_util.py:74-117``)

def _leaf_apply(fun, x):
    """ ``_util.py:104-117`` ufunc_recurse_dtype: apply to every leaf field """
    if x.dtype.names is None:
        return fun(x)
    out = numpy.empty(x.shape, x.dtype)
    for name in x.dtype.names:
        out[name] = _leaf_apply(fun, x[name])
    return out


def _to_float_struct(x):
    """ promote integer leaves to float so that (x - loc)/scale is a true division """
    if x.dtype.names is None:
        return x if numpy.issubdtype(x.dtype, numpy.floating) else x.astype(float)
    return x


def sum_recurse_dtype(fun, x, y):
    """ ``_util.py:74-99``: sum over fields, left to right, of fun(x_f, y_f);
    sub-array fields are summed over their trailing axes first """
    if x.dtype.names is None:
        return fun(x, y)
    acc = None
    for name in x.dtype.names:
        result = sum_recurse_dtype(fun, x[name], y[name])
        dtype = x.dtype[name]
        if dtype.ndim:
            result = numpy.sum(result, axis=tuple(range(-dtype.ndim, 0)))
        acc = result if acc is None else acc + result
    return acc


def _sp():
    import sympy
    return sympy


# --------------------------------------------------------------------------
# kernel objects

class Kernel:
    """
    Minimal stand-in for ``CrossKernel`` (``_crosskernel.py:54-200``): wraps a
    ``core(x, y)`` and applies ``(x - loc) / scale`` to both arguments
    (``:172-185``; see ``_transf_inputs`` for the order).
    """

    def __init__(self, core, *, scale=None, loc=None, dim=None, outops=()):
        self._core = core
        self._scale = scale
        self._loc = loc
        self._dim = dim                                   # ``_Kernel/_ops.py:233-260``: T(f)(x) = f(x[dim]), applied first
        # output algops applied in order: ('mul', c) is kernel * c
        # (``_alg.py:58-82``), ('add', c) is kernel + c (``_alg.py:32-56``)
        self._outops = tuple(outops)

    def _transf_inputs(self, x):
        # ``_crosskernel.py:172-185`` applies linop('scale') FIRST, then
        # linop('loc') (dict order); each linop wraps the core as
        # core'(x) = core(T(x)): k1(x) = k0(x / scale), then
        # k2(x) = k1(x - loc) = k0((x - loc) / scale).
        if self._dim is not None:
            if x.dtype.names is None:
                raise ValueError(f'cannot get dim={self._dim!r} from non-structured input')
            x = x[[self._dim]] if (isinstance(self._dim, str) and x.dtype[self._dim].shape) else x[self._dim]
        if self._loc is not None:
            l = self._loc
            x = _leaf_apply(lambda v: v - l, x)
        if self._scale is not None:
            s = self._scale
            x = _leaf_apply(lambda v: v / s, _to_float_struct(x))
        return x

    def __call__(self, x, y):
        """ ``_crosskernel.py:192-200`` """
        x = numpy.asarray(x)
        y = numpy.asarray(y)
        shape = numpy.broadcast_shapes(x.shape, y.shape)
        x = self._transf_inputs(x)
        y = self._transf_inputs(y)
        result = numpy.asarray(self._core(x, y))
        for op, c in self._outops:
            result = result * c if op == 'mul' else result + c
        assert result.shape == shape, (result.shape, shape)
        return result

    def diff(self, xderiv, yderiv):
        """ ``_Kernel/_ops.py:115-211`` (`diff`): the kernel derived ``xderiv`` times in its first argument and ``yderiv`` times in
        its second; each an integer (the implicit variable of plain one-dimensional points) or a dictionary field -> order.  The
        reference differentiates the transformed core with jax (``elementwise_grad``); here the derivatives are written out for
        the two cores that have them in closed form:

        * ExpQuad (``_kernels/_basic.py:75``), any fields: along field f, d^a/dx^a d^b/dy^b exp(-|u|²/2), u = (x - y)/scale, is
          (-1)^a / scale^(a+b) He_{a+b}(u_f) exp(-|u|²/2) (probabilists' Hermite polynomials), the fields multiply;
        * Maternp (``_kernels/_matern.py:46-49``, derivable p times per side ``:26-27``), implicit variable: with t = c |u|,
          c = sqrt(2p+1), K(u) = g(t), g = exp(-t) P_p: d^a/dx^a d^b/dy^b k = (-1)^b (c sign(u) / scale)^(a+b) g^(a+b)(t), and by
          Leibniz g^(q) = exp(-t) sum_j C(q, j) (-1)^(q-j) P^(j).

        Scalar factors of the kernel multiply the derivative, added constants drop out. """
        def norm(dv):
            if isinstance(dv, dict):
                return {k: int(v) for k, v in dv.items() if v}
            return {None: int(dv)} if int(dv) else {}
        da, db = norm(xderiv), norm(yderiv)
        assert all(v >= 0 for v in list(da.values()) + list(db.values()))
        if not da and not db:
            return self
        kind = getattr(self._core, 'closed_form', None)
        if kind == 'maternp' and (set(da) | set(db)) != {None}:
            kind = None                                             # along named fields: the symbolic route (the closed form below is 1-D)
        if kind is None and getattr(self._core, 'symbolic', None) is not None:
            from .symdiff import symbolic_diff                      # sympy + mpmath: the cores with an elementary expression
            return symbolic_diff(self, da, db, Kernel)
        if kind is None:
            raise NotImplementedError('derivatives in closed form for ExpQuad and Maternp only')
        scale = 1. if self._scale is None else self._scale
        factor = 1.
        for op, c in self._outops:
            if op == 'mul':
                factor = factor * c
        fields = sorted(set(da) | set(db), key=lambda k: (k is not None, k))
        order = sum(da.values()) + sum(db.values())
        if kind == 'expquad':
            factor = factor * (-1.) ** sum(da.values()) / scale ** order

            def core(x, y):
                x = numpy.asarray(x); y = numpy.asarray(y)
                if fields == [None]:
                    assert x.dtype.names is None and y.dtype.names is None, 'derivative along the implicit variable of plain arrays'
                    u = {None: numpy.asarray(x, float) - numpy.asarray(y, float)}
                    r2 = numpy.square(u[None])
                else:
                    assert x.dtype.names is not None and all(f in x.dtype.names for f in fields), 'derivative along missing field'
                    u = {f: x[f] - y[f] for f in fields}
                    r2 = sum_recurse_dtype(lambda a, b: numpy.square(a - b), x, y)
                he = 1.
                for f in fields:
                    q = da.get(f, 0) + db.get(f, 0)
                    he = he * numpy.polynomial.hermite_e.hermeval(u[f], [0] * q + [1])
                return he * numpy.exp(-1/2 * r2)
            return Kernel(core, scale=self._scale, loc=self._loc, outops=(('mul', factor),))
        # Maternp
        p = self._core.p
        assert fields == [None], 'Maternp derivatives along the implicit variable of plain arrays'
        a, b = da.get(None, 0), db.get(None, 0)
        if a > p or b > p:
            raise ValueError(f'Maternp(p={p}) is derivable {p} times on each side')
        q = a + b
        cc = numpy.sqrt(2 * p + 1)
        factor = factor * (-1.) ** b * (cc / scale) ** q
        # coefficients of P_p, highest power first (numpy.polyder's convention), from the Horner recurrence of _bessel.py:106-109
        P = numpy.array([1.])
        for k in reversed(range(p)):
            ck = (p - k) / ((2 * p - k) * (k + 1)) * 2
            P = numpy.polyadd(numpy.polymul(P, [ck, 0.]), [1.])
        terms = [special.comb(q, j) * (-1.) ** (q - j) * (numpy.polyder(P, j) if j else P) for j in range(q + 1)]

        def core(x, y):
            x = numpy.asarray(x, float); y = numpy.asarray(y, float)
            assert x.dtype.names is None and y.dtype.names is None
            u = x - y
            t = numpy.sqrt((2 * p + 1) * numpy.square(u) + 1e-30)
            g = sum(numpy.polyval(term, t) for term in terms) * numpy.exp(-t)
            return numpy.sign(u) ** q * g if q % 2 else g
        return Kernel(core, scale=self._scale, loc=self._loc, outops=(('mul', factor),))

    def _with(self, op, c):
        if isinstance(c, Kernel):
            # k1 + k2, k1 * k2 (``_alg.py:32-82``): each operand with its own input transformations
            a, b = self, c
            core = (lambda x, y: a(x, y) + b(x, y)) if op == 'add' else (lambda x, y: a(x, y) * b(x, y))
            return Kernel(core)
        if not isinstance(c, numbers.Number) and numpy.ndim(c) != 0:
            return NotImplemented
        return Kernel(self._core, scale=self._scale, loc=self._loc, dim=self._dim, outops=self._outops + ((op, c),))

    def __mul__(self, c):
        return self._with('mul', c)

    __rmul__ = __mul__

    def __add__(self, c):
        return self._with('add', c)

    __radd__ = __add__


def _isotropic(core):
    """ ``_isotropic.py:61-81`` with input='squared' """
    def newcore(x, y):
        r2 = sum_recurse_dtype(lambda a, b: numpy.square(a - b), x, y)
        return core(r2)
    return newcore


def expquad_core(r2):
    """ ``_kernels/_basic.py:75`` """
    return numpy.exp(-1/2 * r2)


def kvmodx2_hi(x2, p):
    """ ``_special/_bessel.py:103-110`` """
    x = numpy.sqrt(x2)
    poly = 1
    for k in reversed(range(p)):
        c_kp1_over_ck = (p - k) / ((2 * p - k) * (k + 1))
        poly = 1 + poly * c_kp1_over_ck * 2 * x
    return numpy.exp(-x) * poly


def maternp_core(r2, p):
    """ ``_kernels/_matern.py:46-49`` """
    assert int(p) == p and p >= 0, p
    r2 = (2 * p + 1) * r2
    return kvmodx2_hi(r2 + 1e-30, p)


def gammaexp_core(r2, gamma=1):
    """ ``_kernels/_basic.py:94-115``: exp(-r^gamma), gamma in (0, 2]; the gamma == 2 branch is evaluated on r2 """
    assert 0 < gamma <= 2, gamma
    nondiff = numpy.exp(-(r2 ** (gamma / 2)))
    diff = numpy.exp(-r2)
    return numpy.where(gamma == 2, diff, nondiff)


def cauchy_core(r2, alpha=2, beta=2):
    """ ``_kernels/_basic.py:315-344``: (1 + r^alpha / beta)^(-beta / alpha), alpha in (0, 2], beta > 0 """
    assert 0 < alpha <= 2, alpha
    assert 0 < beta, beta
    power = numpy.where(alpha == 2, r2, r2 ** (alpha / 2))
    return (1 + power / beta) ** (-beta / alpha)


def ExpQuad(*, scale=None, loc=None, dim=None):
    core = _isotropic(expquad_core)
    core.closed_form = 'expquad'          # Kernel.diff has its derivatives in closed form
    return Kernel(core, scale=scale, loc=loc, dim=dim)


def GammaExp(*, gamma=1, scale=None, loc=None, dim=None):
    core_ = _isotropic(lambda r2: gammaexp_core(r2, gamma))
    core_.symbolic = ('iso', lambda r2: _sp().exp(-r2)) if gamma == 2 else None                    # derivable iff gamma = 2 (_basic.py:94)
    return Kernel(core_, scale=scale, loc=loc, dim=dim)


def Cauchy(*, alpha=2, beta=2, scale=None, loc=None, dim=None):
    core_ = _isotropic(lambda r2: cauchy_core(r2, alpha, beta))
    core_.symbolic = ('iso', lambda r2: (1 + r2 / beta) ** (-_sp().Float(beta, 30) / 2)) if alpha == 2 else None       # derivable iff alpha = 2 (_basic.py:312-315)
    return Kernel(core_, scale=scale, loc=loc, dim=dim)


def Maternp(*, p=None, scale=None, loc=None, dim=None):
    p = int(p)
    core = _isotropic(lambda r2: maternp_core(r2, p))
    core.closed_form = 'maternp'
    core.p = p

    def sym(r2):                                             # exp(-t) P_p(t), t = sqrt((2p+1) r2): along named fields (symdiff.py)
        sp = _sp()
        t = sp.sqrt((2 * p + 1) * r2)
        poly = sp.Integer(1)
        for k in reversed(range(p)):
            poly = 1 + poly * sp.Rational(p - k, (2 * p - k) * (k + 1)) * 2 * t
        return sp.exp(-t) * poly
    core.symbolic = ('iso', sym)
    return Kernel(core, scale=scale, loc=loc, dim=dim)


def Matern(*, nu=None, scale=None, loc=None, dim=None):
    """ ``_kernels/_matern.py:54-76``: 2^(1-nu)/Gamma(nu) x^nu K_nu(x), x = sqrt(2 nu) r, with scipy's K_nu (the reference has its own
    series, ``_special/_bessel.py``) """
    assert 0 <= nu < numpy.inf

    def core(r2):
        x = numpy.sqrt(2 * (nu if nu else 1) * r2)
        with numpy.errstate(invalid='ignore', divide='ignore', over='ignore'):
            val = 2 ** (1 - nu) / special.gamma(nu) * x ** nu * special.kv(nu, x)
        return numpy.where(x == 0, 1., val)
    inner = _isotropic(core)

    def sym(r2):                                             # derivative observations (symdiff.py): sympy differentiates K_nu itself
        sp = _sp()
        v = sp.Rational(float(nu))                           # the double, exactly
        x = sp.sqrt(2 * v * r2)
        return 2 ** (1 - v) / sp.gamma(v) * x ** v * sp.besselk(v, x)
    # ceil(nu) - 1 times derivable (_matern.py:51-53).  The derivatives at a coincidence are limits of sums of terms up to
    # x^-(nu + 4) in size that approach them like x^(2 (nu - order)): far smaller an offset, far more digits, than the elementary cores
    inner.symbolic = ('iso', sym, None, dict(dps=400, offset='1e-40')) if nu > 1 else None
    return Kernel(inner, scale=scale, loc=loc, dim=dim)


def Bessel(*, nu=0, scale=None, loc=None, dim=None):
    """ ``_kernels/_matern.py:92-119`` through ``_special/_bessel.py:45-50`` (scipy's jv) """
    assert 0 <= nu < numpy.inf, nu
    maxdim = 2 * (int(numpy.floor(nu)) + 1)

    def core(r2):
        x = numpy.sqrt(r2 * (2 + nu / 2) ** 2)
        with numpy.errstate(all='ignore'):
            normal = (x / 2) ** -nu * special.jv(nu, x)
        return special.gamma(nu + 1) * numpy.where(x, normal, 1 / special.gamma(nu + 1))
    inner = _isotropic(core)

    def checked(x, y):
        if x.dtype.names is not None and len(_leaf_columns_count(x.dtype)) > maxdim:
            raise ValueError(f'kernel called on type with dimensionality > maxdim={maxdim}')
        return inner(x, y)

    def sym(r2):                                             # derivative observations (symdiff.py); int(nu // 2) times derivable (_matern.py:92-94)
        sp = _sp()
        v = sp.Rational(float(nu))
        x = sp.sqrt(r2) * (2 + v / 2)
        return sp.gamma(v + 1) * (x / 2) ** (-v) * sp.besselj(v, x)
    checked.symbolic = ('iso', sym, None, dict(dps=200, offset='1e-20')) if nu >= 2 else None
    return Kernel(checked, scale=scale, loc=loc, dim=dim)


def _isotropic_posabs(core):
    """ ``_isotropic.py:61-81`` with input='posabs': per field (|x - y| + eps)^2 (``_stationary.py:84-85`` _softabs), summed, the
    core takes the square root """
    def newcore(x, y):
        def dist(a, b):
            d = numpy.abs(a - b)
            return numpy.square(d + numpy.finfo(d.dtype if numpy.issubdtype(d.dtype, numpy.floating) else float).eps)
        return core(numpy.sqrt(sum_recurse_dtype(dist, x, y)))
    return newcore


_WENDLAND_POLY = {0: [[1]], 1: [[1, 1], [1]], 2: [[1/3, 4/3, 1], [1, 2], [1]], 3: [[1/15, 3/5, 23/15, 1], [2/5, 12/5, 3], [1, 3], [1]]}


def Wendland(*, k=0, alpha=1, scale=None, loc=None, dim=None):
    """ ``_kernels/_wendland.py:32-84`` """
    assert int(numpy.floor(2 * alpha - 1)) >= 1
    nu = k + alpha
    coeffs = numpy.array([numpy.polyval(numpy.array(pj), nu) for pj in _WENDLAND_POLY[k]])

    def core(r):
        poly = numpy.polyval(coeffs, r)
        with numpy.errstate(invalid='ignore'):
            return numpy.where(r < 1, (1 - r) ** (nu + k) * poly, 0)
    inner = _isotropic_posabs(core)
    maxdim = int(numpy.floor(2 * alpha - 1))                       # ``_wendland.py:29-31``

    def checked(x, y):
        if x.dtype.names is not None and len(_leaf_columns_count(x.dtype)) > maxdim:
            raise ValueError(f'kernel called on type with dimensionality > maxdim={maxdim}')
        return inner(x, y)
    def sym(a):                                              # one-dimensional points: r = |difference|
        sp = _sp()
        poly = sum(sp.Float(float(cf), 30) * a ** (len(coeffs) - 1 - i) for i, cf in enumerate(coeffs))
        return (1 - a) ** sp.Float(nu + k, 30) * poly
    checked.symbolic = ('abs', sym, 1.)
    return Kernel(checked, scale=scale, loc=loc, dim=dim)


def CausalExpQuad(*, alpha=1, scale=None, loc=None, dim=None):
    """ ``_kernels/_basic.py:346-360`` """
    assert alpha >= 0
    core = _isotropic_posabs(lambda r: special.erfc(alpha / 4 * r) * numpy.exp(-1/2 * numpy.square(r)))
    # derivable iff alpha = 0 (_basic.py:346): then exp(-r^2/2) of the softened distance (|d| + eps per field: not seen at 1e-9)
    core.symbolic = ('iso', lambda r2: _sp().exp(-r2 / 2)) if alpha == 0 else None
    return Kernel(core, scale=scale, loc=loc, dim=dim)


def Log(*, scale=None, loc=None, dim=None):
    """ ``_kernels/_basic.py:381-392`` """
    return Kernel(_isotropic_posabs(lambda r: numpy.log1p(r) / r), scale=scale, loc=loc, dim=dim)


def Constant(*, scale=None, loc=None, dim=None):
    """ ``_kernels/_basic.py:34-46`` """
    def core(x, y):
        return numpy.ones(numpy.broadcast_shapes(x.shape, y.shape))
    core.symbolic = ('iso', lambda r2: _sp().Integer(1) + 0 * r2)
    return Kernel(core, scale=scale, loc=loc, dim=dim)


def White(*, scale=None, loc=None, dim=None):
    """ ``_kernels/_basic.py:48-59``: the product over the fields of (x_f == y_f) """
    def core(x, y):
        if x.dtype.names is None:
            return (x == y).astype(int)
        acc = None
        for name in x.dtype.names:
            eq = (x[name] == y[name])
            if x.dtype[name].ndim:
                eq = numpy.prod(eq, axis=tuple(range(-x.dtype[name].ndim, 0)))
            acc = eq if acc is None else acc * eq
        return acc.astype(int)
    return Kernel(core, scale=scale, loc=loc, dim=dim)


def _stationary1d(core, input='signed'):
    """ ``_Kernel/_stationary.py:52-66`` for points of one variable (maxdim = 1): the core takes x - y, |x - y| ('abs') or
    |x - y| + eps ('posabs', ``:84-85``) """
    def newcore(x, y):
        if x.dtype.names is not None:
            if len(_leaf_columns_count(x.dtype)) > 1:
                raise ValueError('kernel called on type with dimensionality > maxdim=1')
            name = x.dtype.names[0]
            x, y = x[name], y[name]
        q = x - y
        if input == 'abs':
            q = numpy.abs(q)
        elif input == 'posabs':
            q = numpy.abs(q) + numpy.finfo(q.dtype if numpy.issubdtype(q.dtype, numpy.floating) else float).eps
        return core(q)
    return newcore


def _leaf_columns_count(dtype):
    out = []
    for name in dtype.names:
        sub = dtype[name]
        if sub.names is not None:
            out += _leaf_columns_count(sub)
        else:
            out += [name] * max(1, int(numpy.prod(sub.shape, dtype=int)))
    return out


def Expon(*, scale=None, loc=None, dim=None):
    """ ``_kernels/_basic.py:254-269`` """
    return Kernel(_stationary1d(lambda d: numpy.exp(-d), 'abs'), scale=scale, loc=loc, dim=dim)


def HoleEffect(*, scale=None, loc=None, dim=None):
    """ ``_kernels/_basic.py:299-310`` """
    return Kernel(_stationary1d(lambda d: (1 - d) * numpy.exp(-d), 'abs'), scale=scale, loc=loc, dim=dim)


def Periodic(*, outerscale=1, scale=None, loc=None, dim=None):
    """ ``_kernels/_basic.py:187-208`` """
    assert 0 < outerscale < numpy.inf
    core_ = _stationary1d(lambda d: numpy.exp(-2 * (numpy.sin(d / 2) / outerscale) ** 2))
    core_.symbolic = ('delta', lambda d: _sp().exp(-2 * (_sp().sin(d / 2) / outerscale) ** 2))
    return Kernel(core_, scale=scale, loc=loc, dim=dim)


def Cos(*, scale=None, loc=None, dim=None):
    """ ``_kernels/_spectral.py:25-38`` """
    core_ = _stationary1d(numpy.cos)
    core_.symbolic = ('delta', lambda d: _sp().cos(d))
    return Kernel(core_, scale=scale, loc=loc, dim=dim)


def sinc(x):
    """ ``_special/_sinc.py:25-32``: six Taylor terms in (pi x)^2 below |x| = 0.1, sin(pi x) / (pi x) elsewhere """
    z = numpy.square(numpy.pi * x)
    m = numpy.arange(6)
    coef = (-1.) ** m / numpy.exp(special.gammaln(2 + 2 * m))
    near = numpy.polyval(coef[::-1], z)
    return numpy.where(numpy.abs(x) < 1e-1, near, numpy.sinc(x))


def Sinc(*, scale=None, loc=None, dim=None):
    """ ``_kernels/_spectral.py:113-122`` """
    core_ = _stationary1d(sinc, 'posabs')
    core_.symbolic = ('delta', lambda d: _sp().sin(_sp().pi * d) / (_sp().pi * d))
    return Kernel(core_, scale=scale, loc=loc, dim=dim)


def StationaryFracBrownian(*, H=1/2, scale=None, loc=None, dim=None):
    """ ``_kernels/_randomwalk.py:156-172`` """
    assert 0 < H <= 1, H
    H2 = 2 * H
    core_ = _stationary1d(lambda d: 1/2 * (numpy.abs(d + 1) ** H2 + numpy.abs(d - 1) ** H2 - 2 * numpy.abs(d) ** H2))
    core_.symbolic = ('delta', lambda d: _sp().Integer(1) + 0 * d) if H == 1 else None
    return Kernel(core_, scale=scale, loc=loc, dim=dim)


def Celerite(*, gamma=1, B=0, scale=None, loc=None, dim=None):
    """ ``_kernels/_celerite.py:34-52`` """
    assert 0 <= gamma < numpy.inf and abs(B) <= gamma
    core_ = _stationary1d(lambda d: numpy.exp(-gamma * d) * (numpy.cos(d) + B * numpy.sin(d)), 'abs')
    core_.symbolic = ('abs', lambda a: _sp().exp(-gamma * a) * (_sp().cos(a) + B * _sp().sin(a)))
    return Kernel(core_, scale=scale, loc=loc, dim=dim)


def Harmonic(*, Q=1, scale=None, loc=None, dim=None):
    """ ``_kernels/_celerite.py:54-121`` """
    assert 0 < Q < numpy.inf

    def core(delta):
        tau = numpy.abs(delta)
        if Q < 1/2:
            etaQ = numpy.sqrt((1 - Q) * (1 + Q))
            tauQ = tau / Q
            pexp = numpy.exp(numpy.expm1(1/2 * numpy.log1p(-numpy.square(Q))) * tauQ)
            mexp = numpy.exp(-(1 + etaQ) * tauQ)
            return (pexp + mexp + (pexp - mexp) / etaQ) / 2
        elif 1/2 <= Q < 1:
            etaQ = numpy.sqrt(1 - numpy.square(Q))
            tauQ = tau / Q
            etatau = etaQ * tauQ
            return numpy.exp(-tauQ) * (numpy.cosh(etatau) + numpy.sinh(etatau) / etaQ)
        elif Q == 1:
            x = tau
            return (1 + x / Q) * numpy.exp(-x / Q) + numpy.exp(-x / Q) * (1 - Q) * numpy.square(x) * (1 + x / 3)
        else:
            etaQ = numpy.sqrt(numpy.square(Q) - 1)
            tauQ = tau / Q
            etatau = etaQ * tauQ
            return numpy.exp(-tauQ) * (numpy.cos(etatau) + numpy.sin(etatau) / etaQ)

    def sym(a):
        sp = _sp()
        Qs = sp.Float(Q, 30)
        tq = a / Qs
        if Q < 1/2:
            e = sp.sqrt((1 - Qs) * (1 + Qs))
            pexp, mexp = sp.exp((sp.sqrt(1 - Qs ** 2) - 1) * tq), sp.exp(-(1 + e) * tq)
            return (pexp + mexp + (pexp - mexp) / e) / 2
        if Q < 1:
            e = sp.sqrt(1 - Qs ** 2)
            return sp.exp(-tq) * (sp.cosh(e * tq) + sp.sinh(e * tq) / e)
        if Q == 1:
            return (1 + a) * sp.exp(-a)
        e = sp.sqrt(Qs ** 2 - 1)
        return sp.exp(-tq) * (sp.cos(e * tq) + sp.sin(e * tq) / e)
    core_ = _stationary1d(core)
    core_.symbolic = ('abs', sym)
    return Kernel(core_, scale=scale, loc=loc, dim=dim)


def Circular(*, tau=4, c=1/2, scale=None, loc=None, dim=None):
    """ ``_kernels/_wendland.py:88-113`` """
    assert tau >= 4 and 0 < c <= 1/2

    def core(delta):
        x = delta % 1
        t = numpy.minimum(x, 1 - x)
        return (1 + tau * t / c) * numpy.maximum(1 - t / c, 0) ** tau
    core_ = _stationary1d(core, 'posabs')
    core_.symbolic = ('circ', lambda t: (1 + tau * t / c) * (1 - t / c) ** _sp().Float(tau, 30), c)
    return Kernel(core_, scale=scale, loc=loc, dim=dim)


def Taylor(*, scale=None, loc=None, dim=None):
    """ ``_kernels/_basic.py:393-408`` """
    def core(x, y):
        mul = x * y
        val = 2 * numpy.sqrt(numpy.abs(mul))
        return numpy.where(mul >= 0, special.i0(val), special.j0(val))
    core_ = _onevariable(core, -numpy.inf, numpy.inf)
    # I0(2 sqrt(w)) = 0F1(; 1; w), w = x y, J0 where w < 0: one entire function of w, derivable any number of times (_basic.py:393)
    core_.symbolic = ('xy', lambda X, Y, upper: _sp().hyper((), (1,), X * Y))
    return Kernel(core_, scale=scale, loc=loc, dim=dim)


def exp1_imag(x):
    """ E_1(-ix) = -Ci(x) + i (pi/2 - Si(x)) (``_special/_expint.py:238-262``; the reference evaluates Si and Ci by the rational
    approximations of Rowe et al. 2015, scipy's sici here) """
    si, ci = special.sici(x)
    return -ci + 1j * (numpy.pi / 2 - si)


def expn_imag(n, x):
    """ E_n(-ix), integer n >= 2, x >= 0 (``_special/_expint.py:31-64``): the closed form DLMF 8.19.7 below the knee
    (Γ(n + nt) / eps)^(1 / (n + nt - 1)), the asymptotic series 8.20.2 with nt = 20 terms above """
    x = numpy.asarray(x, float)
    nt = 20
    eps = numpy.finfo(float).eps
    knee = (special.gamma(n + nt) / eps) ** (1 / (n + nt - 1))
    with numpy.errstate(all='ignore'):
        # small x (``:76-90``)
        fact = numpy.cumprod(numpy.concatenate([[1.], numpy.arange(1., n)]))       # 0!, 1!, ..., (n-1)!
        ix = 1j * x
        e1 = numpy.where(x, exp1_imag(numpy.where(x, x, 1.)), 0)
        small = (ix ** (n - 1) * e1 + numpy.exp(ix) * numpy.polyval(fact[:-1], ix)) / fact[-1]
        # large x (``:93-106``): e^-z / z sum_k (-1)^k (n)_k / z^k
        z = -1j * x
        k = numpy.arange(nt)
        coef = (-1.) ** k * numpy.exp(special.gammaln(n + k) - special.gammaln(n))
        large = numpy.exp(-z) / z * numpy.polyval(coef[::-1], 1 / z)
    return numpy.where(x < knee, small, large)


def Color(*, n=2, scale=None, loc=None, dim=None):
    """ ``_kernels/_spectral.py:73-111`` """
    assert int(n) == n and n >= 2, n
    core_ = _stationary1d(lambda delta: (n - 1) * expn_imag(n, delta).real, 'abs')
    # n // 2 - 1 times derivable (_spectral.py:70-71); the real part written as the mean of the two conjugate values (symdiff.py)
    def sym(a):
        sp = _sp()
        return (n - 1) * (sp.expint(n, -sp.I * a) + sp.expint(n, sp.I * a)) / 2
    core_.symbolic = ('abs', sym) if n >= 4 else None
    return Kernel(core_, scale=scale, loc=loc, dim=dim)


def Pink(*, dw=1, scale=None, loc=None, dim=None):
    """ ``_kernels/_spectral.py:41-66`` """
    def core(delta):
        with numpy.errstate(all='ignore'):
            l = special.sici(delta)[1]
            r = special.sici(delta * (1 + dw))[1]
            return numpy.where(delta * dw < numpy.sqrt(numpy.finfo(float).eps), numpy.cos(delta * (1 + dw / 2)), (r - l) / numpy.log1p(dw))
    core_ = _stationary1d(core, 'abs')
    # derivable once (_spectral.py:41).  Below delta dw = sqrt(eps) the reference's core IS cos(delta (1 + dw / 2)), and jax
    # differentiates the branch taken: the variance of the derivative at a point is (1 + dw / 2)^2, not the limit of the other branch
    def sym(a):
        sp = _sp()
        v = sp.Rational(float(dw))
        return (sp.Ci(a * (1 + v)) - sp.Ci(a)) / sp.log(1 + v)
    tol = numpy.sqrt(numpy.finfo(float).eps)
    core_.symbolic = ('abs', sym, None, dict(small=(tol / dw, lambda a: _sp().cos(a * (1 + _sp().Rational(float(dw)) / 2))))) if dw > 1e-6 else None
    return Kernel(core_, scale=scale, loc=loc, dim=dim)


def _eta_borwein(z, n=40):
    """ Dirichlet eta, P. Borwein's algorithm 2 (2000) """
    k = numpy.arange(n + 1)
    d = numpy.cumsum(n * numpy.exp(special.gammaln(n + k) - special.gammaln(n - k + 1) - special.gammaln(2 * k + 1) + k * numpy.log(4.)))
    kk = numpy.arange(n)
    return -numpy.sum((-1.) ** kk * (d[kk] - d[n]) / (kk + 1.) ** z) / d[n]


def riemann_zeta(z):
    """ ζ(z), real z != 1 (the reference has its own ``_special/_zeta.py:206-264``): scipy above 1, eta on [0, 1), reflection below """
    if z > 1:
        return float(special.zeta(z, 1))
    if z >= 0:
        return _eta_borwein(z) / (1 - 2. ** (1 - z))
    return 2 * (2 * numpy.pi) ** (z - 1) * numpy.sin(numpy.pi * z / 2) * special.gamma(1 - z) * riemann_zeta(1 - z)


def periodic_zeta_real(x, s, nmax=62):
    """ Re Li_s(e^{2πix}), s > 1 (``_special/_zeta.py:76-177``).  s >= 10: the Fourier series, 50 terms (``:96-105,121-131``).  Below:
    DLMF 25.13.2 with ζ(1-s, x) + ζ(1-s, 1-x) = x^(s-1) + 2 Σ_{n even} (1-s)_n / n! ζ(n+1-s) x^n (25.11.3, 25.11.10; ``:133-177``,
    ``:34-74``).  At odd integer s that form is 0 times infinity (the reference moves s by one part in 2^52 and guards the
    cancellation); there, the limit in closed form: (-1)^m θ^(2m) (H_2m - log θ) / (2m)! + Σ_{n != m} (-1)^n ζ(2m+1-2n) θ^(2n) / (2n)! """
    x = numpy.asarray(x, float) % 1
    x = numpy.where(x > 0.5, 1 - x, x)
    k = round(s)
    if (1 < s < 10 and s != k and abs(s - k) < 2e-4 or 1 < s < 1.0002) and numpy.all((x == 0) | (x > 1e-2)):
        # (mpmath's polylog takes minutes per point next to z = 1: only away from it)
        # Next to an integer the two pieces of 25.13.2 cancel (next to an odd one both have poles); the reference guards this with
        # series of its own (``_special/_zeta.py:150-200``, zeta_series_power_diff).  Not restated: the checker evaluates the
        # polylogarithm itself in extended precision, as the reference's tests do for their golden values
        # (``tests/test_special.py:170-188``); pinned by those values, tests/test_reference_special.py
        import mpmath
        with mpmath.workdps(32):
            f = numpy.frompyfunc(lambda t: float(mpmath.zeta(s)) if t == 0 else float(mpmath.polylog(s, mpmath.exp(2j * mpmath.pi * mpmath.mpf(t))).real), 1, 1)
            return f(x).astype(float)
    if s >= 10:
        n = numpy.arange(1, 51.)
        with numpy.errstate(over='ignore'):
            return numpy.sum(numpy.cos(2 * numpy.pi * n * x[..., None]) / n ** s, axis=-1)
    if s == round(s) and int(round(s)) % 2 == 1:
        m = (int(round(s)) - 1) // 2
        th = 2 * numpy.pi * x
        out = numpy.zeros(x.shape)
        for n in range(nmax // 2, -1, -1):
            out = out * th * th + (0. if n == m else (-1.) ** n * riemann_zeta(s - 2 * n) / special.gamma(2 * n + 1) if s - 2 * n > -20 else
                                   (-1.) ** n * 2 * (2 * numpy.pi) ** (s - 2 * n - 1) * numpy.sin(numpy.pi * (s - 2 * n) / 2) *
                                   numpy.exp(special.gammaln(1 - s + 2 * n) - special.gammaln(2 * n + 1)) * float(special.zeta(1 - s + 2 * n, 1)))
        with numpy.errstate(all='ignore'):
            extra = numpy.where(th > 0, (-1.) ** m * th ** (2 * m) / special.gamma(2 * m + 1) * (numpy.sum(1. / numpy.arange(1, 2 * m + 1)) - numpy.log(th)), 0.)
        return out + extra
    coef = numpy.pi * (2 * numpy.pi) ** (s - 1) / (2 * numpy.cos(numpy.pi * s / 2) * special.gamma(s))     # = Γ(s1) (2π)^-s1 cos(π s1 / 2)
    return coef * hurwitz_zeta_pair(1 - s, x, nmax)


def hurwitz_zeta_pair(s1, x, nmax=62):
    """ ζ(s1, x) + ζ(s1, 1 - x) for s1 < 0, 0 <= x <= 1/2: x^-s1 + 2 Σ_{n even} (s1)_n / n! ζ(n + s1) x^n (DLMF 25.11.3, 25.11.10;
    ``_special/_zeta.py:34-74,133-177``) """
    x = numpy.asarray(x, float)
    series = numpy.zeros(x.shape)
    for n in range(nmax, -1, -2):
        if n + s1 == 1:
            b = -1. / n                                              # the zero of the Pochhammer symbol against the pole of ζ
        else:
            b = numpy.prod(s1 + numpy.arange(n)) / special.gamma(n + 1) * riemann_zeta(n + s1)
        series = series * x * x + b
    with numpy.errstate(all='ignore'):
        power = numpy.where(x > 0, x ** -s1, 0.)
    return power + 2 * series


def Zeta(*, nu=None, scale=None, loc=None, dim=None):
    """ ``_kernels/_zeta.py:37-70`` """
    assert nu is not None and 0 <= nu < numpy.inf, nu
    s = 1 + 2 * nu

    def core(delta):
        if s > 1:
            return periodic_zeta_real(delta, s) / special.zeta(s, 1)
        return numpy.where(delta % 1, 0., 1.)
    core_ = _stationary1d(core)
    # ceil(nu) - 1 times derivable (_kernels/_zeta.py:32-35): the polylogarithm on the unit circle, its real part as the mean of the
    # two conjugate values, as a function of the distance t to the nearest integer (symdiff.py differentiates polylog itself)
    def sym(t):
        sp = _sp()
        sv = sp.Rational(float(s))
        return (sp.polylog(sv, sp.exp(2 * sp.pi * sp.I * t)) + sp.polylog(sv, sp.exp(-2 * sp.pi * sp.I * t))) / (2 * sp.zeta(sv))
    core_.symbolic = ('circ', sym) if nu > 1 else None
    return Kernel(core_, scale=scale, loc=loc, dim=dim)


def MA(*, w=None, norm=False, scale=None, loc=None, dim=None):
    """ ``_kernels/_arma.py:31-63``: cov = convolve(w, reversed w), looked up at delta + len(w) - 1, 0 beyond """
    w = numpy.asarray([] if w is None else w, float)
    assert w.ndim == 1

    def core(delta):
        assert numpy.issubdtype(delta.dtype, numpy.integer), 'the inputs must be integers'
        if not len(w):
            return numpy.zeros(delta.shape)
        cov = numpy.convolve(w, w[::-1])
        if norm:
            with numpy.errstate(invalid='ignore', divide='ignore'):
                cov = cov / cov[len(w) - 1]
        idx = delta + len(w) - 1
        return numpy.where(idx < len(cov), cov[numpy.minimum(idx, len(cov) - 1)], 0.)
    return Kernel(_stationary1d(core, 'abs'), scale=scale, loc=loc, dim=dim)


class AR:
    """ ``_kernels/_arma.py:65-640``: the three parametrisations; with roots, the reference's own evaluation -- amplitudes of
    the mixed power-exponentials |m|^l x_j^-|m| fitted to the first p + 1 Yule-Walker values (``:226-232,316-383,484-507``) """

    def __new__(cls, *, phi=None, gamma=None, maxlag=None, slnr=None, lnc=None, norm=False, scale=None, loc=None, dim=None):
        given = (phi is not None, gamma is not None, maxlag is not None, slnr is not None, lnc is not None)
        if given not in ((True, False, True, False, False), (False, True, True, False, False), (False, False, False, True, True)):
            def invalid(delta):                               # (the reference's core runs, and raises, when the kernel is called)
                raise ValueError('invalid set of specified parameters')
            return Kernel(_stationary1d(invalid, 'abs'), scale=scale, loc=loc, dim=dim)
        if slnr is not None:
            slnr, lnc = cls._process_roots(slnr, lnc)
            ph = cls.phi_from_roots(slnr, lnc)
            gam = cls.gamma_from_phi(ph)
            if norm:
                gam = gam / gam[0]
            ampl = cls.ampl_from_roots(slnr, lnc, gam)

            def core(delta):
                assert numpy.issubdtype(delta.dtype, numpy.integer), 'the inputs must be integers'
                return (cls._columns(slnr, lnc, delta.reshape(-1)) @ ampl).reshape(delta.shape)
        else:
            ph = cls.phi_from_gamma(gamma) if phi is None else numpy.asarray(phi, float)
            gam = cls.gamma_from_phi(ph) if gamma is None else numpy.asarray(gamma, float)
            if norm:
                with numpy.errstate(invalid='ignore', divide='ignore'):
                    gam = gam / gam[0]
            acf = cls.extend_gamma(gam, ph, maxlag + 1 - len(gam))

            def core(delta):
                assert numpy.issubdtype(delta.dtype, numpy.integer), 'the inputs must be integers'
                return numpy.where(delta < len(acf), acf[numpy.minimum(delta, len(acf) - 1)], numpy.nan)
        return Kernel(_stationary1d(core, 'abs'), scale=scale, loc=loc, dim=dim)

    @staticmethod
    def _process_roots(slnr, lnc):
        slnr = numpy.sort(numpy.asarray(slnr, float))
        lnc = numpy.asarray(lnc, complex)
        imag = numpy.abs(lnc.imag) % (2 * numpy.pi)
        imag = numpy.where(imag > numpy.pi, 2 * numpy.pi - imag, imag)
        return slnr, numpy.sort(lnc.real + 1j * imag)

    @staticmethod
    def phi_from_gamma(gamma):
        gamma = numpy.asarray(gamma, float)
        t, b = gamma[:-1], gamma[1:]
        return linalg.solve_toeplitz(t, b) if t.size else numpy.empty(0)

    @staticmethod
    def gamma_from_phi(phi):
        phi = numpy.asarray(phi, float)
        p = len(phi)
        m = numpy.arange(p + 1)[:, None]
        n = m.T
        padded = numpy.pad(phi, (1, 1))
        a = numpy.eye(p + 1) - (padded[numpy.clip(m + n, 0, p + 1)] + padded[numpy.clip(m - n, 0, p + 1)]) / numpy.where(n, 1, 2)
        b = numpy.zeros(p + 1)
        b[0] = 1
        return numpy.linalg.solve(a, b)

    @staticmethod
    def extend_gamma(gamma, phi, n):
        vals = list(numpy.asarray(gamma, float))
        phi = numpy.asarray(phi, float)
        for _ in range(max(0, n)):
            vals.append(float(sum(phi[k] * vals[-1 - k] for k in range(len(phi)))))
        return numpy.array(vals)

    @classmethod
    def phi_from_roots(cls, slnr, lnc):
        slnr, lnc = cls._process_roots(slnr, lnc)
        r = numpy.copysign(numpy.exp(-numpy.abs(slnr)), slnr)
        c = numpy.exp(-lnc)
        coef = numpy.atleast_1d(numpy.poly(numpy.sort(numpy.concatenate([r, c, c.conj()]))))
        return -coef.real[1:]

    @staticmethod
    def _columns(slnr, lnc, lag):
        """ the basis functions at the given lags, each divided by (about) its largest value over the lags 0 .. p
        (``_gamma_from_ampl_matmul``, ``_arma.py:316-383``) """
        lag = numpy.asarray(lag, float)
        p = len(slnr) + 2 * len(lnc)
        lagnorm = float(p)

        def logcol(root, lag, llag, repeat):
            return -root * lag + (repeat * llag if repeat else 0.)

        def lognorm(root, repeat):
            with numpy.errstate(divide='ignore', invalid='ignore'):
                maxnorm = repeat * (-1 + numpy.log(repeat / root)) if repeat else 0.
                defnorm = logcol(root, lagnorm, numpy.log(lagnorm) if lagnorm else 0., repeat)
                return maxnorm if (repeat / root if root else numpy.inf) <= lagnorm else defnorm

        cols = [numpy.where(lag, 0., 1.)]
        with numpy.errstate(divide='ignore'):
            llag = numpy.log(lag)
        prev, repeat = None, 0
        for root in slnr:
            repeat = repeat + 1 if root == prev else 0
            prev = root
            aroot = abs(root)
            cols.append(numpy.sign(root) ** lag * numpy.exp(logcol(aroot, lag, llag, repeat) - lognorm(aroot, repeat)))
        prev, repeat = None, 0
        for root in lnc:
            repeat = repeat + 1 if root == prev else 0
            prev = root
            col = numpy.exp(logcol(root, lag, llag, repeat) - lognorm(root.real, repeat))
            cols.append(col.real)
            if root.imag:
                cols.append(-col.imag)
            else:                                                     # a real "complex" root: a pair of identical real roots
                repeat += 1
                cols.append(col.real * lag)
        return numpy.stack(cols, axis=-1)

    @classmethod
    def ampl_from_roots(cls, slnr, lnc, gamma):
        mat = cls._columns(slnr, lnc, numpy.arange(len(gamma)))
        u, sv, vh = numpy.linalg.svd(mat)                             # ``_pseudo_solve``, ``_arma.py:306-312``
        cut = sv[0] * numpy.finfo(float).eps * len(mat) if sv.size else 0.
        inv = numpy.where(sv < cut, 0., 1. / numpy.where(sv, sv, 1.))
        return vh.conj().T @ (inv * (u.conj().T @ gamma))


_bow_regexp = re.compile('\\s|[!«»"“”‘’/()\'?¡¿„‚<>,;.\u003a-\u2013—]')      # ``_basic.py:270``; ':-–' there is a range (U+003A..U+2013)


def BagOfWords(*, scale=None, loc=None, dim=None):
    """ ``_kernels/_basic.py:272-296`` """
    def one(x, y):
        xbag = collections.Counter(_bow_regexp.split(str(x)))
        ybag = collections.Counter(_bow_regexp.split(str(y)))
        xbag[''] = 0
        ybag[''] = 0
        return sum(xbag[k] * ybag[k] for k in set(xbag) & set(ybag))

    def core(x, y):
        if x.dtype.names is not None:
            name = x.dtype.names[0]
            x, y = x[name], y[name]
        return numpy.vectorize(one, otypes=[float])(x, y)
    return Kernel(core, scale=scale, loc=loc, dim=dim)


def Rescaling(*, stdfun=None, scale=None, loc=None, dim=None):
    """ ``_kernels/_basic.py:231-252`` """
    if stdfun is None:
        stdfun = lambda x: numpy.ones(x.shape)
    return Kernel(lambda x, y: stdfun(x) * stdfun(y), scale=scale, loc=loc, dim=dim)


def Categorical(*, cov=None, scale=None, loc=None, dim=None):
    """ ``_kernels/_basic.py:210-229`` """
    cov = numpy.asarray(cov)
    assert cov.ndim == 2 and cov.shape[0] == cov.shape[1] and numpy.allclose(cov, cov.T)

    def core(x, y):
        assert numpy.issubdtype(x.dtype, numpy.integer)
        return cov[x, y]
    return Kernel(core, scale=scale, loc=loc, dim=dim)


def Gibbs(*, scalefun=None, scale=None, loc=None, dim=None):
    """ ``_kernels/_basic.py:156-185`` """
    if scalefun is None:
        scalefun = lambda x: numpy.ones(x.shape)

    def core(x, y):
        sx, sy = scalefun(x), scalefun(y)
        assert numpy.all(sx > 0) and numpy.all(sy > 0)
        denom = sx ** 2 + sy ** 2
        factor = numpy.sqrt(2 * sx * sy / denom)
        distsq = sum_recurse_dtype(lambda a, b: (a - b) ** 2, x, y)
        return factor * numpy.exp(-distsq / denom)
    return Kernel(core, scale=scale, loc=loc, dim=dim)


def NNKernel(*, sigma0=1, scale=None, loc=None, dim=None):
    """ ``_kernels/_basic.py:122-154`` """
    assert 0 < sigma0 < numpy.inf
    q = sigma0 ** 2
    dot = lambda x, y: sum_recurse_dtype(lambda a, b: a * b, x, y)

    def core(x, y):
        denom = (1 + 2 * (q + dot(x, x))) * (1 + 2 * (q + dot(y, y)))
        return 2 / numpy.pi * numpy.arcsin(2 * (q + dot(x, y)) / denom)
    # derivable any number of times (_basic.py:122): a function of the three dot products x.y, x.x, y.y over the fields (symdiff.py)
    core.symbolic = ('dots', lambda xy, xx, yy: 2 / _sp().pi * _sp().asin(2 * (_sp().Float(q, 30) + xy) / ((1 + 2 * (_sp().Float(q, 30) + xx)) * (1 + 2 * (_sp().Float(q, 30) + yy)))))
    return Kernel(core, scale=scale, loc=loc, dim=dim)


def Linear(*, scale=None, loc=None, dim=None):
    """ ``_kernels/_basic.py:76-91`` """
    core_ = lambda x, y: sum_recurse_dtype(lambda a, b: a * b, x, y)
    core_.symbolic = ('xy', lambda X, Y, upper: X * Y)            # (derivatives along the variable of one-dimensional points)
    return Kernel(core_, scale=scale, loc=loc, dim=dim)


def _onevariable(core, lo=0., hi=numpy.inf):
    """ ``kernel(maxdim=1)`` with the domain assertions of the reference's cores """
    def newcore(x, y):
        if x.dtype.names is not None:
            if len(_leaf_columns_count(x.dtype)) > 1:
                raise ValueError('kernel called on type with dimensionality > maxdim=1')
            name = x.dtype.names[0]
            x, y = x[name], y[name]
        assert numpy.all(x >= lo) and numpy.all(x <= hi) and numpy.all(y >= lo) and numpy.all(y <= hi)
        return core(x, y)
    return newcore


def Wiener(*, scale=None, loc=None, dim=None):
    """ ``_kernels/_randomwalk.py:26-41`` """
    return Kernel(_onevariable(numpy.minimum), scale=scale, loc=loc, dim=dim)


def OrnsteinUhlenbeck(*, scale=None, loc=None, dim=None):
    """ ``_kernels/_randomwalk.py:112-132`` """
    return Kernel(_onevariable(lambda x, y: numpy.exp(-numpy.abs(x - y)) - numpy.exp(-(x + y))), scale=scale, loc=loc, dim=dim)


def BrownianBridge(*, scale=None, loc=None, dim=None):
    """ ``_kernels/_randomwalk.py:134-151`` """
    return Kernel(_onevariable(lambda x, y: numpy.minimum(x, y) - x * y, 0., 1.), scale=scale, loc=loc, dim=dim)


def Decaying(*, alpha=1, scale=None, loc=None, dim=None):
    """ ``_kernels/_basic.py:362-379`` """
    assert alpha >= 0
    core_ = _onevariable(lambda x, y: 1 / (x + y + 1) ** alpha)
    core_.symbolic = ('xy', lambda X, Y, upper: (X + Y + 1) ** (-_sp().Float(alpha, 30)))
    return Kernel(core_, scale=scale, loc=loc, dim=dim)


def FracBrownian(*, H=1/2, K=1, scale=None, loc=None, dim=None):
    """ ``_kernels/_randomwalk.py:46-68`` """
    assert 0 < H <= 1 and 0 < K <= 1
    H2 = 2 * H
    core = lambda x, y: 1 / 2 ** K * ((numpy.abs(x) ** H2 + numpy.abs(y) ** H2) ** K - numpy.abs(x - y) ** (H2 * K))
    core_ = _onevariable(core, -numpy.inf, numpy.inf)
    core_.symbolic = ('xy', lambda X, Y, upper: X * Y) if (H, K) == (1, 1) else None      # (x^2 + y^2 - (x - y)^2) / 2
    return Kernel(core_, scale=scale, loc=loc, dim=dim)


def WienerIntegral(*, scale=None, loc=None, dim=None):
    """ ``_kernels/_randomwalk.py:93-110`` """
    def core(x, y):
        a, b = numpy.minimum(x, y), numpy.maximum(x, y)
        return 1/2 * a ** 2 * (b - a / 3)
    core_ = _onevariable(core)
    core_.symbolic = ('xy', lambda X, Y, upper: (X ** 2 * (Y - X / 3) / 2) if upper else (Y ** 2 * (X - Y / 3) / 2))
    return Kernel(core_, scale=scale, loc=loc, dim=dim)


# --------------------------------------------------------------------------
# fasthash (``_jaxext/_fasthash.py:56-102``; original C ``tests/fast-hash/fasthash.c:30-75``)

_M64 = numpy.uint64(0x880355f21e6d1965)
_MIXC = numpy.uint64(0x2127599bf4325c37)


def _mix(h):
    with numpy.errstate(over='ignore'):
        h = h ^ (h >> numpy.uint64(23))
        h = h * _MIXC
        h = h ^ (h >> numpy.uint64(47))
    return h


def fasthash64(buf, seed):
    buf = numpy.ascontiguousarray(buf).view(numpy.uint8).reshape(-1)
    seed = numpy.uint64(seed)
    nfull = buf.size - buf.size % 8
    pos = buf[:nfull].view(numpy.uint64)
    with numpy.errstate(over='ignore'):
        h = seed ^ (numpy.uint64(buf.size) * _M64)
        for v in pos:
            h = h ^ _mix(v)
            h = h * _M64
        pos2 = buf[nfull:]
        v = numpy.uint64(0)
        for i in reversed(range(pos2.size)):
            v = v ^ (numpy.uint64(pos2[i]) << numpy.uint64(8 * i))
        if pos2.size:
            h = h ^ _mix(v)
            h = h * _M64
    return _mix(h)


def fasthash32(buf, seed):
    h = fasthash64(buf, numpy.uint32(seed))
    with numpy.errstate(over='ignore'):
        return numpy.uint32((h - (h >> numpy.uint64(32))) & numpy.uint64(0xffffffff))


# --------------------------------------------------------------------------
# BART (``_kernels/_bart.py``)

def _float_type(*args):
    """ ``_jaxext/__init__.py:140-144`` under JAX's promotion lattice (x64 mode): integer
    arrays never widen a float type (int32 + float32 -> float32, unlike numpy) and Python
    scalars are weakly typed; with no floating input the result is float64 """
    floats = []
    for a in args:
        if isinstance(a, (bool, int, float)) and not isinstance(a, numpy.generic):
            continue                                    # weak Python scalar
        dt = numpy.asarray(a).dtype
        if numpy.issubdtype(dt, numpy.floating):
            floats.append(dt)
    if not floats:
        return numpy.dtype(numpy.float64)
    return numpy.result_type(*floats)


class BART:
    """ namespace mirroring the reference classmethods; ``BART(...)`` builds the kernel object """

    def __new__(cls, **kw):
        return _make_bart_kernel(**kw)

    # -- input munging ----------------------------------------------------

    @staticmethod
    def _check_x(x):
        """ ``_bart.py:478-483`` """
        x = numpy.asarray(x)
        if x.dtype.names:
            x = recfunctions.structured_to_unstructured(x)
        return x

    @staticmethod
    def _check_splits(splits, indices):
        """ ``_bart.py:485-501`` """
        l, s = splits
        l = numpy.asarray(l)
        assert l.ndim == 1
        if not indices:
            s = numpy.asarray(s)
            assert 1 <= s.ndim <= 2
            if s.ndim == 1:
                s = s[:, None]
            assert l.size == s.shape[1]
            assert numpy.all((0 <= l) & (l <= s.shape[0])), 'length out of bounds'
            assert numpy.all(numpy.sort(s, axis=0) == s), 'unsorted splitting points'
        return l, s

    @classmethod
    def splits_from_coord(cls, x):
        """ ``_bart.py:209-259`` """
        x = cls._check_x(x)
        x = x.reshape(-1, x.shape[-1]) if x.size else x.reshape(1, x.shape[-1])
        if numpy.issubdtype(x.dtype, numpy.inexact):
            fill = numpy.finfo(x.dtype).max
        else:
            fill = numpy.iinfo(x.dtype).max
        lengths = []
        mids = []
        for xi in x.T:
            u = numpy.unique(xi)
            u = numpy.concatenate([u, numpy.full(xi.size - u.size, fill, u.dtype)])   # unique(size=, fill_value=)
            with numpy.errstate(over='ignore'):
                m = numpy.where(u[1:] < fill, (u[1:] + u[:-1]) / 2, fill)
            l = numpy.searchsorted(m, fill)
            lengths.append(l)
            mids.append(m)
        return numpy.array(lengths), numpy.stack(mids, axis=1) if mids else numpy.empty((max(len(x) - 1, 0), 0))

    @classmethod
    def indices_from_coord(cls, x, splits):
        """ ``_bart.py:261-299,503-514`` """
        splits = cls._check_splits(splits, False)
        return cls._indices_from_coord(x, splits)

    @classmethod
    def _indices_from_coord(cls, x, checked_splits):
        x = cls._check_x(x)
        if x.shape[-1] != checked_splits[0].size:
            raise ValueError(f'splitting grid is for {checked_splits[0].size} dimensions, found {x.shape[-1]}')
        A = checked_splits[1]
        out = numpy.empty(x.shape, numpy.int64)
        for j in range(A.shape[1]):
            out[..., j] = numpy.searchsorted(A[:, j], x[..., j], side='left')
        return out

    # -- correlation ------------------------------------------------------

    @classmethod
    def correlation(cls, splitsbefore_or_totalsplits, splitsbetween_or_index1, splitsafter_or_index2, *,
        alpha=0.95, beta=2, gamma=1, maxd=2, debug=False, pnt=None, intercept=True,
        weights=None, reset=None, altinput=False):
        """ ``_bart.py:301-455`` """
        a0 = numpy.asarray(splitsbefore_or_totalsplits)
        a1 = numpy.asarray(splitsbetween_or_index1)
        a2 = numpy.asarray(splitsafter_or_index2)
        assert numpy.issubdtype(a0.dtype, numpy.integer)
        assert numpy.issubdtype(a1.dtype, numpy.integer)
        assert numpy.issubdtype(a2.dtype, numpy.integer)
        assert numpy.all(a0 >= 0), 'splitting counts must be nonnegative'
        if altinput:
            assert numpy.all((0 <= a1) & (a1 <= a0)), 'splitting index must be in [0, n]'
            assert numpy.all((0 <= a2) & (a2 <= a0)), 'splitting index must be in [0, n]'
        else:
            assert numpy.all(a1 >= 0) and numpy.all(a2 >= 0), 'splitting counts must be nonnegative'

        if pnt is None:                                           # :373-384
            assert maxd == int(maxd) and maxd >= 0, maxd
            ft = _float_type(alpha, beta)     # JAX lattice: the int64 depths do not widen float32 parameters
            alpha = numpy.asarray(alpha)
            beta = numpy.asarray(beta)
            assert numpy.all((0 <= alpha) & (alpha <= 1)), 'alpha must be in [0, 1]'
            assert numpy.all(beta >= 0), 'beta must be in [0, inf)'
            d = numpy.arange(maxd + 1)
            pnt = alpha[..., None].astype(ft) / (1 + d).astype(ft) ** beta[..., None].astype(ft)
        else:
            pnt = numpy.asarray(pnt)

        if weights is None:                                       # :387-390
            weights = numpy.ones(a0.shape[-1], pnt.dtype)
        else:
            weights = numpy.asarray(weights)

        if isinstance(gamma, str):                                # :393-401
            if gamma == 'auto':
                assert reset is None and 1 <= pnt.shape[-1] - 1 <= 3
                gamma = cls._gamma(weights.shape[-1], pnt)
            else:
                raise KeyError(gamma)
        elif isinstance(gamma, (bool, int, float)) and not isinstance(gamma, numpy.generic):
            # weakly typed Python scalar (jnp.asarray keeps weak_type): adopts the float type of pnt, weights
            gamma = numpy.asarray(gamma, dtype=_float_type(pnt, weights))
        else:
            gamma = numpy.asarray(gamma)

        assert numpy.all((0 <= gamma) & (gamma <= 1)), 'gamma must be in [0, 1]'
        assert numpy.all((0 <= pnt) & (pnt <= 1)), 'pnt must be in [0, 1]'
        assert numpy.all(weights >= 0), 'weights must be in [0, inf)'

        if not intercept:                                         # :411-412
            pnt = pnt.copy()
            pnt[..., 0] = 1

        if reset is None:                                         # :415-421
            reset = []
        if not hasattr(reset, '__len__'):
            reset = [reset]
        reset = [0] + list(reset) + [pnt.shape[-1] - 1]
        for i, j in zip(reset, reset[1:]):
            assert int(j) == j and i <= j, (i, j)

        brackets_norep = list(zip(reset, reset[1:]))              # :424-431
        brackets = [brackets_norep[0] + (1,)]
        for t, b in brackets_norep[1:]:
            lt, lb, lr = brackets[-1]
            if altinput and not debug and lr * (b - t) == lb - lt and b - t <= 2:
                brackets[-1] = lt, b, lr + 1
            else:
                brackets.append((t, b, 1))

        corr = gamma                                              # :434-455
        for t, b, repeat in reversed(brackets):
            probs = pnt[..., t:b + 1]
            if t > 0:
                probs = probs.copy()
                probs[..., 0] = 1
            if repeat > 1:
                head = probs[..., 0:1]
                one = numpy.ones_like(head)
                pieces = [
                    [head if i == 0 else one, p]
                    for i, p in enumerate(numpy.split(probs[..., 1:], repeat, axis=-1))
                ]
                probs = numpy.concatenate(sum(reversed(pieces), start=[]), axis=-1)
            else:
                repeat = None
            corr = cls._correlation_vectorized(a0, a1, a2, probs, corr, weights, debug, altinput, repeat)
        return corr

    @staticmethod
    def _gamma(p, pnt):
        """ ``_bart.py:458-476`` """
        gamma_0 = 0.611 + 0.021 * numpy.exp(-1.3 * (p - 1))
        gamma_d = -0.0034 + 0.084 * numpy.exp(-2.02 * (p - 1))
        s = 2.03 - 0.69 * numpy.exp(-0.72 * (p - 1))
        t = 4.01 - 1.49 * numpy.exp(-0.77 * (p - 1))
        maxd = pnt.shape[-1] - 1
        floor = numpy.clip(gamma_0 - gamma_d * maxd, 0, 1)
        P0 = pnt[..., 0]
        P1 = numpy.minimum(P0, pnt[..., 1])
        with numpy.errstate(divide='ignore', invalid='ignore'):
            corner = numpy.where(P0, 1 - P0 ** (s - t) * P1 ** t, 1)
        # the fit constants are weakly typed Python floats in the reference: the result keeps pnt's float type
        return (floor * corner).astype(pnt.dtype if numpy.issubdtype(pnt.dtype, numpy.floating) else float)

    @classmethod
    def _correlation_vectorized(cls, a0, a1, a2, pnt, gamma, w, debug, altinput, repeat):
        """ ``_bart.py:808-815``: jnp.vectorize with signature '(p),(p),(p),(d),(),(p)->()' """
        a0, a1, a2 = numpy.broadcast_arrays(a0, a1, a2)
        pnt = numpy.asarray(pnt)
        gamma = numpy.asarray(gamma)
        w = numpy.asarray(w)
        bshape = numpy.broadcast_shapes(a0.shape[:-1], pnt.shape[:-1], gamma.shape, w.shape[:-1])
        flt = _float_type(pnt, gamma, w)
        fast = (altinput and not debug and pnt.ndim == 1 and w.ndim == 1
                and pnt.size // (repeat or 1) <= 3)
        if fast:
            gb = numpy.broadcast_to(gamma, bshape)
            nlead = bshape[0] if len(bshape) >= 2 else 0
            chunk = max(1, (1 << 17) // max(1, int(numpy.prod(bshape[1:], dtype=int)) * max(1, a0.shape[-1]))) if nlead else 0
            if nlead and chunk < nlead:
                # elementwise over the leading axes: evaluating slabs of leading rows bounds the temporaries
                # (N x M x p arrays otherwise) without changing any value
                a0b = numpy.broadcast_to(a0, bshape + a0.shape[-1:])
                a1b = numpy.broadcast_to(a1, bshape + a1.shape[-1:])
                a2b = numpy.broadcast_to(a2, bshape + a2.shape[-1:])
                out = numpy.empty(bshape, flt)
                for r0 in range(0, nlead, chunk):
                    sl = slice(r0, r0 + chunk)
                    out[sl] = cls._correlation_fast(a0b[sl], a1b[sl], a2b[sl], pnt, gb[sl], w, repeat or 1, flt)
                return out
            return cls._correlation_fast(a0, a1, a2, pnt, gb, w, repeat or 1, flt)
        a0b = numpy.broadcast_to(a0, bshape + a0.shape[-1:])
        a1b = numpy.broadcast_to(a1, bshape + a1.shape[-1:])
        a2b = numpy.broadcast_to(a2, bshape + a2.shape[-1:])
        pntb = numpy.broadcast_to(pnt, bshape + pnt.shape[-1:])
        gb = numpy.broadcast_to(gamma, bshape)
        wb = numpy.broadcast_to(w, bshape + w.shape[-1:])
        out = numpy.empty(bshape, flt)
        for idx in numpy.ndindex(*bshape):
            if altinput:
                out[idx] = cls._correlation(a0b[idx], a1b[idx], a2b[idx], pntb[idx], gb[idx], wb[idx], bool(debug), repeat)
            else:
                out[idx] = cls._correlation_old(a0b[idx], a1b[idx], a2b[idx], pntb[idx], gb[idx], wb[idx], bool(debug))
        return out

    # The vectorised fast path restates ``_bart.py:628-757`` (depth <= 3
    # closed forms) with the pair axes kept as leading numpy axes.
    @classmethod
    def _correlation_fast(cls, n, ix, iy, pnt, gamma, w, repeat, flt):
        if n.shape[-1] == 0:
            return numpy.ones(gamma.shape, flt)
        pnt = pnt.astype(flt)
        gamma = gamma.astype(flt)
        w = w.astype(flt)
        n = numpy.where(w, n, 0)
        ix = numpy.where(w, ix, 0)
        iy = numpy.where(w, iy, 0)
        anyn0 = numpy.any(ix != iy, axis=-1)    # reference: 64-bit hash inequality (:674-681)
        one = flt.type(1)
        size = pnt.size // repeat
        rows = pnt.reshape(repeat, -1)

        if size == 1:                                             # :685-691
            for row in rows:
                gamma = numpy.where(anyn0, one - (one - gamma) * row[0], one)
            return gamma.astype(flt)

        with numpy.errstate(divide='ignore', invalid='ignore'):
            Wn = numpy.sum(numpy.where(n, w, 0), axis=-1, dtype=flt)          # :694
            if size == 2:                                         # :697-708
                n0 = numpy.abs(ix - iy)
                wn = numpy.where(n, w / n, 0).astype(flt)
                sum_term = numpy.sum(wn * n0, axis=-1, dtype=flt)
                for row in rows:
                    Q = one - row[1] + gamma * row[1]
                    P0 = row[0]
                    result = one - P0 + Q * (P0 - P0 / Wn * sum_term)
                    gamma = numpy.where(anyn0, result, one)
                return gamma.astype(flt)

            xlty = ix < iy                                        # :711-714
            minxy = numpy.where(xlty, ix, iy)
            maxxy = numpy.where(xlty, iy, ix)
            n0 = maxxy - minxy

            assert size == 3                                      # :716-757
            nminus0 = maxxy
            nplus0 = n - minxy
            nout = n - n0
            inv_Wn = (one / Wn)[..., None]
            inv_Wnmod = one / (Wn[..., None] - numpy.where(n, w, 0))
            inv_Wnminus = numpy.where(nplus0, inv_Wn, inv_Wnmod)
            inv_Wnplus = numpy.where(nminus0, inv_Wn, inv_Wnmod)
            wn = numpy.where(n, w / n, 0).astype(flt)
            S = numpy.sum(wn * nout, axis=-1, dtype=flt)
            t = wn * n0
            terms1 = (S[..., None] + t) * (inv_Wnminus + inv_Wnplus + inv_Wn * (nout - 2))
            terms2 = numpy.where(nplus0, w * inv_Wn * n0 / nplus0, w * inv_Wnmod)
            terms2 = terms2 + numpy.where(nminus0, w * inv_Wn * n0 / nminus0, w * inv_Wnmod)
            # digamma is only ever taken at integers cast to `flt`: a table of scipy's values at those integers,
            # indexed by the integer, gives bit for bit what evaluating the function elementwise does
            psi = _digamma_at_integers(int(numpy.max(n, initial=0)) + 2, flt)
            psin = psi[numpy.where(n, n, 1)]
            psiminus = numpy.where(xlty, psi[1 + iy], psi[1 + ix])
            psiplus = numpy.where(xlty, psi[1 + n - ix], psi[1 + n - iy])
            terms3 = w * inv_Wn * n0 * (2 * psin - psiminus - psiplus)
            terms = terms1 - terms2 - terms3
            sumi = numpy.sum(wn * terms, axis=-1, dtype=flt)
            inv_Wn = inv_Wn[..., 0]
            for row in rows:
                Q = one + row[2] * (gamma - one)
                sump = S + row[1] * (Q * sumi - S)
                result = one + row[0] * (inv_Wn * sump - one)
                gamma = numpy.where(anyn0, result, one)
            return gamma.astype(flt)

    @classmethod
    def _correlation(cls, n, ix, iy, pnt, gamma, w, debug, repeat):
        """ ``_bart.py:628-806`` for one pair (scalar loops for the deep recursion) """
        assert n.ndim == 1
        assert n.shape == ix.shape == iy.shape == w.shape
        assert pnt.ndim == 1 and pnt.size > 0
        gamma = numpy.asarray(gamma)
        assert gamma.ndim == 0
        if repeat is not None:
            assert not debug and repeat > 0 and pnt.size % repeat == 0 and pnt.size // repeat <= 3
        else:
            repeat = 1
        flt = _float_type(pnt, gamma, w)
        if n.size == 0:
            return flt.type(1)
        if not debug and pnt.size // repeat <= 3:
            return cls._correlation_fast(n, ix, iy, pnt, gamma, w, repeat, flt)[()]
        pnt = pnt.astype(flt)
        gamma = gamma.astype(flt)
        w = w.astype(flt)
        n = numpy.where(w, n, 0)
        ix = numpy.where(w, ix, 0)
        iy = numpy.where(w, iy, 0)
        anyn0 = bool(numpy.any(ix != iy))
        if pnt.size == 1:
            return (1 - (1 - gamma) * pnt[0]) if anyn0 else flt.type(1)
        Wn = numpy.sum(numpy.where(n, w, 0))
        minxy = numpy.minimum(ix, iy)
        maxxy = numpy.maximum(ix, iy)
        n0 = maxxy - minxy
        nminus = minxy.copy()                                     # :759-806
        nplus = n - maxxy
        p = len(nminus)
        sump = flt.type(0)
        if anyn0:
            for i in range(p):
                nminusi, n0i, nplusi = nminus[i], n0[i], nplus[i]
                ni = nminusi + n0i + nplusi
                if not ni:
                    continue
                sumn = flt.type(0)
                for k in range(nminusi + nplusi):
                    nm = nminus.copy()
                    npl = nplus.copy()
                    if k < nminusi:
                        nm[i] = k
                    else:
                        npl[i] = k - nminusi
                    nn = nm + n0 + npl
                    sumn += cls._correlation(nn, nm, nm + n0, pnt[1:], gamma, w, debug, None)
                sump += w[i] * sumn / ni
            return 1 - pnt[0] * (1 - sump / Wn)
        return flt.type(1)

    @classmethod
    def _correlation_old(cls, nminus, n0, nplus, pnt, gamma, w, debug):
        """ ``_bart.py:516-610``: the reference's older, independent implementation """
        assert nminus.shape == n0.shape == nplus.shape == w.shape
        assert nminus.ndim == 1 and pnt.ndim == 1 and pnt.size > 0
        nminus = numpy.where(w, nminus, 0)
        n0 = numpy.where(w, n0, 0)
        nplus = numpy.where(w, nplus, 0)
        flt = _float_type(pnt, gamma, w)
        if nminus.size == 0:
            return flt.type(1)
        anyn0 = bool(numpy.any(numpy.logical_and(n0, w)))
        if pnt.size == 1:
            return flt.type(1 - (1 - gamma) * pnt[0]) if anyn0 else flt.type(1)
        nout = nminus + nplus
        n = nout + n0
        Wn = numpy.sum(numpy.where(n, w, 0))
        with numpy.errstate(divide='ignore', invalid='ignore'):
            if pnt.size == 2 and not debug:
                Q = 1 - (1 - gamma) * pnt[1]
                sump = Q * numpy.sum(numpy.where(n, w * nout / n, 0))
                return flt.type(1 - pnt[0] * (1 - sump / Wn)) if anyn0 else flt.type(1)
            if pnt.size == 3 and not debug:
                Q = 1 - (1 - gamma) * pnt[2]
                s = w * nout / n
                S = numpy.sum(numpy.where(n, s, 0))
                t = w * n0 / n
                psin = special.digamma(n.astype(flt))
                def terms(nminus, nplus):
                    nminus0 = nminus + n0
                    Wnmod = Wn - numpy.where(nminus0, 0, w)
                    frac = numpy.where(nminus0, w * nminus / nminus0, 0)
                    terms1 = (S - s + frac) / Wnmod
                    psi1nminus0 = special.digamma((1 + nminus0).astype(flt))
                    terms2 = ((nplus - 1) * (S + t) - w * n0 * (psin - psi1nminus0)) / Wn
                    return numpy.where(nplus, terms1 + terms2, 0)
                tplus = terms(nminus, nplus)
                tminus = terms(nplus, nminus)
                tall = numpy.where(n, w * (tplus + tminus) / n, 0)
                sump = (1 - pnt[1]) * S + pnt[1] * Q * numpy.sum(tall)
                return flt.type(1 - pnt[0] * (1 - sump / Wn)) if anyn0 else flt.type(1)
        p = len(nminus)
        sump = 0.
        if anyn0:
            for i in range(p):
                nminusi, n0i, nplusi = nminus[i], n0[i], nplus[i]
                ni = nminusi + n0i + nplusi
                if not ni:
                    continue
                sumn = 0.
                for k in range(nminusi + nplusi):
                    nm = nminus.copy()
                    npl = nplus.copy()
                    if k < nminusi:
                        nm[i] = k
                    else:
                        npl[i] = k - nminusi
                    sumn += cls._correlation_old(nm, n0, npl, pnt[1:], gamma, w, debug)
                sump += w[i] * sumn / ni
            return flt.type(1 - pnt[0] * (1 - sump / Wn))
        return flt.type(1)


_PSI_CACHE = {}


def _digamma_at_integers(count, flt):
    """ ``special.digamma(numpy.arange(count).astype(flt))`` (index 0 holds the pole, never used), cached """
    key = numpy.dtype(flt).char
    tab = _PSI_CACHE.get(key)
    if tab is None or len(tab) < count:
        with numpy.errstate(divide='ignore', invalid='ignore'):
            tab = special.digamma(numpy.arange(max(count, 4096)).astype(flt))
        _PSI_CACHE[key] = tab
    return tab


def _make_bart_kernel(*, alpha=0.95, beta=2, maxd=2, gamma=1, splits=None, pnt=None,
    intercept=True, weights=None, reset=None, indices=False, scale=None, loc=None, dim=None):
    """ ``_bart.py:32-202`` wrapper: kernel(x, y) -> BART.correlation(splits[0], ix, iy, altinput=True) """
    checked = BART._check_splits(splits, indices)

    def core(x, y):
        if not x.dtype.names:
            x = x[..., None]
        if not y.dtype.names:
            y = y[..., None]
        if indices:
            ix = BART._check_x(x)
            iy = BART._check_x(y)
        else:
            ix = BART._indices_from_coord(x, checked)
            iy = BART._indices_from_coord(y, checked)
        return BART.correlation(
            checked[0], ix, iy,
            pnt=pnt, alpha=alpha, beta=beta, gamma=gamma, maxd=maxd,
            intercept=intercept, weights=weights, reset=reset, altinput=True,
        )

    return Kernel(core, scale=scale, loc=loc, dim=dim)

"""
First-order forward-mode differentiation of the HOST arithmetic that a ``gpfactory`` does with its hyperparameters.

The reference differentiates ``gpfactory`` with jax (``_fit.py:598-754``: ``jax.jacfwd`` / ``jacrev`` of the traced factory).
There is no tracer here, and none is needed for what a factory does with a hyperparameter before it reaches a kernel, a
covariance block or a transformation coefficient: a little scalar / array arithmetic.  `Dual` carries that arithmetic's
first-order tangents, ``value + sum_k tangents[k] eps_k``, through the Python operators and the numpy ufuncs factories use
(``exp``, ``log``, ``sqrt``, ``square``, ``power`` ...).  The objects that consume hyperparameters understand it:

* `Kernel` (``scale=``, ``loc=``, ``kernel * c``, ``kernel + c``) -> `gram_tangents`: the Gram matrix and its tangents as device
  matrices (``lgp_gram_isotropic_dscale`` for the scale);
* `GP.addcov` / ``givencov`` blocks (``hp['s2'] * numpy.eye(n)``, ``lsqfitgp_amd.diag(...)``), `GP.addtransf` coefficients and
  the data themselves (``y - hp['mean']``);

and `GP._prior_decomp_tangents` assembles, besides the matrix to factor, its derivative with respect to every tangent key, which
`Chol.normal_density_derivs` contracts on the device -- what `empbayes_fit` needs for the gradient and the Fisher matrix.  An
operation that is not covered raises `NotImplementedError`; `empbayes_fit` then falls back to central differences.
"""

import numbers

import numpy


def _is_plain(x):
    return isinstance(x, (numbers.Number, numpy.ndarray, numpy.generic, list, tuple))


class Dual:
    """ ``value`` (a float or an array) with its tangents ``{key: array broadcastable to value}`` """

    __array_priority__ = 3000
    __slots__ = ('value', 'tangents')

    def __init__(self, value, tangents=None):
        if isinstance(value, Dual):
            tangents = dict(value.tangents) if tangents is None else tangents
            value = value.value
        self.value = value if isinstance(value, (float, numpy.ndarray)) else (float(value) if numpy.ndim(value) == 0 else numpy.asarray(value))
        self.tangents = {k: t for k, t in (tangents or {}).items() if numpy.any(t != 0)}

    # ---- inspection (what validation code asks of an array or a number)
    @property
    def shape(self):
        return numpy.shape(self.value)

    @property
    def ndim(self):
        return numpy.ndim(self.value)

    @property
    def size(self):
        return numpy.size(self.value)

    @property
    def dtype(self):
        return numpy.asarray(self.value).dtype

    def __array__(self, dtype=None, copy=None):
        a = numpy.asarray(self.value)
        return a if dtype is None else a.astype(dtype)

    def __float__(self):
        return float(self.value)

    def __len__(self):
        return len(self.value)

    def __repr__(self):
        return f'Dual({self.value!r}, tangents={sorted(self.tangents, key=repr)!r})'

    def tangent(self, key):
        """ the tangent along ``key`` broadcast to the shape of the value (zeros when it does not depend on it) """
        t = self.tangents.get(key)
        if t is None:
            return numpy.zeros(self.shape)
        return numpy.broadcast_to(numpy.asarray(t, float), self.shape)

    # ---- shape operations
    def _map(self, f):
        return Dual(f(numpy.asarray(self.value)), {k: f(numpy.broadcast_to(numpy.asarray(t, float), self.shape)) for k, t in self.tangents.items()})

    def reshape(self, *shape):
        return self._map(lambda a: a.reshape(*shape))

    def __getitem__(self, idx):
        return self._map(lambda a: a[idx])

    @property
    def T(self):
        return self._map(lambda a: a.T)

    # ---- arithmetic
    @staticmethod
    def _parts(x):
        if isinstance(x, Dual):
            return x.value, x.tangents
        if _is_plain(x):
            return (x if isinstance(x, (numbers.Number, numpy.ndarray, numpy.generic)) else numpy.asarray(x)), {}
        return None

    @classmethod
    def _binary(cls, a, b, f, dfa, dfb):
        """ f(a, b) with partial derivatives dfa(av, bv, out), dfb(av, bv, out) """
        pa, pb = cls._parts(a), cls._parts(b)
        if pa is None or pb is None:
            return NotImplemented
        (av, at), (bv, bt) = pa, pb
        out = f(av, bv)
        tan = {}
        if at:
            da = dfa(av, bv, out)
            for k, t in at.items():
                tan[k] = da * t
        if bt:
            db = dfb(av, bv, out)
            for k, t in bt.items():
                tan[k] = tan[k] + db * t if k in tan else db * t
        return cls(out, tan)

    def __add__(self, o):
        return self._binary(self, o, lambda a, b: a + b, lambda a, b, r: 1., lambda a, b, r: 1.)

    def __radd__(self, o):
        return self._binary(o, self, lambda a, b: a + b, lambda a, b, r: 1., lambda a, b, r: 1.)

    def __sub__(self, o):
        return self._binary(self, o, lambda a, b: a - b, lambda a, b, r: 1., lambda a, b, r: -1.)

    def __rsub__(self, o):
        return self._binary(o, self, lambda a, b: a - b, lambda a, b, r: 1., lambda a, b, r: -1.)

    def __mul__(self, o):
        return self._binary(self, o, lambda a, b: a * b, lambda a, b, r: b, lambda a, b, r: a)

    def __rmul__(self, o):
        return self._binary(o, self, lambda a, b: a * b, lambda a, b, r: b, lambda a, b, r: a)

    def __truediv__(self, o):
        return self._binary(self, o, lambda a, b: a / b, lambda a, b, r: 1. / b, lambda a, b, r: -r / b)

    def __rtruediv__(self, o):
        return self._binary(o, self, lambda a, b: a / b, lambda a, b, r: 1. / b, lambda a, b, r: -r / b)

    def __pow__(self, o):
        return self._binary(self, o, lambda a, b: a ** b, lambda a, b, r: b * a ** (b - 1), lambda a, b, r: r * numpy.log(a))

    def __rpow__(self, o):
        return self._binary(o, self, lambda a, b: a ** b, lambda a, b, r: b * a ** (b - 1), lambda a, b, r: r * numpy.log(a))

    def __neg__(self):
        return Dual(-self.value, {k: -t for k, t in self.tangents.items()})

    def __pos__(self):
        return self

    def __matmul__(self, o):
        return self._matmul(self, o)

    def __rmatmul__(self, o):
        return self._matmul(o, self)

    @classmethod
    def _matmul(cls, a, b):
        pa, pb = cls._parts(a), cls._parts(b)
        if pa is None or pb is None:
            return NotImplemented
        (av, at), (bv, bt) = pa, pb
        tan = {k: numpy.broadcast_to(numpy.asarray(t, float), numpy.shape(av)) @ bv for k, t in at.items()}
        for k, t in bt.items():
            term = av @ numpy.broadcast_to(numpy.asarray(t, float), numpy.shape(bv))
            tan[k] = tan[k] + term if k in tan else term
        return cls(av @ bv, tan)

    # comparisons look at the value (assertions on ranges, branches on signs)
    def __lt__(self, o):
        return self.value < (o.value if isinstance(o, Dual) else o)

    def __le__(self, o):
        return self.value <= (o.value if isinstance(o, Dual) else o)

    def __gt__(self, o):
        return self.value > (o.value if isinstance(o, Dual) else o)

    def __ge__(self, o):
        return self.value >= (o.value if isinstance(o, Dual) else o)

    def __eq__(self, o):
        return self.value == (o.value if isinstance(o, Dual) else o)

    def __ne__(self, o):
        return self.value != (o.value if isinstance(o, Dual) else o)

    __hash__ = None

    def __bool__(self):
        return bool(self.value)

    # ---- numpy ufuncs: numpy.exp(hp['log_s']), numpy.sqrt(...), arrays on the left of an operator
    _UNARY = {
        numpy.exp: lambda v, r: r,
        numpy.expm1: lambda v, r: r + 1.,
        numpy.log: lambda v, r: 1. / v,
        numpy.log1p: lambda v, r: 1. / (1. + v),
        numpy.log2: lambda v, r: 1. / (v * numpy.log(2.)),
        numpy.log10: lambda v, r: 1. / (v * numpy.log(10.)),
        numpy.sqrt: lambda v, r: 0.5 / r,
        numpy.square: lambda v, r: 2. * v,
        numpy.negative: lambda v, r: -1.,
        numpy.positive: lambda v, r: 1.,
        numpy.reciprocal: lambda v, r: -r * r,
        numpy.tanh: lambda v, r: 1. - r * r,
        numpy.sin: lambda v, r: numpy.cos(v),
        numpy.cos: lambda v, r: -numpy.sin(v),
        numpy.arctan: lambda v, r: 1. / (1. + v * v),
        numpy.abs: lambda v, r: numpy.sign(v),
    }

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        if method != '__call__' or kw:
            return NotImplemented
        if ufunc in self._UNARY and len(inputs) == 1:
            v = self.value
            r = ufunc(v)
            d = self._UNARY[ufunc](v, r)
            return Dual(r, {k: d * t for k, t in self.tangents.items()})
        a, b = inputs if len(inputs) == 2 else (None, None)
        ops = {numpy.add: '__add__', numpy.subtract: '__sub__', numpy.multiply: '__mul__', numpy.true_divide: '__truediv__',
               numpy.power: '__pow__', numpy.matmul: '__matmul__'}
        if ufunc in ops and len(inputs) == 2:
            if isinstance(a, Dual):
                return getattr(a, ops[ufunc])(b)
            return getattr(b, '__r' + ops[ufunc][2:])(a)
        cmp = {numpy.less: '__lt__', numpy.less_equal: '__le__', numpy.greater: '__gt__', numpy.greater_equal: '__ge__',
               numpy.equal: '__eq__', numpy.not_equal: '__ne__'}
        if ufunc in cmp and len(inputs) == 2:
            av = a.value if isinstance(a, Dual) else a
            bv = b.value if isinstance(b, Dual) else b
            return ufunc(av, bv)
        if ufunc in (numpy.isfinite, numpy.isnan, numpy.isinf) and len(inputs) == 1:
            return ufunc(self.value)
        raise NotImplementedError(f'numpy.{ufunc.__name__} of a hyperparameter with tangents')


def split(x):
    """ ``(plain value, tangents)`` of a Dual, ``(x, {})`` of anything else """
    if isinstance(x, Dual):
        return x.value, x.tangents
    return x, {}


def keys_of(*objs):
    out = []
    for o in objs:
        if isinstance(o, Dual):
            for k in o.tangents:
                if k not in out:
                    out.append(k)
    return out

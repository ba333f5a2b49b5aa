"""
Host-side kernel objects for the three kernels of the hot path, mirroring the
call contract of ``lsqfitgp.CrossKernel`` (``src/lsqfitgp/_Kernel/_crosskernel.py:152-200``):

* ``kernel(x, y)`` -- broadcasting elementwise evaluation on host arrays
  (plain or structured), result shape ``broadcast(x, y).shape``;
* constructor arguments ``scale`` and ``loc`` (``_Kernel/_ops.py:292-326``):
  inputs are transformed as ``(x - loc) / scale``;
* ``kernel * c`` and ``kernel + c`` with scalar ``c`` (``_Kernel/_alg.py:32-82``);
* plus the fast path the GP uses: ``kernel.gram(xs, ys)`` -> ``DeviceArray``
  (the ``kernel(x[:, None], y[None, :])`` of ``_GP/_elements.py:575-577``).

All evaluation happens in ``csrc/gram.hip``.
"""

import collections
import ctypes
import numbers
import re

import numpy
from numpy.lib import recfunctions
from scipy import special as _special   # digamma table for BART (scipy is a dependency of the reference too)

from . import _lib
from . import _dual
from . import _zeta
from ._device import Context, DeviceArray


def _leaf_columns(x):
    """
    Flat (1-D) host array -> list of 1-D leaf columns in field order
    (``_Kernel/_util.py:74-99`` recursion order; a sub-array field contributes
    its elements consecutively).  Columnar like ``StructuredArray``
    (``src/lsqfitgp/_array.py:104-105``).
    """
    x = numpy.asarray(x)
    assert x.ndim == 1
    if x.dtype.names is None:
        return [x]
    cols = []
    for name in x.dtype.names:
        f = x[name]
        if f.dtype.names is not None:
            if f.ndim > 1:
                for idx in numpy.ndindex(*f.shape[1:]):
                    cols.extend(_leaf_columns(numpy.ascontiguousarray(f[(slice(None),) + idx])))
            else:
                cols.extend(_leaf_columns(f))
        elif f.ndim > 1:
            f2 = f.reshape(len(f), -1)
            cols.extend(f2[:, k] for k in range(f2.shape[1]))
        else:
            cols.append(f)
    return cols


def _float_dtype_of(cols, *scalars):
    """ float32 only if every input is float32 (x64 mode, ``_patch_jax.py:27``) """
    dts = [c.dtype for c in cols] + [numpy.asarray(s).dtype for s in scalars if s is not None and isinstance(s, numpy.generic)]
    if dts and all(d == numpy.float32 for d in dts):
        return numpy.dtype(numpy.float32)
    return numpy.dtype(numpy.float64)


def _ops_arrays(outops, F):
    n = len(outops)
    codes = (ctypes.c_int * max(n, 1))(*[_lib.LGP_OP_MUL if op == 'mul' else _lib.LGP_OP_ADD for op, _ in outops])
    vals = (F * max(n, 1))(*[float(c) for _, c in outops])
    return n, codes, vals


def _is_outer(xshape, yshape):
    """ x of shape (..., n, 1) and y (..., 1, m) with only those axes non trivial? """
    nd = max(len(xshape), len(yshape))
    xs = (1,) * (nd - len(xshape)) + tuple(xshape)
    ys = (1,) * (nd - len(yshape)) + tuple(yshape)
    xa = [i for i, s in enumerate(xs) if s != 1]
    ya = [i for i, s in enumerate(ys) if s != 1]
    if len(xa) > 1 or len(ya) > 1:
        return None
    if xa and ya and xa[0] == ya[0]:
        return None
    return (xa[0] if xa else None), (ya[0] if ya else None), nd


class DevicePoints:
    """ a point set already resident on the device in the layout a kernel's Gram entry point reads (``kernel.prepare``):
    columnar coordinates (d x n) for the isotropic kernels, bin indices (n x p) for BART.  ``sub(i0, i1)`` is a view of
    points i0..i1.  Lets callers that evaluate many blocks over the same points (the sharded Gram build) upload them once. """

    def __init__(self, kind, array, dtype, ctx):
        self.kind, self.array, self.dtype, self.ctx = kind, array, dtype, ctx

    def __len__(self):
        return self.array.shape[1] if self.kind == 'coords' else self.array.shape[0]

    def sub(self, i0, i1):
        if self.kind == 'coords':
            return DevicePoints(self.kind, self.array.block(0, self.array.shape[0], i0, i1), self.dtype, self.ctx)
        return DevicePoints(self.kind, self.array.block(i0, i1, 0, self.array.shape[1]), self.dtype, self.ctx)


class _PointBundle:
    """ what `prepare` of a composite kernel returns: the prepared points of each operand (uploaded once per operand), sliced
    together; for kernels evaluated from host quantities, the host array itself """

    def __init__(self, parts, n, dtype):
        self.parts, self._n, self.dtype = parts, n, numpy.dtype(dtype)

    def __len__(self):
        return self._n

    def sub(self, i0, i1):
        return _PointBundle([p.sub(i0, i1) if hasattr(p, 'sub') else p[i0:i1] for p in self.parts], i1 - i0, self.dtype)


class Kernel:
    """ base class: input transformation, output algops, call contract """

    def __init__(self, *, scale=None, loc=None, dim=None, outops=()):
        # hyperparameters may carry first-order tangents (`_dual.Dual`, what `empbayes_fit` hands a gpfactory): the kernel keeps
        # the plain values for everything it already does and the tangents for `gram_tangents`
        scale, self._dscale = _dual.split(scale)
        loc, self._dloc = _dual.split(loc)
        if scale is not None:
            if not (0 < scale < numpy.inf):                       # _ops.py:324-325
                raise AssertionError(scale)
        if loc is not None:
            if not (-numpy.inf < loc < numpy.inf):                # _ops.py:306-307
                raise AssertionError(loc)
        if dim is not None and not isinstance(dim, (str, list)):
            raise TypeError(f'dim must be a (list of) string, found {dim!r}')           # _Kernel/_ops.py:250-251
        self._scale = scale
        self._loc = loc
        self._dim = dim
        self._outops = tuple(outops)
        self._doutops = tuple({} for _ in self._outops)          # tangents of the constants of `_outops`, entry by entry

    def _select(self, x):
        """ ``dim=``: the kernel acts on one field (or a list of fields) of structured points, T(f)(x) = f(x[dim])
        (``_Kernel/_ops.py:233-260``); applied before everything else, so ``maxdim`` counts the selected fields """
        dim = getattr(self, '_dim', None)
        if dim is None or isinstance(x, (DevicePoints, _PointBundle)):
            return x
        x = numpy.asarray(x)
        if x.dtype.names is None:
            raise ValueError(f'cannot get dim={dim!r} from non-structured input')
        if isinstance(dim, str) and x.dtype[dim].shape:
            return x[[dim]]
        return x[dim]

    def _copy(self, **kw):
        new = self.__class__.__new__(self.__class__)
        new.__dict__.update(self.__dict__)
        for memo in ('_index_memo', '_psi_memo'):                 # device-side memos of the original (BART) are its own
            new.__dict__.pop(memo, None)
        new.__dict__.update(kw)
        return new

    # -- algops (``_Kernel/_alg.py:32-82``), scalars only
    def _algop(self, op, c):
        if isinstance(c, Kernel):
            return _Composite(op, self, c)                          # k1 + k2, k1 * k2 (_Kernel/_alg.py:32-82)
        c, dc = _dual.split(c)
        if not (isinstance(c, numbers.Number) or (isinstance(c, numpy.ndarray) and c.ndim == 0) or isinstance(c, numpy.generic)):
            return NotImplemented
        if len(self._outops) >= 8:
            raise NotImplementedError('at most 8 chained scalar operations')
        return self._copy(_outops=self._outops + ((op, c),), _doutops=getattr(self, '_doutops', ()) + (dict(dc),))

    def __mul__(self, c):
        return self._algop('mul', c)

    __rmul__ = __mul__

    def __add__(self, c):
        return self._algop('add', c)

    __radd__ = __add__

    # -- evaluation
    def __call__(self, x, y):
        """ ``_crosskernel.py:192-200`` """
        x = numpy.asarray(x)
        y = numpy.asarray(y)
        shape = numpy.broadcast_shapes(x.shape, y.shape)
        outer = _is_outer(x.shape, y.shape)
        if outer is not None and x.size and y.size:
            res = self.gram(x.reshape(-1), y.reshape(-1)).to_host()     # (n, m)
            xa, ya, nd = outer
            if xa is not None and ya is not None and xa > ya:
                res = res.T
            return res.reshape(shape)
        if 0 in shape:
            return numpy.empty(shape, self._result_dtype(x, y))
        xb = numpy.broadcast_to(x, shape).reshape(-1)
        yb = numpy.broadcast_to(y, shape).reshape(-1)
        return self.pairs(xb, yb).to_host().reshape(shape)

    def gram(self, xs, ys, *, ctx=None):  # pragma: no cover
        raise NotImplementedError

    def pairs(self, xs, ys, *, ctx=None):  # pragma: no cover
        raise NotImplementedError

    # -- first-order tangents of the hyperparameters (`_dual`)
    def tangent_keys(self):
        """ the keys the kernel's hyperparameters carry tangents for """
        keys = []
        for d in (getattr(self, '_dscale', {}), getattr(self, '_dloc', {})) + tuple(getattr(self, '_doutops', ())):
            for k in d:
                if k not in keys:
                    keys.append(k)
        return keys

    def gram_tangents(self, xs, ys, *, ctx=None, **kw):
        """ ``(K, {key: dK/dkey})`` as device matrices """
        if self.tangent_keys():
            raise NotImplementedError(f'tangents of the hyperparameters of {type(self).__name__}')
        return self.gram(xs, ys, ctx=ctx, **kw), {}


def _dev_axpby(alpha, A, beta, B, out=None):
    ref = A if A is not None else B
    if out is None:
        out = DeviceArray.empty(ref.shape, ref.dtype, ref.ctx)
    r2 = ref.as2d()
    f, F = _lib.fn('lgp_axpby', ref.dtype)
    _lib.check(f(ref.ctx.handle, r2.shape[0], r2.shape[1], F(alpha), None if A is None else A.ptr, 0 if A is None else A.as2d().ld,
                 F(beta), None if B is None else B.ptr, 0 if B is None else B.as2d().ld, out.ptr, out.as2d().ld), 'lgp_axpby')
    return out


def _dev_mul(alpha, A, B, beta=0., C0=None, out=None):
    """ out = alpha A o B + beta C0 """
    if out is None:
        out = DeviceArray.empty(A.shape, A.dtype, A.ctx)
    a2 = A.as2d()
    f, F = _lib.fn('lgp_mul', A.dtype)
    _lib.check(f(A.ctx.handle, a2.shape[0], a2.shape[1], F(alpha), A.ptr, a2.ld, B.ptr, B.as2d().ld, F(beta),
                 None if C0 is None else C0.ptr, 0 if C0 is None else C0.as2d().ld, out.ptr, out.as2d().ld), 'lgp_mul')
    return out


def _dev_astype(A, dtype):
    dtype = numpy.dtype(dtype)
    if A.dtype == dtype:
        return A
    out = DeviceArray.empty(A.shape, dtype, A.ctx)
    if A.size:
        a2, o2 = A.as2d(), out.as2d()
        name = 'lgp_convert_f32_f64' if dtype == numpy.float64 else 'lgp_convert_f64_f32'
        _lib.check(getattr(_lib.load(), name)(A.ctx.handle, a2.shape[0], a2.shape[1], o2.ptr, o2.ld, a2.ptr, a2.ld), name)
    return out


class _Composite(Kernel):
    """
    Sum or product of two kernels (``_Kernel/_alg.py:32-82``: ``k1 + k2``, ``k1 * k2``): each operand is evaluated by its own
    Gram kernel on the same points and the matrices are combined entry by entry on the device (`lgp_axpby` / `lgp_mul`); scalar
    operations applied to the composite afterwards follow.  The hyperparameter tangents (`_dual`) combine by linearity and
    by the product rule.  Derivative observations: of a sum the terms' derivatives add, of a product Leibniz' rule (`_gram_derived`).
    """

    def __init__(self, op, a, b):
        super().__init__()
        self._op, self._a, self._b = op, a, b

    def _result_dtype(self, x, y):
        return numpy.result_type(self._a._result_dtype(x, y), self._b._result_dtype(x, y))

    def tangent_keys(self):
        keys = super().tangent_keys()
        for k in self._a.tangent_keys() + self._b.tangent_keys():
            if k not in keys:
                keys.append(k)
        return keys

    def _finish(self, M):
        """ the scalar operations applied to the composite, in place """
        for op, c in self._outops:
            if op == 'mul':
                _dev_axpby(float(c), M, 0., None, M)
            else:
                m2 = M.as2d()
                f, F = _lib.fn('lgp_add_scalar', M.dtype)
                _lib.check(f(M.ctx.handle, m2.shape[0], m2.shape[1], M.ptr, m2.ld, F(float(c))), 'lgp_add_scalar')
        return M

    def _combine(self, A, B, out=None):
        dt = numpy.result_type(A.dtype, B.dtype)
        A, B = _dev_astype(A, dt), _dev_astype(B, dt)
        if out is not None and out.dtype != dt:
            raise TypeError(f'output of dtype {out.dtype} for a kernel of dtype {dt}')
        M = _dev_axpby(1., A, 1., B, out) if self._op == 'add' else _dev_mul(1., A, B, out=out)
        return self._finish(M)

    def prepare(self, xs, *, ctx=None, dtype=None):
        """ the operands' prepared points, side by side (`_PointBundle`) """
        parts = [self._a.prepare(xs, ctx=ctx), self._b.prepare(xs, ctx=ctx)]
        return _PointBundle(parts, len(parts[0]), dtype or numpy.result_type(*[p.dtype for p in parts]))

    @staticmethod
    def _part(pts, i):
        return pts.parts[i] if isinstance(pts, _PointBundle) else pts

    def gram(self, xs, ys, *, ctx=None, out=None, deriv=(0, 0)):
        if isinstance(xs, _PointBundle) != isinstance(ys, _PointBundle):
            raise TypeError('prepared and plain points mixed')
        if any(numpy.any(numpy.asarray(o) != 0) for o in deriv):
            return self._gram_derived(xs, ys, ctx, out, deriv)
        return self._combine(self._a.gram(self._part(xs, 0), self._part(ys, 0), ctx=ctx),
                             self._b.gram(self._part(xs, 1), self._part(ys, 1), ctx=ctx), out)

    def _derived_gram_tangents(self, xs, ys, ctx, deriv):
        """ a derived block of a composite and its tangents: sums add; products by Leibniz over both sides, each term
        (dA) o B + A o (dB); the composite's own scalar factors (which may carry tangents themselves) last """
        ga = lambda d: self._a.gram_tangents(self._part(xs, 0), self._part(ys, 0), ctx=ctx, deriv=d)
        gb = lambda d: self._b.gram_tangents(self._part(xs, 1), self._part(ys, 1), ctx=ctx, deriv=d)
        K0, tans = None, {}

        def acc(target, w, X, Y):          # target += w X o Y  (Y None: w X)
            term = _dev_mul(w, X, Y) if Y is not None else _dev_axpby(w, X, 0., None)
            return term if target is None else _dev_axpby(1., target, 1., term, target)
        if self._op == 'add':
            for K, dK in (ga(deriv), gb(deriv)):
                K0 = acc(K0, 1., K, None)
                for k, T in dK.items():
                    tans[k] = acc(tans.get(k), 1., T, None)
        else:
            for wx, ax, bx in self._splits(deriv[0]):
                for wy, ay, by in self._splits(deriv[1]):
                    (A, dA), (B, dB) = ga((ax, ay)), gb((bx, by))
                    w = float(wx * wy)
                    K0 = acc(K0, w, A, B)
                    for k, T in dA.items():
                        tans[k] = acc(tans.get(k), w, T, B)
                    for k, T in dB.items():
                        tans[k] = acc(tans.get(k), w, A, T)
        a = _dual.Dual(1.)
        for (op, c), dc in zip(self._outops, self._doutops):
            if op == 'mul':
                a = a * _dual.Dual(float(c), dc)
        K = _dev_axpby(a.value, K0, 0., None)
        out = {}
        for k in set(tans) | set(a.tangents):
            da = float(a.tangents.get(k, 0.))
            out[k] = _dev_axpby(a.value, tans[k], da, K0) if k in tans else _dev_axpby(da, K0, 0., None)
        return K, out

    @staticmethod
    def _splits(spec):
        """ the ways of sharing the derivative orders of one side between two factors: ``[(binomial weight, spec of a, spec of b)]`` """
        import itertools
        import math
        if isinstance(spec, (int, numpy.integer)):
            return [(math.comb(int(spec), k), k, int(spec) - k) for k in range(int(spec) + 1)]
        orders = tuple(int(v) for v in spec)
        res = []
        for part in itertools.product(*(range(o + 1) for o in orders)):
            res.append((math.prod(math.comb(o, k) for o, k in zip(orders, part)), tuple(part), tuple(o - k for o, k in zip(orders, part))))
        return res

    def _gram_derived(self, xs, ys, ctx, out, deriv):
        """ derivative observations of a composite (``_Kernel/_ops.py:115-211`` differentiates whatever the algebra of
        ``_Kernel/_alg.py:32-82`` built): of a sum the terms' derivatives add; of a product Leibniz' rule over the orders of
        both sides, every term the product of two derived blocks of the factors; constants added to the composite drop out, its
        scalar factors multiply the result """
        ga = lambda d: self._a.gram(self._part(xs, 0), self._part(ys, 0), ctx=ctx, deriv=d)
        gb = lambda d: self._b.gram(self._part(xs, 1), self._part(ys, 1), ctx=ctx, deriv=d)
        if self._op == 'add':
            A, B = ga(deriv), gb(deriv)
            dt = numpy.result_type(A.dtype, B.dtype)
            M = _dev_axpby(1., _dev_astype(A, dt), 1., _dev_astype(B, dt), out)
        else:
            M = None
            for wx, ax, bx in self._splits(deriv[0]):
                for wy, ay, by in self._splits(deriv[1]):
                    A, B = ga((ax, ay)), gb((bx, by))
                    dt = numpy.result_type(A.dtype, B.dtype)
                    M = _dev_mul(float(wx * wy), _dev_astype(A, dt), _dev_astype(B, dt), *((0., None) if M is None else (1., M)), out=M)
            if out is not None:
                out.assign(M)
                M = out
        for op, c in self._outops:                     # (added constants: their derivative is zero)
            if op == 'mul':
                _dev_axpby(float(c), M, 0., None, M)
        return M

    def pairs(self, xs, ys, *, ctx=None):
        return self._combine(self._a.pairs(xs, ys, ctx=ctx), self._b.pairs(xs, ys, ctx=ctx))

    def gram_tangents(self, xs, ys, *, ctx=None, **kw):
        if kw.get('deriv') and any(numpy.any(numpy.asarray(o) != 0) for o in kw['deriv']):
            return self._derived_gram_tangents(xs, ys, ctx, kw['deriv'])
        A, dA = self._a.gram_tangents(xs, ys, ctx=ctx)
        B, dB = self._b.gram_tangents(xs, ys, ctx=ctx)
        dt = numpy.result_type(A.dtype, B.dtype)
        A, B = _dev_astype(A, dt), _dev_astype(B, dt)
        K0 = _dev_axpby(1., A, 1., B) if self._op == 'add' else _dev_mul(1., A, B)
        # the scalar operations after the composition: K = a K0 + b with Dual a, b
        a, b = _dual.Dual(1.), _dual.Dual(0.)
        for (op, c), dc in zip(self._outops, self._doutops):
            cd = _dual.Dual(float(c), dc)
            if op == 'mul':
                a, b = a * cd, b * cd
            else:
                b = b + cd
        tangents = {}
        for k in self.tangent_keys():
            ta, tb = dA.get(k), dB.get(k)
            T = None
            if self._op == 'add':
                if ta is not None and tb is not None:
                    T = _dev_axpby(1., _dev_astype(ta, dt), 1., _dev_astype(tb, dt))
                elif ta is not None or tb is not None:
                    T = _dev_axpby(1., _dev_astype(ta if ta is not None else tb, dt), 0., None)
            else:                                              # d(A o B) = dA o B + A o dB
                if ta is not None:
                    T = _dev_mul(1., _dev_astype(ta, dt), B)
                if tb is not None:
                    T = _dev_mul(1., A, _dev_astype(tb, dt), 1. if T is not None else 0., T, T)
            da, db = float(a.tangents.get(k, 0.)), float(b.tangents.get(k, 0.))
            if T is None:
                if da == 0. and db == 0.:
                    continue
                T = _dev_axpby(da, K0, 0., None)
            else:
                _dev_axpby(float(a.value), T, da, K0 if da != 0. else None, T)
            if db != 0.:
                t2 = T.as2d()
                f, F = _lib.fn('lgp_add_scalar', T.dtype)
                _lib.check(f(T.ctx.handle, t2.shape[0], t2.shape[1], T.ptr, t2.ld, F(db)), 'lgp_add_scalar')
            tangents[k] = T
        return self._finish(K0), tangents


# kernels whose Gram build the library fuses with the regularisation and the factorisation (lgp_gram_chol_factor_rhs, csrc/gram.hip)
FUSED_KINDS = (_lib.LGP_KERNEL_EXPQUAD, _lib.LGP_KERNEL_MATERNP, _lib.LGP_KERNEL_GAMMAEXP, _lib.LGP_KERNEL_CAUCHY)


class _Isotropic(Kernel):
    """ ``_Kernel/_isotropic.py:61-81`` with input='squared' """

    _kind = None
    _p = 0
    _par = ()          # real parameters handed to the device core (host-evaluated Python floats, like the reference)

    def _par_args(self, F):
        """ (npar, hpar) of the C-ABI """
        if not self._par:
            return 0, None
        return len(self._par), (F * len(self._par))(*self._par)

    def _result_dtype(self, x, y):
        x, y = self._select(x), self._select(y)
        return _float_dtype_of(_leaf_columns(x.reshape(-1)[:0]) + _leaf_columns(y.reshape(-1)[:0]), self._scale, self._loc)

    def prepare(self, xs, *, ctx=None, dtype=None):
        """ upload a flat point set once: `DevicePoints` accepted by `gram` in place of the host array """
        ctx = ctx or Context.default()
        xc = _leaf_columns(numpy.asarray(self._select(xs)))
        dtype = dtype or _float_dtype_of(xc, self._scale, self._loc)
        return DevicePoints('coords', DeviceArray.from_host(numpy.stack([numpy.asarray(c, dtype) for c in xc]), ctx=ctx), numpy.dtype(dtype), ctx)

    def _prepare(self, xs, ys, ctx):
        if isinstance(xs, DevicePoints) or isinstance(ys, DevicePoints):
            if not (isinstance(xs, DevicePoints) and isinstance(ys, DevicePoints)):
                dt = (xs if isinstance(xs, DevicePoints) else ys).dtype
                xs = xs if isinstance(xs, DevicePoints) else self.prepare(xs, ctx=ctx, dtype=dt)
                ys = ys if isinstance(ys, DevicePoints) else self.prepare(ys, ctx=ctx, dtype=dt)
            if xs.dtype != ys.dtype or xs.array.shape[0] != ys.array.shape[0]:
                raise ValueError('device point sets of different dtype or field count')
            return xs.array, ys.array, xs.dtype, xs.ctx, xs.array.shape[0]
        xc = _leaf_columns(numpy.asarray(self._select(xs)))
        yc = _leaf_columns(numpy.asarray(self._select(ys)))
        if len(xc) != len(yc):
            raise ValueError(f'x has {len(xc)} leaf fields, y has {len(yc)}')
        dtype = _float_dtype_of(xc + yc, self._scale, self._loc)
        ctx = ctx or Context.default()
        xd = DeviceArray.from_host(numpy.stack([numpy.asarray(c, dtype) for c in xc]), ctx=ctx)   # SoA d x n
        same = xs is ys
        yd = xd if same else DeviceArray.from_host(numpy.stack([numpy.asarray(c, dtype) for c in yc]), ctx=ctx)
        return xd, yd, dtype, ctx, len(xc)

    @staticmethod
    def _orders(spec, d):
        """ a derivative specification of one side -> orders along the d leaf fields: an integer (the implicit variable of
        one-dimensional points) or a sequence of d integers """
        if isinstance(spec, (int, numpy.integer)):
            if spec and d != 1:
                raise NotImplementedError('an integer derivative order refers to the one variable of one-dimensional points; name the fields')
            orders = (int(spec),) + (0,) * (d - 1)
        else:
            orders = tuple(int(v) for v in spec)
            if len(orders) != d:
                raise ValueError(f'{len(orders)} derivative orders for {d} fields')
        if any(o < 0 for o in orders):
            raise ValueError(f'negative derivative order {spec!r}')
        return orders

    # ---- derivative observations (``_Kernel/_ops.py:115-211``) ------------------------------------------------------------
    # how many times the PROCESS is derivable (the reference's ``derivable=`` of the kernel decorators: True = any order);
    # None: no derivative core on the device yet
    _derivable = 0

    def _derivability(self):
        return self._derivable

    def _has_closed_form_diff(self, d):
        """ ExpQuad (Hermite polynomials, any fields) and Maternp on one-dimensional points keep their closed forms (`_diff_call`);
        everything else goes through the Taylor arithmetic of csrc/gram_jet.hip """
        return self._kind == _lib.LGP_KERNEL_EXPQUAD or (self._kind == _lib.LGP_KERNEL_MATERNP and d == 1)

    def _jet_pars(self):
        """ (p, real parameters, coefficient table) of the derivative core """
        return int(self._p), tuple(self._par), ()

    def _gram_jet(self, xd, yd, dtype, ctx, d, da, db, out):
        """ d^a/dx^a d^b/dy^b k along ONE field per side, a, b <= 2 (``lgp_gram_deriv``) """
        deg = self._derivability()
        name = type(self).__name__
        if deg is None:
            raise NotImplementedError(f'derivatives of {name}: no derivative core on the device (ExpQuad, Maternp, Matern, Bessel, GammaExp(gamma=2), Cauchy(alpha=2), CausalExpQuad(alpha=0), Taylor, Color, Pink, NNKernel, Zeta, '
                                      'Constant, Linear, Periodic, Cos, Sinc, Celerite, Harmonic, Wendland, Circular, Decaying, WienerIntegral, '
                                      'FracBrownian(H=1, K=1), StationaryFracBrownian(H=1) have one)')
        sides = []
        for orders in (da, db):
            nz = [(f, o) for f, o in enumerate(orders) if o]
            if len(nz) > 1:
                raise NotImplementedError(f'derivatives of {name} along more than one field of a point')
            f, o = nz[0] if nz else (-1, 0)
            if o > deg:
                # the reference's error (``_Kernel/_ops.py:85-90``)
                raise ValueError(f'Took {o} derivatives > limit {deg} on argument of a kernel ({name}).')
            if o > 2:
                raise NotImplementedError(f'derivatives of {name} above the second on one side')
            sides.append((f, o))
        factor = 1.
        for op, c in self._outops:
            if op == 'mul':
                factor = factor * c
        p, par, coef = self._jet_pars()
        f, F = _lib.fn('lgp_gram_deriv', dtype)
        hpar = (F * len(par))(*par) if par else None
        hcoef = (F * len(coef))(*coef) if len(coef) else None
        n, m = xd.shape[1], yd.shape[1]
        _lib.check(f(ctx.handle, self._kind, p, len(par), hpar, len(coef), hcoef, d, xd.ptr, xd.ld, n, yd.ptr, yd.ld, m,
                     int(self._loc is not None), F(self._loc or 0), int(self._scale is not None), F(self._scale or 1), F(factor),
                     sides[0][0], sides[0][1], sides[1][0], sides[1][1], out.ptr, out.ld), 'lgp_gram_deriv')
        return out

    def _diff_call(self, deriv, d):
        """
        ``(kernel id, p, real parameters, output operations)`` of the derived kernel ``kernel.linop('diff', dx, dy)``
        (``_Kernel/_ops.py:115-211``): the process derived ``dx`` times at the first argument and ``dy`` times at the second,
        each an order per leaf field (an integer: the implicit variable of one-dimensional points).  In closed form for

        * `ExpQuad`, any fields: with u = (x - y)/scale field by field, along field f
          d^a/dx^a d^b/dy^b exp(-|u|²/2) = (-1)^a / scale^(a+b) He_{a+b}(u_f) exp(-|u|²/2)  (total order at most 4 per field);
        * `Maternp` (p times derivable on each side, ``_kernels/_matern.py:26-49``), one-dimensional points: K(u) = exp(-t) P(t),
          t = c |u|, c = sqrt(2p+1), has K^(q)(u) = (c sign u)^q exp(-t) Q_q(t) with Q_0 = P, Q_{q+1} = Q_q' - Q_q, so
          d^a/dx^a d^b/dy^b k = (-1)^b (c / scale)^(a+b) sign(u)^(a+b) exp(-t) Q_{a+b}(t).

        Scalar factors of the kernel multiply the derivative, added constants drop out of it.
        """
        da, db = self._orders(deriv[0], d), self._orders(deriv[1], d)
        if not any(da) and not any(db):
            return self._kind, int(self._p), self._par, self._outops
        scale = float(self._scale if self._scale is not None else 1.)
        factor = 1.
        for op, c in self._outops:
            if op == 'mul':
                factor = factor * c
        if self._kind == _lib.LGP_KERNEL_EXPQUAD:
            if d > 8:
                raise NotImplementedError('derivatives of points with more than 8 fields')
            total = [a + b for a, b in zip(da, db)]
            if max(total) > 4:
                raise NotImplementedError('total derivative order above 4 along one field')
            factor = factor * (-1.) ** sum(da) / scale ** sum(total)
            return _lib.LGP_KERNEL_EXPQUAD_DIFF, sum(total), tuple(float(t) for t in total), (('mul', factor),)
        if self._kind == _lib.LGP_KERNEL_MATERNP:
            p = int(self._p)
            a, b = da[0], db[0]
            if d != 1:
                raise NotImplementedError('derivatives of Maternp are taken along the one variable of one-dimensional points')
            if a > p or b > p:
                raise ValueError(f'Maternp(p={p}) is derivable {p} times on each side, asked for {(a, b)}')      # _Kernel/_ops.py:156-165
            if p >= 16:
                raise NotImplementedError('Maternp derivatives for p < 16')
            P = numpy.polynomial.Polynomial([1.])
            t = numpy.polynomial.Polynomial([0., 1.])
            for k in reversed(range(p)):                           # the Horner recurrence of _special/_bessel.py:106-109 as a polynomial
                P = 1 + P * ((p - k) / ((2 * p - k) * (k + 1)) * 2) * t
            Q = P
            for _ in range(a + b):
                Q = Q.deriv() - Q
            coef = numpy.zeros(p + 1)
            coef[:len(Q.coef)] = Q.coef
            c = numpy.sqrt(2 * p + 1)
            factor = factor * (-1.) ** b * (c / scale) ** (a + b)
            return _lib.LGP_KERNEL_MATERNP_DIFF, p, (float((a + b) % 2),) + tuple(float(v) for v in coef), (('mul', factor),)
        raise NotImplementedError('closed-form derivative cores exist for ExpQuad and Maternp (one-dimensional points); the other derivable kernels go through the truncated Taylor arithmetic of `lgp_gram_deriv`')

    def gram(self, xs, ys, *, ctx=None, out=None, deriv=(0, 0)):
        """ device Gram matrix of two flat point sets; ``deriv=(dx, dy)``: of the kernel derived in x and in y (`_diff_call`) """
        xd, yd, dtype, ctx, d = self._prepare(xs, ys, ctx)
        n, m = xd.shape[1], yd.shape[1]
        if out is None:
            out = DeviceArray.empty((n, m), dtype, ctx)
        assert out.shape == (n, m) and out.dtype == dtype
        if n == 0 or m == 0:
            return out
        da, db = self._orders(deriv[0], d), self._orders(deriv[1], d)
        if (any(da) or any(db)) and not self._has_closed_form_diff(d):
            return self._gram_jet(xd, yd, dtype, ctx, d, da, db, out)
        kind, order, par, outops = self._diff_call(deriv, d)
        f, F = _lib.fn('lgp_gram_isotropic', dtype)
        nops, codes, vals = _ops_arrays(outops, F)
        npar, hpar = (len(par), (F * len(par))(*par)) if par else (0, None)
        _lib.check(f(ctx.handle, kind, order, npar, hpar, d, xd.ptr, xd.ld, n, yd.ptr, yd.ld, m,
                     int(self._loc is not None), F(self._loc or 0), int(self._scale is not None), F(self._scale or 1),
                     nops, codes, vals, int(xd is yd and kind == self._kind), out.ptr, out.ld), 'lgp_gram_isotropic')
        return out

    def gram_dscale(self, xs, ys, *, ctx=None):
        """ ``(K, dK/dscale)`` as two device matrices from one pass over the pairs (``lgp_gram_isotropic_dscale``): the
        tangent that ``jax.jacfwd`` of a reference gpfactory produces for a ``scale`` hyperparameter
        (``_Kernel/_ops.py:310-326`` through the core); with ``scale=None`` the derivative at scale = 1 """
        if self._kind not in (_lib.LGP_KERNEL_EXPQUAD, _lib.LGP_KERNEL_MATERNP, _lib.LGP_KERNEL_GAMMAEXP, _lib.LGP_KERNEL_CAUCHY, _lib.LGP_KERNEL_MATERN,
                              _lib.LGP_KERNEL_COLOR, _lib.LGP_KERNEL_PINK, _lib.LGP_KERNEL_BESSEL):
            raise NotImplementedError(f'derivative of {type(self).__name__} with respect to scale')
        xd, yd, dtype, ctx, d = self._prepare(xs, ys, ctx)
        n, m = xd.shape[1], yd.shape[1]
        out = DeviceArray.empty((n, m), dtype, ctx)
        dout = DeviceArray.empty((n, m), dtype, ctx)
        if n == 0 or m == 0:
            return out, dout
        f, F = _lib.fn('lgp_gram_isotropic_dscale', dtype)
        nops, codes, vals = _ops_arrays(self._outops, F)
        _lib.check(f(ctx.handle, self._kind, int(self._p), *self._par_args(F), d, xd.ptr, xd.ld, n, yd.ptr, yd.ld, m,
                     int(self._loc is not None), F(self._loc or 0), int(self._scale is not None), F(self._scale or 1),
                     nops, codes, vals, out.ptr, dout.ptr, out.ld), 'lgp_gram_isotropic_dscale')
        return out, dout

    def gram_tangents(self, xs, ys, *, ctx=None, deriv=(0, 0)):
        """
        The Gram matrix and its tangents with respect to the keys of the hyperparameters that came as `_dual.Dual`:
        ``scale`` (through ``lgp_gram_isotropic_dscale``: the core's derivative in one pass with the values), the constants
        of ``kernel * c`` and ``kernel + c`` (the output operations are an affine map  a C + b  of the core's matrix C, so
        dK = a dC/dscale dscale + da C + db), ``loc`` (nothing: the kernels are stationary).  What ``jax.jacfwd`` of a
        reference gpfactory produces for these hyperparameters (``_Kernel/_ops.py:310-326``, ``_alg.py:32-82``).
        """
        keys = self.tangent_keys()
        K = self.gram(xs, ys, ctx=ctx, deriv=deriv)
        live = [k for k in keys if k in self._dscale or any(k in d for d in self._doutops)]
        if not live:
            return K, {}
        if any(numpy.any(numpy.asarray(o) != 0) for o in deriv):
            return K, self._derived_tangents(xs, ys, ctx, deriv, live)
        core = self._copy(_outops=(), _doutops=(), _dscale={})
        if self._dscale:
            C, dC = core.gram_dscale(xs, ys, ctx=ctx)
        else:
            C, dC = core.gram(xs, ys, ctx=ctx), None
        a, b = _dual.Dual(1.), _dual.Dual(0.)
        for (op, c), dc in zip(self._outops, self._doutops):
            cd = _dual.Dual(float(c), dc)
            if op == 'mul':
                a, b = a * cd, b * cd
            else:
                b = b + cd
        fa, F = _lib.fn('lgp_axpby', C.dtype)
        fs, _ = _lib.fn('lgp_add_scalar', C.dtype)
        tangents = {}
        for k in live:
            ds, da, db = float(self._dscale.get(k, 0.)), float(a.tangents.get(k, 0.)), float(b.tangents.get(k, 0.))
            T = DeviceArray.empty(C.shape, C.dtype, C.ctx)
            if C.size:
                # T = (a ds) dC + da C   (+ db)
                _lib.check(fa(C.ctx.handle, C.shape[0], C.shape[1], F(a.value * ds), None if dC is None else dC.ptr, 0 if dC is None else dC.ld,
                              F(da), C.ptr, C.ld, T.ptr, T.ld), 'lgp_axpby')
                if db != 0.:
                    _lib.check(fs(C.ctx.handle, C.shape[0], C.shape[1], T.ptr, T.ld, F(db)), 'lgp_add_scalar')
            tangents[k] = T
        return K, tangents

    def _derived_tangents(self, xs, ys, ctx, deriv, live):
        """ tangents of a DERIVED block (derivative observations, ``_Kernel/_ops.py:115-211``) -- what ``jax.jacfwd`` of the
        reference's factory gives through ``diff``.  The kernels are stationary functions of u = (x - y) / scale, so with orders
        alpha on the x side, beta on the y side and q = |alpha| + |beta|,  K_ab = scale^-q (-1)^|beta| (d^(alpha+beta) f)(u)  and

            dK_ab / dscale = -(1 / scale) [ q K_ab + sum_f (x_f - y_f) K_(alpha + e_f, beta) ]

        -- one more derived block per field (the extra order goes to the side that has room: d/dx = -d/dy).  Scalar factors
        multiply, added constants have dropped out of a derived block. """
        xd, yd, dtype, ctx, d = self._prepare(xs, ys, ctx)
        da, db = self._orders(deriv[0], d), self._orders(deriv[1], d)
        core = self._copy(_outops=(), _doutops=(), _dscale={})
        C = core.gram(xs, ys, ctx=ctx, deriv=(da, db))
        a = _dual.Dual(1.)
        for (op, c), dc in zip(self._outops, self._doutops):
            if op == 'mul':
                a = a * _dual.Dual(float(c), dc)
        dC = None
        if self._dscale:
            scale = float(self._scale)
            q = sum(da) + sum(db)
            xh, yh = xd.to_host().astype(float), (xd if yd is xd else yd).to_host().astype(float)
            dC = _dev_axpby(-q / scale, C, 0., None)
            for f in range(d):
                sign = 1.
                try:
                    E = core.gram(xs, ys, ctx=ctx, deriv=(tuple(o + (g == f) for g, o in enumerate(da)), db))
                except (NotImplementedError, ValueError):
                    E = core.gram(xs, ys, ctx=ctx, deriv=(da, tuple(o + (g == f) for g, o in enumerate(db))))
                    sign = -1.
                D = DeviceArray.from_host((xh[f][:, None] - yh[f][None, :]).astype(dtype), ctx=ctx)
                dC = _dev_mul(-sign / scale, D, E, 1., dC, out=dC)
        tangents = {}
        for k in live:
            ds, dak = float(self._dscale.get(k, 0.)), float(a.tangents.get(k, 0.))
            # T = (a ds) dC + da C
            tangents[k] = _dev_axpby(a.value * ds, dC, dak, C) if dC is not None else _dev_axpby(dak, C, 0., None)
        return tangents

    def pairs(self, xs, ys, *, ctx=None):
        xd, yd, dtype, ctx, d = self._prepare(xs, ys, ctx)
        n = xd.shape[1]
        assert yd.shape[1] == n
        out = DeviceArray.empty((n,), dtype, ctx)
        if n == 0:
            return out
        f, F = _lib.fn('lgp_pairs_isotropic', dtype)
        nops, codes, vals = _ops_arrays(self._outops, F)
        _lib.check(f(ctx.handle, self._kind, int(self._p), *self._par_args(F), d, xd.ptr, xd.ld, yd.ptr, yd.ld, n,
                     int(self._loc is not None), F(self._loc or 0), int(self._scale is not None), F(self._scale or 1),
                     nops, codes, vals, out.ptr), 'lgp_pairs_isotropic')
        return out


class ExpQuad(_Isotropic):
    """ Exponential quadratic kernel, k(r) = exp(-r²/2) (``_kernels/_basic.py:61-75``) """
    _kind = _lib.LGP_KERNEL_EXPQUAD
    _derivable = float('inf')

    def __init__(self, *, scale=None, loc=None, dim=None):
        super().__init__(scale=scale, loc=loc, dim=dim)


class Maternp(_Isotropic):
    """ Matérn kernel of half-integer order ν = p + 1/2 (``_kernels/_matern.py:29-49``) """
    _kind = _lib.LGP_KERNEL_MATERNP

    def __init__(self, *, p=None, scale=None, loc=None, dim=None):
        if p is None or int(p) != p or p < 0:                      # _matern.py:46-47
            raise AssertionError(p)
        if p > 16:
            raise NotImplementedError('Maternp order above 16')
        super().__init__(scale=scale, loc=loc, dim=dim)
        self._p = int(p)

    def _derivability(self):
        if self._kind == _lib.LGP_KERNEL_MATERNP:
            return int(self._p)
        return int(max(0, numpy.ceil(self._nu) - 1))              # real order: _matern.py:51-53

    def _jet_pars(self):
        """ k = f(s), s = (2p+1) r^2, t = sqrt(s): f^(k)(s) = exp(-t) R_k(t) with R_0 = P_p (``_special/_bessel.py:101-110``) and
        R_(k+1) = (R_k' - R_k) / (2 t) -- Laurent polynomials in t (the division is exact up to the p-th derivative; beyond it
        the negative powers meet powers of the distance in the chain rule), entry i of a row the coefficient of t^(i - 7) """
        from fractions import Fraction
        if self._kind != _lib.LGP_KERNEL_MATERNP:       # real order: the library derives its constants from nu (csrc/bessel_k_dev.h)
            return 0, tuple(self._par), ()
        p = int(self._p)
        # exact rational arithmetic: a rounding residue of 1e-17 in a coefficient of 1 / t is 1e-2 at a coincidence (t = 1e-15)
        OFF, W = 7, 24
        P = [Fraction(1)]                                              # ascending powers of t
        for k in reversed(range(p)):
            ck = Fraction(2 * (p - k), (2 * p - k) * (k + 1))
            P = [Fraction(1)] + [ck * c for c in P]                    # 1 + ck t P
        row = [Fraction(0)] * W
        row[OFF:OFF + len(P)] = P
        rows = [row]
        for _ in range(4):
            r = rows[-1]
            d = [r[i + 1] * (i + 1 - OFF) for i in range(W - 1)] + [Fraction(0)]       # derivative: coefficient of t^(e - 1) -> slot of t^(e - 1)
            q = [(di - ri) / 2 for di, ri in zip(d, r)]
            assert q[0] == 0
            rows.append(q[1:] + [Fraction(0)])                                          # divided by t
        return p, (), tuple(float(v) for row in rows for v in row)


class Matern(Maternp):
    """ Matérn kernel of real order ν, 2^(1-ν)/Γ(ν) x^ν K_ν(x), x = sqrt(2ν) r (``_kernels/_matern.py:54-76``).  For the half-integer
    orders ν = p + 1/2 this is the polynomial-times-exponential form that `Maternp` evaluates, and that core is used; the other
    orders go through ``LGP_KERNEL_MATERN``, which evaluates the modified Bessel function of real order on the device (Temme's
    series / Steed's continued fraction and the upward recurrence); ν = 0 is 1 where r = 0 (``_matern.py:74``) """

    def __init__(self, *, nu=None, scale=None, loc=None, dim=None):
        assert nu is not None and 0 <= nu < numpy.inf, nu
        p = float(nu) - 0.5
        if p >= 0 and p == int(p) and p <= 16:
            super().__init__(p=int(p), scale=scale, loc=loc, dim=dim)
        else:
            if nu > 100:
                raise NotImplementedError('Matern order above 100')
            _Isotropic.__init__(self, scale=scale, loc=loc, dim=dim)
            self._kind = _lib.LGP_KERNEL_MATERN
            self._par = (float(nu),)
        self._nu = nu


class Bessel(_Isotropic):
    """ Bessel kernel, k(r) = Γ(ν + 1) 2^ν (s r)^-ν J_ν(s r), s = 2 + ν/2, ν >= 0 (``_kernels/_matern.py:92-119``): usable in up to
    2 (⌊ν⌋ + 1) dimensions.  J_ν of real order on the device (power series, Miller's backward recurrence, Hankel's expansion) """
    _kind = _lib.LGP_KERNEL_BESSEL

    def _derivability(self):
        return int(self._par[0] // 2)                                      # _matern.py:92-94

    def __init__(self, *, nu=0, scale=None, loc=None, dim=None):
        assert 0 <= nu < numpy.inf, nu                                     # _matern.py:117
        if nu > 100:
            raise NotImplementedError('Bessel order above 100')
        super().__init__(scale=scale, loc=loc, dim=dim)
        self._par = (float(nu),)
        self._maxdim = 2 * (int(numpy.floor(nu)) + 1)                      # _matern.py:97-99

    def _prepare(self, xs, ys, ctx):
        out = super()._prepare(xs, ys, ctx)
        if out[4] > self._maxdim:
            raise ValueError(f'kernel called on type with dimensionality {out[4]} > maxdim={self._maxdim}')
        return out


class GammaExp(_Isotropic):
    """ Gamma exponential kernel, k(r) = exp(-r^γ), γ ∈ (0, 2] (``_kernels/_basic.py:94-115``) """
    _kind = _lib.LGP_KERNEL_GAMMAEXP

    def __init__(self, *, gamma=1, scale=None, loc=None, dim=None):
        if not 0 < gamma <= 2:                                     # _basic.py:109-110
            raise AssertionError(gamma)
        super().__init__(scale=scale, loc=loc, dim=dim)
        self._gamma = gamma
        self._par = (float(gamma), float(gamma / 2))

    def _derivability(self):
        return float('inf') if self._gamma == 2 else 0              # _basic.py:94


class Cauchy(_Isotropic):
    """ Generalized Cauchy kernel, k(r) = (1 + r^α/β)^(-β/α), α ∈ (0, 2], β > 0 (``_kernels/_basic.py:315-344``);
    α = 2 is the rational quadratic kernel """
    _kind = _lib.LGP_KERNEL_CAUCHY

    def __init__(self, *, alpha=2, beta=2, scale=None, loc=None, dim=None):
        if not 0 < alpha <= 2:                                     # _basic.py:336-338
            raise AssertionError(alpha)
        if not 0 < beta:
            raise AssertionError(beta)
        super().__init__(scale=scale, loc=loc, dim=dim)
        self._alpha, self._beta = alpha, beta
        self._par = (float(alpha), float(beta), float(-beta / alpha))

    def _derivability(self):
        return float('inf') if self._alpha == 2 else 0              # _basic.py:312-315

    def _jet_pars(self):
        return 0, (float(self._alpha / 2), float(self._beta), float(-self._beta / self._alpha)), ()


class Wendland(_Isotropic):
    """ Wendland kernel, (1 - r)_+^(nu + k) poly_k(r) with nu = k + alpha: compact support r < 1, k in {0, 1, 2, 3} sets the
    differentiability, usable up to floor(2 alpha - 1) dimensions (``_kernels/_wendland.py:26-84``); input='posabs' """
    _kind = _lib.LGP_KERNEL_WENDLAND
    _POLY = {0: [[1]], 1: [[1, 1], [1]], 2: [[1 / 3, 4 / 3, 1], [1, 2], [1]], 3: [[1 / 15, 3 / 5, 23 / 15, 1], [2 / 5, 12 / 5, 3], [1, 3], [1]]}

    def __init__(self, *, k=0, alpha=1, scale=None, loc=None, dim=None):
        if k not in self._POLY:
            raise NotImplementedError(k)                              # _wendland.py:77-78
        self._maxdim = int(numpy.floor(2 * alpha - 1))               # :29-31
        assert self._maxdim >= 1, self._maxdim
        super().__init__(scale=scale, loc=loc, dim=dim)
        nu = k + alpha
        self._p = int(k)
        self._par = (float(nu + k),) + tuple(float(numpy.polyval(pj, nu)) for pj in self._POLY[k])

    def _derivability(self):
        return int(self._p)                                         # _wendland.py:25-26

    def _jet_pars(self):
        return int(self._p), (self._par[0],), tuple(self._par[1:])

    def _gram_jet(self, xd, yd, dtype, ctx, d, da, db, out):
        if d != 1:
            raise NotImplementedError('derivatives of Wendland on one-dimensional points only')
        return super()._gram_jet(xd, yd, dtype, ctx, d, da, db, out)

    def _prepare(self, xs, ys, ctx):
        out = super()._prepare(xs, ys, ctx)
        if out[4] > self._maxdim:
            raise ValueError(f'kernel called on type with dimensionality {out[4]} > maxdim={self._maxdim}')   # _crosskernel.py maxdim check
        return out


class CausalExpQuad(_Isotropic):
    """ Causal exponential quadratic kernel, k(r) = erfc(alpha r / 4) exp(-r²/2), alpha >= 0 (``_kernels/_basic.py:346-360``) """
    _kind = _lib.LGP_KERNEL_CAUSALEXPQUAD

    def __init__(self, *, alpha=1, scale=None, loc=None, dim=None):
        assert alpha >= 0, alpha
        super().__init__(scale=scale, loc=loc, dim=dim)
        self._par = (float(alpha / 4),)

    def _derivability(self):
        return float('inf') if self._par[0] == 0 else 0                    # _basic.py:346

    def _jet_pars(self):
        return 0, (), ()


class Log(_Isotropic):
    """ Log kernel, k(r) = log(1 + r) / r (``_kernels/_basic.py:381-392``) """
    _kind = _lib.LGP_KERNEL_LOG

    def __init__(self, *, scale=None, loc=None, dim=None):
        super().__init__(scale=scale, loc=loc, dim=dim)


class Constant(_Isotropic):
    """ Constant kernel, k(x, y) = 1 (``_kernels/_basic.py:34-46``) """
    _kind = _lib.LGP_KERNEL_CONSTANT
    _derivable = float('inf')

    def __init__(self, *, scale=None, loc=None, dim=None):
        super().__init__(scale=scale, loc=loc, dim=dim)


class White(_Isotropic):
    """ White noise kernel, k(x, y) = 1 if x == y (every field) else 0 (``_kernels/_basic.py:48-59``); the values come as
    floats where the reference returns integers """
    _kind = _lib.LGP_KERNEL_WHITE

    def __init__(self, *, scale=None, loc=None, dim=None):
        super().__init__(scale=scale, loc=loc, dim=dim)


class _Stationary1D(_Isotropic):
    """ stationary kernels of one variable (``stationarykernel(maxdim=1)``, ``_Kernel/_stationary.py``): functions of the
    difference of the (transformed) inputs, evaluated by the generic core of the isotropic Gram kernel """

    def __init__(self, *, scale=None, loc=None, dim=None):
        super().__init__(scale=scale, loc=loc, dim=dim)

    def _prepare(self, xs, ys, ctx):
        out = super()._prepare(xs, ys, ctx)
        if out[4] > 1:
            raise ValueError(f'kernel called on type with dimensionality {out[4]} > maxdim=1')                 # _crosskernel.py maxdim check
        return out


class Expon(_Stationary1D):
    """ Exponential kernel, k(Δ) = exp(-|Δ|) (``_kernels/_basic.py:254-269``) """
    _kind = _lib.LGP_KERNEL_EXPON


class HoleEffect(_Stationary1D):
    """ Hole effect kernel, k(Δ) = (1 - |Δ|) exp(-|Δ|) (``_kernels/_basic.py:299-310``) """
    _kind = _lib.LGP_KERNEL_HOLEEFFECT


class Periodic(_Stationary1D):
    """ Periodic Gaussian kernel, k(Δ) = exp(-2 (sin(Δ/2) / outerscale)²): ``scale`` sets the period (2π at scale = 1),
    ``outerscale`` the correlation length (``_kernels/_basic.py:187-208``) """
    _kind = _lib.LGP_KERNEL_PERIODIC
    _derivable = float('inf')

    def __init__(self, *, outerscale=1, scale=None, loc=None, dim=None):
        assert 0 < outerscale < numpy.inf, outerscale
        super().__init__(scale=scale, loc=loc, dim=dim)
        self._par = (float(outerscale),)


class Cos(_Stationary1D):
    """ Cosine kernel, k(Δ) = cos(Δ) (``_kernels/_spectral.py:25-38``) """
    _kind = _lib.LGP_KERNEL_COS
    _derivable = float('inf')


class Sinc(_Stationary1D):
    """ Sinc kernel, k(Δ) = sin(πΔ) / (πΔ) (``_kernels/_spectral.py:113-122``; six Taylor terms below |Δ| = 0.1 as
    ``_special/_sinc.py:25-32``), input='posabs' """
    _kind = _lib.LGP_KERNEL_SINC
    _derivable = float('inf')


class StationaryFracBrownian(_Stationary1D):
    """ Stationary fractional Brownian motion kernel, k(Δ) = (|Δ + 1|^2H + |Δ - 1|^2H - 2 |Δ|^2H) / 2, H in (0, 1]
    (``_kernels/_randomwalk.py:156-172``) """
    _kind = _lib.LGP_KERNEL_STATIONARYFRACBROWNIAN

    def __init__(self, *, H=1 / 2, scale=None, loc=None, dim=None):
        assert 0 < H <= 1, H
        super().__init__(scale=scale, loc=loc, dim=dim)
        self._par = (float(2 * H),)
        self._H = H

    def _derivability(self):
        return float('inf') if self._H == 1 else 0                  # _randomwalk.py:153-154


class Color(_Stationary1D):
    """ Colored noise kernel, k(Δ) = (n - 1) Re E_n(-iΔ) = (n - 1) ∫_1^∞ cos(ωΔ) / ω^n dω, integer n >= 2: a power spectrum
    1/ω^n truncated below ω = 1 (``_kernels/_spectral.py:70-111``).  E_n on the device by its power series and its continued
    fraction (the reference's closed form and asymptotic series meet at a knee where both have lost digits, 1e-4 at n = 20) """
    _kind = _lib.LGP_KERNEL_COLOR

    def _derivability(self):
        return int(self._p) // 2 - 1                                        # _spectral.py:70-71

    def _jet_pars(self):
        return int(self._p), (), ()

    def __init__(self, *, n=2, scale=None, loc=None, dim=None):
        assert int(n) == n and n >= 2, n                                   # _spectral.py:110
        super().__init__(scale=scale, loc=loc, dim=dim)
        self._p = int(n)


class Pink(_Stationary1D):
    """ Pink noise kernel, k(Δ) = (Ci(Δ (1 + dw)) - Ci(Δ)) / log(1 + dw): a power spectrum 1/ω between 1 and 1 + dw
    (``_kernels/_spectral.py:41-66``); where Δ dw is below sqrt(eps), cos(Δ (1 + dw/2)) """
    _kind = _lib.LGP_KERNEL_PINK
    _derivable = 1                                                          # _spectral.py:41

    def __init__(self, *, dw=1, scale=None, loc=None, dim=None):
        assert 0 <= dw < numpy.inf, dw
        super().__init__(scale=scale, loc=loc, dim=dim)
        self._par = (float(dw),)


class Zeta(_Stationary1D):
    """ Zeta kernel, k(Δ) = Re Li_s(e^{2πiΔ}) / ζ(s) = Σ_k cos(2πkΔ) / k^s / ζ(s), s = 1 + 2ν, ν >= 0: a Fourier series of period 1
    with independent coefficients of variance 1 / (ζ(s) k^s) (``_kernels/_zeta.py:37-70``).  The host computes the coefficients of
    the polylogarithm's expansion on the unit circle once per kernel (`_zeta.coefficients`), the device sums it pair by pair
    (``lgp_gram_pseries``) """

    def __init__(self, *, nu=None, scale=None, loc=None, dim=None):
        assert nu is not None and 0 <= nu < numpy.inf, nu              # _zeta.py:28-30
        super().__init__(scale=scale, loc=loc, dim=dim)
        self._nu = float(nu)
        self._mode, self._uselog, self._coef, self._hpar4 = _zeta.coefficients(nu)
        self._coef_memo = {}

    def _copy(self, **kw):
        new = super()._copy(**kw)
        new._coef_memo = {}
        return new

    _kind = _lib.LGP_KERNEL_PSERIES                                         # (the derivative core; the values go through lgp_gram_pseries)

    def _derivability(self):
        return int(max(0, numpy.ceil(self._nu) - 1))                      # _kernels/_zeta.py:32-35

    def _has_closed_form_diff(self, d):
        return False

    def _jet_pars(self):
        if self._mode == _zeta.MODE_INTEGERS:
            raise ValueError('Took 1 derivatives > limit 0 on argument of a kernel (Zeta).')
        return int(self._mode) | int(self._uselog) << 4 | int(self._coef.size) << 8, tuple(float(v) for v in self._hpar4), tuple(float(v) for v in self._coef)

    def _table(self, ctx, dtype):
        key = (id(ctx), numpy.dtype(dtype).str)
        tab = self._coef_memo.get(key)
        if tab is None:
            tab = DeviceArray.from_host(numpy.ascontiguousarray(self._coef, dtype), ctx=ctx) if self._coef.size else None
            self._coef_memo = {key: tab}
        return tab

    def gram(self, xs, ys, *, ctx=None, out=None, deriv=(0, 0)):
        if any(numpy.any(numpy.asarray(o) != 0) for o in deriv):
            return _Isotropic.gram(self, xs, ys, ctx=ctx, out=out, deriv=deriv)      # -> `_gram_jet`: the expansion differentiated term by term
        xd, yd, dtype, ctx, d = self._prepare(xs, ys, ctx)
        n, m = xd.shape[1], yd.shape[1]
        if out is None:
            out = DeviceArray.empty((n, m), dtype, ctx)
        assert out.shape == (n, m) and out.dtype == dtype
        if n == 0 or m == 0:
            return out
        f, F = _lib.fn('lgp_gram_pseries', dtype)
        tab = self._table(ctx, dtype)
        nops, codes, vals = _ops_arrays(self._outops, F)
        _lib.check(f(ctx.handle, self._mode, self._uselog, None if tab is None else tab.ptr, self._coef.size, (F * 4)(*self._hpar4),
                     xd.ptr, xd.ld, n, yd.ptr, yd.ld, m, int(self._loc is not None), F(self._loc or 0), int(self._scale is not None), F(self._scale or 1),
                     nops, codes, vals, int(xd is yd), out.ptr, out.ld), 'lgp_gram_pseries')
        return out

    def gram_dscale(self, xs, ys, *, ctx=None):
        raise NotImplementedError('derivative of Zeta with respect to scale')

    def pairs(self, xs, ys, *, ctx=None):
        xd, yd, dtype, ctx, d = self._prepare(xs, ys, ctx)
        n = xd.shape[1]
        assert yd.shape[1] == n
        out = DeviceArray.empty((n,), dtype, ctx)
        if n == 0:
            return out
        f, F = _lib.fn('lgp_pairs_pseries', dtype)
        tab = self._table(ctx, dtype)
        nops, codes, vals = _ops_arrays(self._outops, F)
        _lib.check(f(ctx.handle, self._mode, self._uselog, None if tab is None else tab.ptr, self._coef.size, (F * 4)(*self._hpar4),
                     xd.ptr, xd.ld, yd.ptr, yd.ld, n, int(self._loc is not None), F(self._loc or 0), int(self._scale is not None), F(self._scale or 1),
                     nops, codes, vals, out.ptr), 'lgp_pairs_pseries')
        return out


def _split_vector(v):
    """ a 1-D array of hyperparameters, some of them possibly `_dual.Dual` (or one Dual array): ``(values, {key: tangents})`` """
    if isinstance(v, _dual.Dual):
        val = numpy.atleast_1d(numpy.asarray(v.value, float))
        return val, {k: numpy.atleast_1d(v.tangent(k)).astype(float) for k in v.tangents}
    items = list(v) if not isinstance(v, numpy.ndarray) else list(v.reshape(-1) if v.dtype != object else v)
    val = numpy.array([float(e.value) if isinstance(e, _dual.Dual) else float(e) for e in items], float)
    tan = {}
    for i, e in enumerate(items):
        if isinstance(e, _dual.Dual):
            for k, t in e.tangents.items():
                tan.setdefault(k, numpy.zeros(len(items)))[i] = float(t)
    return val, tan


class _LagTable(Kernel):
    """ stationary kernels of one INTEGER variable tabulated over the lag |x - y| (the reference's discrete-time kernels,
    ``_kernels/_arma.py``): the host computes the table, the device looks it up pair by pair (``lgp_gram_lagtable``) """

    _fill = 0.

    def __init__(self, *, scale=None, loc=None, dim=None):
        if scale is not None or loc is not None:
            # (x - loc) / scale makes the lags floating point numbers, which the reference cannot index its table with either
            raise TypeError(f'{type(self).__name__}: the inputs must be integers, scale and loc do not apply')
        super().__init__(dim=dim)
        self._table_memo = {}

    def _copy(self, **kw):
        new = super()._copy(**kw)
        new._table_memo = {}
        return new

    def _table(self, maxlag):  # pragma: no cover
        """ values at the lags 0 .. (at least) maxlag, or fewer when the kernel is defined by a fixed table """
        raise NotImplementedError

    def _result_dtype(self, x, y):
        return numpy.dtype(numpy.float64)

    def _ints(self, x):
        x = numpy.asarray(self._select(x))
        if x.dtype.names is not None:
            cols = _leaf_columns(x.reshape(-1))
            if len(cols) != 1:
                raise ValueError(f'kernel called on type with dimensionality {len(cols)} > maxdim=1')
            x = cols[0]
        x = x.reshape(-1)
        if not (numpy.issubdtype(x.dtype, numpy.integer) or numpy.issubdtype(x.dtype, numpy.bool_)):
            raise TypeError(f'{type(self).__name__}: the inputs must be integers, got {x.dtype}')
        return x.astype(numpy.int64)

    def prepare(self, xs, *, ctx=None, dtype=None):
        x = self._ints(xs)
        return _PointBundle([x], len(x), numpy.float64)

    def _device_args(self, xs, ys, ctx):
        ctx = ctx or Context.default()
        x = xs.parts[0] if isinstance(xs, _PointBundle) else self._ints(xs)
        y = x if ys is xs else (ys.parts[0] if isinstance(ys, _PointBundle) else self._ints(ys))
        maxlag = 0
        if x.size and y.size:
            maxlag = int(max(x.max() - y.min(), y.max() - x.min(), 0))
        key = (id(ctx), maxlag)
        tab = self._table_memo.get(key)
        if tab is None:
            host = numpy.ascontiguousarray(self._table(maxlag), float).reshape(-1)
            tab = (DeviceArray.from_host(host, ctx=ctx) if host.size else None, host.size)
            self._table_memo = {key: tab}
        # the device works on the points as doubles: shift them so that the offsets stay exact whatever the absolute times
        base = int(min(x.min(), y.min())) if x.size and y.size else 0
        xd = DeviceArray.from_host((x - base).astype(float)[None, :], ctx=ctx)
        yd = xd if y is x else DeviceArray.from_host((y - base).astype(float)[None, :], ctx=ctx)
        return ctx, xd, yd, tab

    def gram(self, xs, ys, *, ctx=None, out=None, deriv=(0, 0), _table=None):
        if any(numpy.any(numpy.asarray(o) != 0) for o in deriv):
            raise NotImplementedError(f'derivatives of {type(self).__name__}')         # derivable=False
        ctx, xd, yd, (tab, ntab) = self._device_args(xs, ys, ctx)
        if _table is not None:                          # (a tangent of the table: same look-up, other values)
            host = numpy.ascontiguousarray(_table, float).reshape(-1)
            tab, ntab = (DeviceArray.from_host(host, ctx=ctx) if host.size else None), host.size
        n, m = xd.shape[1], yd.shape[1]
        if out is None:
            out = DeviceArray.empty((n, m), numpy.float64, ctx)
        if n == 0 or m == 0:
            return out
        f, F = _lib.fn('lgp_gram_lagtable', numpy.float64)
        nops, codes, vals = _ops_arrays(self._outops, F)
        _lib.check(f(ctx.handle, None if tab is None else tab.ptr, ntab, F(self._fill), xd.ptr, xd.ld, n, yd.ptr, yd.ld, m, 0, F(0), 0, F(1),
                     nops, codes, vals, int(xd is yd), out.ptr, out.ld), 'lgp_gram_lagtable')
        return out

    # -- hyperparameters of the table as `_dual.Dual` (MA weights, AR coefficients): the tangent of the Gram matrix is the same
    # look-up in the tangent of the table
    _dtables = {}

    def _table_tangents(self, maxlag):
        """ ``{key: d table / d key}`` for the lags of `_table(maxlag)` """
        return {}

    def tangent_keys(self):
        keys = super().tangent_keys()
        for k in self._dkeys():
            if k not in keys:
                keys.append(k)
        return keys

    def _dkeys(self):
        return []

    def gram_tangents(self, xs, ys, *, ctx=None, deriv=(0, 0)):
        keys = self.tangent_keys()
        K = self.gram(xs, ys, ctx=ctx, deriv=deriv)
        if not keys:
            return K, {}
        core = self._copy(_outops=(), _doutops=())
        C = core.gram(xs, ys, ctx=ctx)
        x = xs.parts[0] if isinstance(xs, _PointBundle) else self._ints(xs)
        y = x if ys is xs else (ys.parts[0] if isinstance(ys, _PointBundle) else self._ints(ys))
        maxlag = int(max(x.max() - y.min(), y.max() - x.min(), 0)) if x.size and y.size else 0
        dtabs = self._table_tangents(maxlag)
        a, b = _dual.Dual(1.), _dual.Dual(0.)
        for (op, c), dc in zip(self._outops, self._doutops):
            cd = _dual.Dual(float(c), dc)
            if op == 'mul':
                a, b = a * cd, b * cd
            else:
                b = b + cd
        fa, F = _lib.fn('lgp_axpby', C.dtype)
        fs, _ = _lib.fn('lgp_add_scalar', C.dtype)
        tangents = {}
        for k in keys:
            da, db = float(a.tangents.get(k, 0.)), float(b.tangents.get(k, 0.))
            dC = core.gram(xs, ys, ctx=ctx, _table=dtabs[k]) if k in dtabs else None
            T = DeviceArray.empty(C.shape, C.dtype, C.ctx)
            if C.size:
                _lib.check(fa(C.ctx.handle, C.shape[0], C.shape[1], F(a.value), None if dC is None else dC.ptr, 0 if dC is None else dC.ld,
                              F(da), C.ptr, C.ld, T.ptr, T.ld), 'lgp_axpby')
                if db != 0.:
                    _lib.check(fs(C.ctx.handle, C.shape[0], C.shape[1], T.ptr, T.ld, F(db)), 'lgp_add_scalar')
            tangents[k] = T
        return K, tangents

    def pairs(self, xs, ys, *, ctx=None):
        ctx, xd, yd, (tab, ntab) = self._device_args(xs, ys, ctx)
        n = xd.shape[1]
        assert yd.shape[1] == n
        out = DeviceArray.empty((n,), numpy.float64, ctx)
        if n == 0:
            return out
        f, F = _lib.fn('lgp_pairs_lagtable', numpy.float64)
        nops, codes, vals = _ops_arrays(self._outops, F)
        _lib.check(f(ctx.handle, None if tab is None else tab.ptr, ntab, F(self._fill), xd.ptr, xd.ld, yd.ptr, yd.ld, n, 0, F(0), 0, F(1),
                     nops, codes, vals, out.ptr), 'lgp_pairs_lagtable')
        return out


class MA(_LagTable):
    """ Discrete moving average kernel, k(Δ) = Σ_{k ≥ |Δ|} w_k w_{k-|Δ|} on integer inputs: the autocovariance of the moving
    average with weights w of white noise; ``norm=True`` makes the variance 1 (``_kernels/_arma.py:31-63``) """

    def __init__(self, *, w=None, norm=False, scale=None, loc=None, dim=None):
        super().__init__(scale=scale, loc=loc, dim=dim)
        w, dw = _split_vector([] if w is None else w)
        assert w.ndim == 1

        def acov_of(w, dw=None):
            if not len(w):
                return numpy.zeros(0)
            if dw is None:
                return numpy.correlate(w, w, 'full')[len(w) - 1:]        # lags 0 .. len(w) - 1
            return (numpy.correlate(dw, w, 'full') + numpy.correlate(w, dw, 'full'))[len(w) - 1:]
        acov = acov_of(w)
        self._dacov = {k: acov_of(w, t) for k, t in dw.items()}
        if norm and len(w):
            with numpy.errstate(invalid='ignore', divide='ignore'):      # all-zero weights: not a number, as in the reference
                self._dacov = {k: t / acov[0] - acov * t[0] / acov[0] ** 2 for k, t in self._dacov.items()}
                acov = acov / acov[0]
        self._acov = acov

    def _table(self, maxlag):
        return self._acov                                              # beyond the last weight: 0 (`mode='fill', fill_value=0`)

    def _dkeys(self):
        return list(self._dacov)

    def _table_tangents(self, maxlag):
        return self._dacov


class AR(_LagTable):
    """ Discrete autoregressive kernel on integer inputs: the autocovariance γ_m of  y_i = Σ_k φ_k y_{i-k} + ε_i  with unit
    white noise ε (``_kernels/_arma.py:65-189``).  One of three parametrisations: ``phi`` (coefficients at lag 1..p) with
    ``maxlag``; ``gamma`` (autocovariance at lag 0..p) with ``maxlag``; the roots of the characteristic polynomial
    1 - Σ φ_k x^k as ``slnr`` (real roots: sign = sign of the root, magnitude = log |root|) and ``lnc`` (logarithms of the complex
    roots, each standing for itself and its conjugate).  Lags above ``maxlag`` give nan; ``norm=True`` makes the variance 1.
    The first p + 1 values come from the Yule-Walker equations; all three parametrisations then run the recursion
    γ_m = Σ_k φ_k γ_{m-k} (the reference evaluates the roots' form through amplitudes of |m|^l x_j^-|m|, the same function) """

    _fill = numpy.nan

    def __init__(self, *, phi=None, gamma=None, maxlag=None, slnr=None, lnc=None, norm=False, scale=None, loc=None, dim=None):
        super().__init__(scale=scale, loc=loc, dim=dim)
        given = (phi is not None, gamma is not None, maxlag is not None, slnr is not None, lnc is not None)
        self._invalid = given not in ((True, False, True, False, False), (False, True, True, False, False), (False, False, False, True, True))
        if self._invalid:              # _arma.py:164-170: the reference's core raises when the kernel is evaluated, not when it is made
            self._phi, self._gamma, self._maxlag = numpy.empty(0), numpy.ones(1), 0
            return
        self._dphi, self._dgamma = {}, {}
        if slnr is not None:
            phi = self.phi_from_roots(slnr, lnc)
        elif phi is None:
            phi = self.phi_from_gamma(gamma)
        else:
            phi, self._dphi = _split_vector(phi)          # coefficients as `_dual.Dual`: tangents of the autocovariance below
        if gamma is None:
            gamma = self.gamma_from_phi(phi)
            if self._dphi:                                 # A(phi) gamma = e0, A[m, |m-k|] -= phi_k:  d gamma = A^-1 (sum_k dphi_k E_k) gamma
                p = len(phi)
                A = numpy.eye(p + 1)
                for m in range(p + 1):
                    for k in range(1, p + 1):
                        A[m, abs(m - k)] -= phi[k - 1]
                for key, dph in self._dphi.items():
                    rhs = numpy.array([sum(dph[k - 1] * gamma[abs(m - k)] for k in range(1, p + 1)) for m in range(p + 1)])
                    self._dgamma[key] = numpy.linalg.solve(A, rhs)
        gamma = numpy.asarray(gamma, float)
        self._phi = numpy.asarray(phi, float)
        with numpy.errstate(invalid='ignore', divide='ignore'):
            if norm:
                self._dgamma = {k: t / gamma[0] - gamma * t[0] / gamma[0] ** 2 for k, t in self._dgamma.items()}
            self._gamma = gamma / gamma[0] if norm else gamma
        self._maxlag = None if maxlag is None else int(maxlag)
        if slnr is not None:
            self._fill = 0.     # never used: the table covers every lag of the inputs

    def _table(self, maxlag):
        if self._invalid:
            raise ValueError('invalid set of specified parameters')
        last = self._maxlag if self._maxlag is not None else maxlag
        return self.extend_gamma(self._gamma, self._phi, max(0, last + 1 - len(self._gamma)))[:last + 1]

    def _dkeys(self):
        return list(self._dphi)

    def _table_tangents(self, maxlag):
        """ the recursion differentiated: d gamma_m = sum_k (dphi_k gamma_{m-k} + phi_k dgamma_{m-k}) """
        if not self._dphi:
            return {}
        g = self._table(maxlag)
        p = len(self._phi)
        out = {}
        for key, dph in self._dphi.items():
            dg = numpy.zeros(len(g))
            head = self._dgamma[key][:len(g)]
            dg[:len(head)] = head
            for i in range(len(self._gamma), len(g)):
                dg[i] = sum(dph[k] * g[i - 1 - k] + self._phi[k] * dg[i - 1 - k] for k in range(p))
            out[key] = dg
        return out

    @staticmethod
    def phi_from_gamma(gamma):
        """ autoregressive coefficients at lag 1..p from the autocovariance at lag 0..p (Yule-Walker: the symmetric Toeplitz
        system  Σ_k φ_k γ_|m-k| = γ_m, m = 1..p) """
        gamma = numpy.asarray(gamma, float)
        assert gamma.ndim == 1 and gamma.size >= 1
        p = gamma.size - 1
        if p == 0:
            return numpy.empty(0)
        lag = numpy.abs(numpy.arange(p)[:, None] - numpy.arange(p)[None, :])
        return numpy.linalg.solve(gamma[lag], gamma[1:])

    @staticmethod
    def gamma_from_phi(phi):
        """ autocovariance at lag 0..p from the coefficients, unit noise variance:  γ_m - Σ_k φ_k γ_|m-k| = δ_m0, m = 0..p """
        phi = numpy.asarray(phi, float)
        assert phi.ndim == 1
        p = phi.size
        A = numpy.eye(p + 1)
        for m in range(p + 1):
            for k in range(1, p + 1):
                A[m, abs(m - k)] -= phi[k - 1]
        rhs = numpy.zeros(p + 1)
        rhs[0] = 1
        return numpy.linalg.solve(A, rhs)

    @staticmethod
    def extend_gamma(gamma, phi, n):
        """ ``n`` further lags of an autocovariance (at least p + 1 consecutive values given) by γ_m = Σ_k φ_k γ_{m-k} """
        gamma = numpy.asarray(gamma, float)
        phi = numpy.asarray(phi, float)
        assert gamma.ndim == 1 and phi.ndim == 1 and gamma.size > phi.size
        out = numpy.concatenate([gamma, numpy.zeros(max(0, int(n)))])
        rev = phi[::-1]
        for i in range(gamma.size, out.size):
            out[i] = rev @ out[i - phi.size:i] if phi.size else 0.
        return out

    @staticmethod
    def phi_from_roots(slnr, lnc):
        """ the coefficients from the roots of 1 - Σ φ_k x^k:  Π_j (z - 1/x_j) = z^p - Σ_k φ_k z^(p-k) """
        slnr = numpy.asarray(slnr, float)
        lnc = numpy.asarray(lnc, complex)
        assert slnr.ndim == lnc.ndim == 1
        inv_real = numpy.copysign(numpy.exp(-numpy.abs(slnr)), slnr)
        inv_cplx = numpy.exp(-lnc)
        coef = numpy.atleast_1d(numpy.poly(numpy.sort(numpy.concatenate([inv_real, inv_cplx, inv_cplx.conj()]))))
        assert numpy.all(numpy.abs(numpy.imag(coef)) <= 1e-4)
        return -numpy.real(coef)[1:]


class Celerite(_Stationary1D):
    """ Celerite kernel, k(Δ) = exp(-γ|Δ|) (cos Δ + B sin|Δ|), |B| <= γ (``_kernels/_celerite.py:34-52``) """
    _kind = _lib.LGP_KERNEL_CELERITE

    def __init__(self, *, gamma=1, B=0, scale=None, loc=None, dim=None):
        assert 0 <= gamma < numpy.inf, gamma
        assert abs(B) <= gamma, (B, gamma)
        super().__init__(scale=scale, loc=loc, dim=dim)
        self._par = (float(gamma), float(B))

    def _derivability(self):
        return 1 if self._par[0] == self._par[1] else 0             # _celerite.py:26-32: B == gamma


class Harmonic(_Stationary1D):
    """ Damped stochastically driven harmonic oscillator of quality factor Q (``_kernels/_celerite.py:54-121``): the four
    branches of the reference (Q < 1/2 in the cancellation-free form, Q < 1, Q = 1, Q > 1), chosen on the host """
    _kind = _lib.LGP_KERNEL_HARMONIC
    _derivable = 1

    def __init__(self, *, Q=1, scale=None, loc=None, dim=None):
        assert 0 < Q < numpy.inf, Q
        super().__init__(scale=scale, loc=loc, dim=dim)
        if Q < 1 / 2:
            self._p, eta = 0, numpy.sqrt((1 - Q) * (1 + Q))
        elif Q < 1:
            self._p, eta = 1, numpy.sqrt(1 - numpy.square(Q))
        elif Q == 1:
            self._p, eta = 2, 0.
        else:
            self._p, eta = 3, numpy.sqrt(numpy.square(Q) - 1)
        self._par = (float(Q), float(eta), float(numpy.expm1(1 / 2 * numpy.log1p(-numpy.square(Q)))) if Q < 1 else 0.)


class Circular(_Stationary1D):
    """ Circular kernel of period 1, W_c(t) = (1 + τ t/c)(1 - t/c)_+^τ of the geodesic distance t, τ >= 4, c in (0, 1/2]
    (``_kernels/_wendland.py:88-113``) """
    _kind = _lib.LGP_KERNEL_CIRCULAR
    _derivable = 1

    def __init__(self, *, tau=4, c=1 / 2, scale=None, loc=None, dim=None):
        assert tau >= 4, tau
        assert 0 < c <= 1 / 2, c
        super().__init__(scale=scale, loc=loc, dim=dim)
        self._par = (float(tau), float(c))


class Linear(_Isotropic):
    """ Dot product kernel, k(x, y) = sum_i x_i y_i (``_kernels/_basic.py:76-91``) """
    _kind = _lib.LGP_KERNEL_LINEAR
    _derivable = float('inf')

    def __init__(self, *, scale=None, loc=None, dim=None):
        super().__init__(scale=scale, loc=loc, dim=dim)


class _OneVariable(_Stationary1D):
    """ kernels of x and y themselves, one variable (``kernel(maxdim=1)``): random walks and the like, most of them defined
    for nonnegative inputs -- checked on the transformed points, as the reference's assertions do """
    _domain = (0., numpy.inf)

    def _prepare(self, xs, ys, ctx):
        lo, hi = self._domain
        for pts in (xs, ys):
            if isinstance(pts, DevicePoints) or lo == -numpy.inf:
                continue
            cols = _leaf_columns(numpy.asarray(self._select(pts)))
            for c in cols:
                v = numpy.asarray(c, float)
                if self._loc is not None:
                    v = v - self._loc
                if self._scale is not None:
                    v = v / self._scale
                assert numpy.all(v >= lo) and numpy.all(v <= hi), f'{type(self).__name__}: inputs outside [{lo}, {hi}]'
        return super()._prepare(xs, ys, ctx)


class Wiener(_OneVariable):
    """ Wiener kernel, k(x, y) = min(x, y), x, y >= 0 (``_kernels/_randomwalk.py:26-41``) """
    _kind = _lib.LGP_KERNEL_WIENER


class OrnsteinUhlenbeck(_OneVariable):
    """ Ornstein-Uhlenbeck process started at 0, k(x, y) = exp(-|x - y|) - exp(-(x + y)), x, y >= 0
    (``_kernels/_randomwalk.py:112-132``) """
    _kind = _lib.LGP_KERNEL_ORNSTEINUHLENBECK


class BrownianBridge(_OneVariable):
    """ Brownian bridge kernel, k(x, y) = min(x, y) - x y, x, y in [0, 1] (``_kernels/_randomwalk.py:134-151``) """
    _kind = _lib.LGP_KERNEL_BROWNIANBRIDGE
    _domain = (0., 1.)


class Decaying(_OneVariable):
    """ Decaying kernel, k(x, y) = 1 / (1 + x + y)^alpha, x, y, alpha >= 0 (``_kernels/_basic.py:362-379``) """
    _kind = _lib.LGP_KERNEL_DECAYING
    _derivable = float('inf')

    def __init__(self, *, alpha=1, scale=None, loc=None, dim=None):
        assert alpha >= 0, alpha
        super().__init__(scale=scale, loc=loc, dim=dim)
        self._par = (float(alpha),)


class FracBrownian(_OneVariable):
    """ Bifractional Brownian motion kernel, k(x, y) = ((|x|^2H + |y|^2H)^K - |x - y|^2HK) / 2^K, H, K in (0, 1]
    (``_kernels/_randomwalk.py:46-68``) """
    _kind = _lib.LGP_KERNEL_FRACBROWNIAN
    _domain = (-numpy.inf, numpy.inf)

    def __init__(self, *, H=1 / 2, K=1, scale=None, loc=None, dim=None):
        assert 0 < H <= 1, H
        assert 0 < K <= 1, K
        super().__init__(scale=scale, loc=loc, dim=dim)
        H2 = 2 * H
        self._par = (float(H2), float(K), float(1 / 2 ** K), float(H2 * K))
        self._HK = (H, K)

    def _derivability(self):
        return float('inf') if self._HK == (1, 1) else 0            # _randomwalk.py:43-44


class WienerIntegral(_OneVariable):
    """ Integral of the Wiener process, k(x, y) = a² (b - a / 3) / 2 with a = min(x, y), b = max(x, y), x, y >= 0
    (``_kernels/_randomwalk.py:93-110``) """
    _kind = _lib.LGP_KERNEL_WIENERINTEGRAL
    _derivable = 1


class Taylor(_OneVariable):
    """ Exponential-like power series kernel, k(x, y) = sum_k x^k y^k / k!^2 = I0(2 sqrt(x y)) (``_kernels/_basic.py:393-408``;
    J0 of the same argument where x y < 0) """
    _kind = _lib.LGP_KERNEL_TAYLOR
    _derivable = float('inf')                                              # _basic.py:393
    _domain = (-numpy.inf, numpy.inf)


class _HostFactors(Kernel):
    """ kernels that are products of per-point quantities evaluated on the host (user functions, table look-ups): the Gram
    matrix is a product of two thin device matrices """

    def _transformed(self, x):
        """ the inputs as the kernel's core sees them: (x - loc) / scale on every leaf field (``_crosskernel.py:172-185``) """
        x = numpy.asarray(self._select(x)) if hasattr(self, '_select') else numpy.asarray(x)
        if self._loc is None and self._scale is None:
            return x
        def tr(v):
            if self._loc is not None:
                v = v - self._loc
            if self._scale is not None:
                v = v / self._scale
            return v
        if x.dtype.names is None:
            return tr(x)
        out = numpy.empty(x.shape, [(nm, numpy.result_type(x.dtype[nm].base, float), x.dtype[nm].shape) for nm in x.dtype.names])
        for nm in x.dtype.names:
            if x.dtype[nm].names is not None:
                raise NotImplementedError('nested fields')
            out[nm] = tr(x[nm])
        return out

    def _factors(self, xs, ys):  # pragma: no cover
        """ (U, V): K = U V^T, U (n, c), V (m, c) host arrays """
        raise NotImplementedError

    def _result_dtype(self, x, y):
        return numpy.dtype(numpy.float64)

    def _finish(self, M):
        return _Composite._finish(self, M)

    def prepare(self, xs, *, ctx=None, dtype=None):
        xs = numpy.asarray(xs).reshape(-1)
        return _PointBundle([xs], len(xs), dtype or numpy.float64)

    def gram(self, xs, ys, *, ctx=None, out=None, deriv=(0, 0)):
        if any(numpy.any(numpy.asarray(o) != 0) for o in deriv):
            raise NotImplementedError(f'derivatives of {type(self).__name__}')
        ctx = ctx or Context.default()
        xs = xs.parts[0] if isinstance(xs, _PointBundle) else xs
        ys = ys.parts[0] if isinstance(ys, _PointBundle) else ys
        U, V = self._factors(numpy.asarray(xs).reshape(-1), numpy.asarray(ys).reshape(-1))
        n, m, c = U.shape[0], V.shape[0], U.shape[1]
        if out is None:
            out = DeviceArray.empty((n, m), numpy.float64, ctx)
        if n and m:
            Ud = DeviceArray.from_host(numpy.ascontiguousarray(U, float), ctx=ctx)
            Vd = DeviceArray.from_host(numpy.ascontiguousarray(V, float), ctx=ctx)
            f, F = _lib.fn('lgp_gemm', numpy.float64)
            _lib.check(f(ctx.handle, 0, 1, n, m, c, F(1.), Ud.ptr, Ud.ld, Vd.ptr, Vd.ld, F(0.), out.ptr, out.ld), 'lgp_gemm')
            self._finish(out)
        return out

    def pairs(self, xs, ys, *, ctx=None):
        ctx = ctx or Context.default()
        U, V = self._factors(numpy.asarray(xs).reshape(-1), numpy.asarray(ys).reshape(-1))
        out = DeviceArray.from_host(numpy.ascontiguousarray(numpy.sum(U * V, axis=1), float), ctx=ctx)
        return self._finish(out) if out.size else out


class Rescaling(_HostFactors):
    """ Outer product kernel, k(x, y) = stdfun(x) stdfun(y) (``_kernels/_basic.py:231-252``): multiply another kernel by it to
    modulate its variance.  ``stdfun`` takes the (transformed) points and returns one number per point; it is evaluated on
    the host """

    def __init__(self, *, stdfun=None, scale=None, loc=None, dim=None):
        super().__init__(scale=scale, loc=loc, dim=dim)
        self._stdfun = stdfun

    def _std(self, x):
        x = self._transformed(x)
        if self._stdfun is None:
            return numpy.ones(x.shape)
        s = numpy.asarray(self._stdfun(x), float)
        if s.shape != x.shape:
            raise ValueError(f'stdfun returned shape {s.shape} for points of shape {x.shape}')
        return s

    def _factors(self, xs, ys):
        return self._std(xs)[:, None], self._std(ys)[:, None]


class Gibbs(_Isotropic):
    """ Gibbs kernel, k(x, y) = sqrt(2 s(x) s(y) / (s(x)² + s(y)²)) exp(-|x - y|² / (s(x)² + s(y)²)), s = ``scalefun``
    (``_kernels/_basic.py:156-185``): a Gaussian kernel whose scale changes from point to point.  ``scalefun`` takes the
    (transformed) points and returns a positive number per point; it is evaluated on the host and travels to the device as one
    more coordinate row, so the input transformation (``loc``, ``scale``: they act BEFORE ``scalefun``, as in the reference) is
    applied on the host too -- the same two IEEE operations """
    _kind = _lib.LGP_KERNEL_GIBBS

    def __init__(self, *, scalefun=None, scale=None, loc=None, dim=None):
        super().__init__(scale=scale, loc=loc, dim=dim)
        self._hloc, self._hscale = self._loc, self._scale
        self._loc = self._scale = None                   # (nothing left for the device to transform)
        self._scalefun = scalefun

    def _rows(self, xs, dtype=None):
        """ the SoA rows of a point set: transformed coordinates, then scalefun of the transformed points """
        x = numpy.asarray(self._select(xs))
        helper = _HostFactors.__new__(_HostFactors)
        helper._dim = None
        helper._loc, helper._scale = self._hloc, self._hscale
        xt = _HostFactors._transformed(helper, x.reshape(-1))
        cols = _leaf_columns(xt)
        s = numpy.ones(xt.shape) if self._scalefun is None else numpy.asarray(self._scalefun(xt), float)
        if s.shape != xt.shape:
            raise ValueError(f'scalefun returned shape {s.shape} for points of shape {xt.shape}')
        assert numpy.all(s > 0), 'scalefun must be positive'
        dtype = dtype or _float_dtype_of(cols, self._hscale, self._hloc)
        return numpy.stack([numpy.asarray(c, dtype) for c in cols] + [numpy.asarray(s, dtype)]), numpy.dtype(dtype)

    def prepare(self, xs, *, ctx=None, dtype=None):
        ctx = ctx or Context.default()
        rows, dtype = self._rows(xs, dtype)
        return DevicePoints('coords', DeviceArray.from_host(rows, ctx=ctx), dtype, ctx)

    def _result_dtype(self, x, y):
        x, y = self._select(x), self._select(y)
        return _float_dtype_of(_leaf_columns(x.reshape(-1)[:0]) + _leaf_columns(y.reshape(-1)[:0]), self._hscale, self._hloc)

    def _prepare(self, xs, ys, ctx):
        if isinstance(xs, DevicePoints) or isinstance(ys, DevicePoints):
            return super()._prepare(xs, ys, ctx)
        ctx = ctx or Context.default()
        rx, dtx = self._rows(xs)
        same = xs is ys
        ry, dty = (rx, dtx) if same else self._rows(ys)
        dtype = numpy.result_type(dtx, dty)
        if rx.shape[0] != ry.shape[0]:
            raise ValueError(f'x has {rx.shape[0] - 1} leaf fields, y has {ry.shape[0] - 1}')
        xd = DeviceArray.from_host(rx.astype(dtype), ctx=ctx)
        yd = xd if same else DeviceArray.from_host(ry.astype(dtype), ctx=ctx)
        return xd, yd, dtype, ctx, rx.shape[0]

    def _diff_call(self, deriv, d):
        if any(numpy.any(numpy.asarray(o) != 0) for o in deriv):
            raise NotImplementedError('derivatives of Gibbs')
        return self._kind, 0, (), self._outops


class _NNJet(_Isotropic):
    """ the derivative core of `NNKernel` (``LGP_KERNEL_NNKERNEL``, only through ``lgp_gram_deriv``): derivable any number of times
    (``_kernels/_basic.py:122``) """
    _kind = _lib.LGP_KERNEL_NNKERNEL
    _derivable = float('inf')

    def __init__(self, q, *, scale=None, loc=None, dim=None):
        super().__init__(scale=scale, loc=loc, dim=dim)
        self._par = (float(q),)

    def _has_closed_form_diff(self, d):
        return False


class NNKernel(Kernel):
    """ Neural network kernel, k(x, y) = 2/π asin(2 (q + x·y) / ((1 + 2 (q + x·x)) (1 + 2 (q + y·y)))), q = sigma0²
    (``_kernels/_basic.py:122-154``): the argument is `Linear` + q times an outer product of per-point factors, a composite of
    this package's own kernels, and the arcsine is applied to the matrix on the device """

    def __init__(self, *, sigma0=1, scale=None, loc=None, dim=None):
        assert 0 < sigma0 < numpy.inf, sigma0
        super().__init__()
        q = float(sigma0) ** 2

        def inv_norm(x):
            if x.dtype.names is None:
                xx = x * x
            else:
                xx = sum(numpy.sum(numpy.square(x[nm]), axis=tuple(range(x.ndim, x[nm].ndim))) for nm in x.dtype.names)
            return 1 / (1 + 2 * (q + xx))
        self._inner = (Linear(scale=scale, loc=loc, dim=dim) + q) * Rescaling(stdfun=inv_norm, scale=scale, loc=loc, dim=dim) * 2.
        self._jet = _NNJet(q, scale=scale, loc=loc, dim=dim)          # derivative observations: the whole expression in Taylor arithmetic

    def _result_dtype(self, x, y):
        return numpy.dtype(numpy.float64)

    def _outer(self, M):
        m2 = M.as2d()
        f, F = _lib.fn('lgp_unary', M.dtype)
        _lib.check(f(M.ctx.handle, m2.shape[0], m2.shape[1], M.ptr, m2.ld, 0, F(2 / numpy.pi), F(0.)), 'lgp_unary')
        return _Composite._finish(self, M)

    def prepare(self, xs, *, ctx=None, dtype=None):
        return self._inner.prepare(xs, ctx=ctx, dtype=dtype)

    def gram(self, xs, ys, *, ctx=None, out=None, deriv=(0, 0)):
        if any(numpy.any(numpy.asarray(o) != 0) for o in deriv):
            jet = self._jet._copy(_outops=tuple(self._outops), _doutops=tuple({} for _ in self._outops))     # (scalar factors of this kernel multiply the derivative)
            return jet.gram(xs, ys, ctx=ctx, out=out, deriv=deriv)
        return self._outer(self._inner.gram(xs, ys, ctx=ctx, out=out))

    def pairs(self, xs, ys, *, ctx=None):
        return self._outer(self._inner.pairs(xs, ys, ctx=ctx))


class Categorical(_HostFactors):
    """ Categorical kernel, k(x, y) = cov[x, y] over the integers 0 .. N - 1 (``_kernels/_basic.py:210-229``) """

    def __init__(self, *, cov=None, scale=None, loc=None, dim=None):
        super().__init__(scale=scale, loc=loc, dim=dim)
        cov = numpy.asarray(cov)
        assert cov.ndim == 2 and cov.shape[0] == cov.shape[1], cov.shape
        assert numpy.allclose(cov, cov.T)
        self._cov = cov.astype(float)

    def _index(self, x):
        x = numpy.asarray(self._select(x))
        if x.dtype.names is not None:
            if len(x.dtype.names) > 1:
                raise ValueError(f'kernel called on type with dimensionality {len(x.dtype.names)} > maxdim=1')
            x = x[x.dtype.names[0]]
        assert numpy.issubdtype(x.dtype, numpy.integer), x.dtype
        if x.size and (x.min() < -len(self._cov) or x.max() >= len(self._cov)):
            raise IndexError(f'category outside 0 .. {len(self._cov) - 1}')
        return x

    def _factors(self, xs, ys):
        ix, iy = self._index(xs), self._index(ys)
        onehot = numpy.zeros((len(iy), len(self._cov)))
        onehot[numpy.arange(len(iy)), iy] = 1.
        return self._cov[ix], onehot


# ------------------------------------------------------------------------------
# BART

# What the reference's pattern  r'\s|[!«»"“”‘’/()\'?¡¿„‚<>,;.:-–—]'  (``_kernels/_basic.py:270``) matches, written out: white space, the
# listed punctuation -- and, because ':-–' inside a character class is a RANGE, every character from ':' (U+003A) to the en
# dash (U+2013), which contains all letters.  In effect only digits and a few symbols (# $ % & * +) make words.  Kept as it
# is: a drop-in gives the reference's numbers.
_BOW_DELIMITERS = re.compile('\\s|[!«»"“”‘’/()\'?¡¿„‚<>,;.—]|[\u003a-\u2013]')


class BagOfWords(_HostFactors):
    """ Bag of words kernel, k(x, y) = Σ_w c_w(x) c_w(y) with c_w(x) the number of times word w appears in the string x
    (``_kernels/_basic.py:272-296``).  The counts are taken on the host; the Gram matrix is the product of the two count
    matrices on the device.  Words are what is left between the reference's delimiters -- see `_BOW_DELIMITERS` """

    def __init__(self, *, scale=None, loc=None, dim=None):
        if scale is not None or loc is not None:
            raise TypeError('BagOfWords: the inputs are strings, scale and loc do not apply')
        super().__init__(dim=dim)

    @staticmethod
    def _bags(strings):
        return [collections.Counter(w for w in _BOW_DELIMITERS.split(str(s)) if w != '') for s in strings]

    def _factors(self, xs, ys):
        xs, ys = self._transformed(xs), self._transformed(ys)
        bx, by = self._bags(xs), self._bags(ys)
        words = sorted(set().union(*bx) & set().union(*by)) if bx and by else []
        col = {w: i for i, w in enumerate(words)}
        U = numpy.zeros((len(bx), max(len(words), 1)))
        V = numpy.zeros((len(by), max(len(words), 1)))
        for M, bags in ((U, bx), (V, by)):
            for i, bag in enumerate(bags):
                for w, c in bag.items():
                    if w in col:
                        M[i, col[w]] = c
        return U, V


def _bart_brackets(npnt, reset, altinput=True, debug=False):
    """
    Cut the depth range [0, npnt - 1] at the `reset` depths into recursion slices ``(top, bottom, repeat)``
    (``_kernels/_bart.py:415-431``).  On the closed-form path (``altinput`` and not ``debug``) a slice at most two levels
    deep that is as deep as each of the ``repeat`` pieces of the slice before it is folded into that slice, which is then
    iterated once more.
    """
    inner = [] if reset is None else [int(v) if int(v) == v else v for v in numpy.atleast_1d(reset).tolist()]
    cuts = [0, *inner, npnt - 1]
    for above, below in zip(cuts, cuts[1:]):
        assert int(below) == below and above <= below, (above, below)
    foldable = altinput and not debug
    slices = []
    for top, bottom in zip(cuts, cuts[1:]):
        depth = bottom - top
        if slices and foldable and depth <= 2:
            first, last, repeat = slices[-1]
            if last - first == repeat * depth:
                slices[-1] = (first, bottom, repeat + 1)
                continue
        slices.append((top, bottom, 1))
    return slices


def _bart_probs(pnt, t, b, repeat):
    """ ``_kernels/_bart.py:436-447``: probabilities of one slice, laid out for iteration """
    probs = numpy.array(pnt[t:b + 1])
    if t > 0:
        probs[0] = 1
    if repeat > 1:
        head = probs[0:1]
        one = numpy.ones_like(head)
        pieces = [[head if i == 0 else one, p] for i, p in enumerate(numpy.split(probs[1:], repeat))]
        probs = numpy.concatenate(sum(reversed(pieces), start=[]))
    return probs


class BART(Kernel):
    """
    BART kernel (``_kernels/_bart.py:32-202``): prior correlation of the
    infinite-trees limit of BART, depth-truncated, with the `reset` slicing.
    Slices of at most 3 probabilities (depth <= 2 per slice: ``maxd <= 2`` or
    ``reset`` depths at most 2 apart -- the configurations the reference
    recommends, ``maxd=4, reset=2`` and ``maxd=10, reset=[2,4,6,8]``,
    ``_bart.py:47-49``) are closed forms; deeper slices (up to depth 10, at most
    16 covariates) recurse over the splits on the device too, much more slowly
    (``_bart.py:759-806``).
    """

    def __init__(self, *, alpha=0.95, beta=2, maxd=2, gamma=1, splits=None, pnt=None,
                 intercept=True, weights=None, reset=None, indices=False, scale=None, loc=None, dim=None):
        super().__init__(scale=scale, loc=loc, dim=dim)
        if scale is not None or loc is not None:
            if indices:
                raise ValueError('scale/loc make no sense with indices=True')
        alpha, self._dalpha = _dual.split(alpha)                     # (tangents of the hyperparameters: `gram_tangents`)
        beta, self._dbeta = _dual.split(beta)
        self._splits = self._check_splits(splits, indices)
        self._indices = bool(indices)
        self._dpnt_ab = None
        self._vector = None
        if ((pnt is None and (numpy.ndim(alpha) or numpy.ndim(beta))) or (pnt is not None and numpy.ndim(pnt) > 1)
                or (not isinstance(gamma, str) and numpy.ndim(gamma))):
            # alpha / beta / gamma / pnt with leading axes of their own: in the reference they broadcast against the shape of the
            # kernel's output (_bart.py:301-303 "fully vectorized", :373-401).  The device kernels take one set of parameters per
            # launch: the pairs are grouped by distinct parameters (`correlation`), a scalar-parameter kernel per group
            if self._dalpha or self._dbeta:
                raise NotImplementedError('tangents of vectorised alpha / beta')
            first = lambda v: v if not numpy.ndim(v) else numpy.asarray(v).reshape(-1)[0]
            if numpy.size(alpha) == 0 or numpy.size(beta) == 0 or (pnt is not None and numpy.size(pnt) == 0):
                raise ValueError('empty parameter array')
            self._vector = dict(alpha=alpha, beta=beta, gamma=gamma, maxd=maxd, pnt=None if pnt is None else numpy.asarray(pnt),
                                intercept=intercept, weights=weights, reset=reset)
            # the scalar-parameter kernel of the first parameter set: argument checks, float type, index conversion
            proto = BART(alpha=first(alpha), beta=first(beta), maxd=maxd, gamma=gamma if isinstance(gamma, str) else first(gamma), splits=splits,
                         pnt=None if pnt is None else numpy.asarray(pnt).reshape(-1, numpy.shape(pnt)[-1])[0], intercept=intercept, weights=weights,
                         reset=reset, indices=indices)
            for name in ('_flt', '_pnt', '_gamma0', '_weights', '_brackets', '_deep'):
                setattr(self, name, getattr(proto, name))
            return
        # splitting probabilities, _bart.py:373-384
        if pnt is None:
            assert maxd == int(maxd) and maxd >= 0, maxd
            # JAX promotion lattice (x64): Python scalars are weak, the int64 depths do not widen float32
            fl = [numpy.asarray(v).dtype for v in (alpha, beta)
                  if not (isinstance(v, (bool, int, float)) and not isinstance(v, numpy.generic))
                  and numpy.issubdtype(numpy.asarray(v).dtype, numpy.floating)]
            ft = numpy.result_type(*fl) if fl else numpy.dtype(numpy.float64)
            alpha_a = numpy.asarray(alpha)
            beta_a = numpy.asarray(beta)
            assert 0 <= alpha_a <= 1, 'alpha must be in [0, 1]'
            assert beta_a >= 0, 'beta must be in [0, inf)'
            d = numpy.arange(int(maxd) + 1)
            pnt = alpha_a.astype(ft) / (1 + d).astype(ft) ** beta_a.astype(ft)
            # tangents of the probabilities with respect to alpha and beta, for `gram_jvp`
            self._dpnt_ab = (1 / (1 + d).astype(ft) ** beta_a.astype(ft), -pnt * numpy.log((1 + d).astype(ft)))
        else:
            pnt = numpy.asarray(pnt)
        p = len(self._splits[0])
        if weights is None:
            weights = numpy.ones(p, pnt.dtype)                      # :387-390
        else:
            weights = numpy.asarray(weights)
            if weights.shape != (p,):
                raise ValueError(f'weights must have shape ({p},)')
        if isinstance(gamma, str):                                   # :393-401
            if gamma != 'auto':
                raise KeyError(gamma)
            assert reset is None and 1 <= pnt.shape[-1] - 1 <= 3
            gamma = self._gamma(p, pnt)
        gamma_weak = isinstance(gamma, (bool, int, float)) and not isinstance(gamma, numpy.generic)
        gamma = numpy.asarray(gamma)
        assert 0 <= gamma <= 1, 'gamma must be in [0, 1]'
        assert numpy.all((0 <= pnt) & (pnt <= 1)), 'pnt must be in [0, 1]'
        assert numpy.all(weights >= 0), 'weights must be in [0, inf)'
        if not intercept:                                            # :411-412
            pnt = pnt.copy()
            pnt[0] = 1
            if self._dpnt_ab is not None:
                self._dpnt_ab = tuple(numpy.concatenate([[0], t[1:]]) for t in self._dpnt_ab)
        # float type of (pnt, gamma, w): fp32 iff all three are fp32 (:657, tests/kernels/test_bart.py:313-330)
        # JAX promotion lattice: integer arrays and weak Python scalars never widen a float type
        fl = [d for d in ([pnt.dtype, weights.dtype] + ([] if gamma_weak else [gamma.dtype])) if numpy.issubdtype(d, numpy.floating)]
        t = numpy.result_type(*fl) if fl else numpy.dtype(numpy.float64)
        self._flt = numpy.dtype(numpy.float32) if t == numpy.float32 else numpy.dtype(numpy.float64)
        self._pnt = pnt.astype(self._flt)
        self._gamma0 = self._flt.type(gamma)
        self._weights = weights.astype(self._flt)
        self._brackets = _bart_brackets(len(pnt), reset)
        # slices of at most three probabilities are closed forms (_bart.py:685-757); deeper ones recurse over the splits
        # (:759-806) on the device as well, one thread per pair with an explicit stack: the slow path, as in the reference
        self._deep = any(rep == 1 and b_ - t_ + 1 > 3 for t_, b_, rep in self._brackets)
        for t_, b_, rep in self._brackets:
            if rep == 1 and b_ - t_ + 1 > 11:
                raise NotImplementedError(f'recursion slice of depth {b_ - t_} > 10: use `reset`')
        if self._deep and p > 16:
            raise NotImplementedError('recursion slices deeper than two levels take at most 16 covariates on the device: use `reset`')

    # ---- classmethods kept from the reference API ------------------------------

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
        if splits is None:
            raise ValueError('splits is required')
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
        """ Generate splitting points from data (``_bart.py:209-259``): per
        dimension, midpoints between consecutive distinct values; host
        sort/unique (O(N log N), not on the O(N^2) path). """
        x = cls._check_x(x)
        x = x.reshape(-1, x.shape[-1]) if x.size else x.reshape(1, x.shape[-1])
        if numpy.issubdtype(x.dtype, numpy.inexact):
            fill = numpy.finfo(x.dtype).max
        else:
            fill = numpy.iinfo(x.dtype).max
        nrow = x.shape[0]
        length = numpy.empty(x.shape[1], numpy.int64)
        midpoints = numpy.empty((max(nrow - 1, 0), x.shape[1]), numpy.result_type(x.dtype, numpy.float32) if not numpy.issubdtype(x.dtype, numpy.inexact) else x.dtype)
        for j in range(x.shape[1]):
            u = numpy.unique(x[:, j])
            u = numpy.concatenate([u, numpy.full(nrow - u.size, fill, u.dtype)])
            with numpy.errstate(over='ignore'):
                m = numpy.where(u[1:] < fill, (u[1:] + u[:-1]) / 2, fill)
            length[j] = numpy.searchsorted(m, fill)
            midpoints[:, j] = m
        return length, midpoints

    @classmethod
    def indices_from_coord(cls, x, splits):
        """ Convert coordinates to bin indices w.r.t. splitting points (``_bart.py:261-299``) """
        splits = cls._check_splits(splits, False)
        return cls._indices_from_coord(x, splits)

    @classmethod
    def _indices_from_coord(cls, x, checked_splits):
        x = cls._check_x(x)
        if x.shape[-1] != checked_splits[0].size:
            raise ValueError(f'splitting grid is for {checked_splits[0].size} dimensions, found {x.shape[-1]}')
        A = checked_splits[1]
        out = numpy.empty(x.shape, numpy.int32)
        for j in range(A.shape[1]):
            out[..., j] = numpy.searchsorted(A[:, j], x[..., j], side='left')     # _bart.py:511-514
        return out

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
    def correlation(cls, splitsbefore_or_totalsplits, splitsbetween_or_index1, splitsafter_or_index2, *,
        alpha=0.95, beta=2, gamma=1, maxd=2, debug=False, pnt=None, intercept=True,
        weights=None, reset=None, altinput=False):
        """
        Compute the BART prior correlation between two points
        (``_bart.py:301-455``), vectorised over leading axes of the three
        integer inputs.  With ``altinput=False`` the counts
        (before, between, after) are converted to (total, index1, index2).
        """
        a0 = numpy.asarray(splitsbefore_or_totalsplits)
        a1 = numpy.asarray(splitsbetween_or_index1)
        a2 = numpy.asarray(splitsafter_or_index2)
        for a in (a0, a1, a2):
            assert numpy.issubdtype(a.dtype, numpy.integer)
        # alpha / beta / gamma (or pnt) with leading axes of their own (``_bart.py:373-401``: they broadcast against the leading axes
        # of the counts): the device kernels take one set of parameters per launch, so the pairs are grouped by distinct parameters
        if ((pnt is None and (numpy.ndim(alpha) or numpy.ndim(beta))) or (pnt is not None and numpy.ndim(pnt) > 1)
                or (not isinstance(gamma, str) and numpy.ndim(gamma))):
            if pnt is None:
                al, be = numpy.asarray(alpha), numpy.asarray(beta)
                plead = numpy.broadcast_shapes(al.shape, be.shape)
                cols = [(numpy.broadcast_to(al, plead), al.dtype if al.ndim else None), (numpy.broadcast_to(be, plead), be.dtype if be.ndim else None)]
            else:
                pn = numpy.asarray(pnt)
                plead = pn.shape[:-1]
                cols = [(pn[..., i], pn.dtype) for i in range(pn.shape[-1])]
            npar = len(cols)
            glead = ()
            if not isinstance(gamma, str):
                ga = numpy.asarray(gamma)
                glead = ga.shape
                cols.append((ga, ga.dtype if ga.ndim else None))
            b0, b1, b2 = numpy.broadcast_arrays(a0, a1, a2)
            full = numpy.broadcast_shapes(b0.shape[:-1], plead, glead)
            nvar = b0.shape[-1]
            flat = [numpy.broadcast_to(b, full + (nvar,)).reshape(-1, nvar) for b in (b0, b1, b2)]
            table = numpy.stack([numpy.broadcast_to(c, full).reshape(-1).astype(numpy.float64) for c, _ in cols], axis=1)
            uniq, inv = numpy.unique(table, axis=0, return_inverse=True)
            inv = inv.reshape(-1)
            # (a parameter given as a scalar keeps its own type, Python scalars stay weak; array values carry the array's type)
            typed = lambda v, dt, orig: orig if dt is None else dt.type(v)
            out = None
            for u, row in enumerate(uniq):
                sel = inv == u
                kw = dict(maxd=maxd, debug=debug, intercept=intercept, weights=weights, reset=reset, altinput=altinput)
                if pnt is None:
                    kw.update(alpha=typed(row[0], cols[0][1], alpha), beta=typed(row[1], cols[1][1], beta))
                else:
                    kw.update(alpha=alpha, beta=beta, pnt=numpy.array(row[:npar], cols[0][1]))
                kw['gamma'] = gamma if isinstance(gamma, str) else typed(row[npar], cols[npar][1], gamma)
                res = cls.correlation(flat[0][sel], flat[1][sel], flat[2][sel], **kw)
                if out is None:
                    out = numpy.empty(len(inv), res.dtype)
                out[sel] = res
            return out.reshape(full)
        assert numpy.all(a0 >= 0), 'splitting counts must be nonnegative'
        if altinput:
            assert numpy.all((0 <= a1) & (a1 <= a0)), 'splitting index must be in [0, n]'
            assert numpy.all((0 <= a2) & (a2 <= a0)), 'splitting index must be in [0, n]'
            n, ix, iy = a0, a1, a2
        else:
            assert numpy.all(a1 >= 0) and numpy.all(a2 >= 0), 'splitting counts must be nonnegative'
            if debug:
                raise NotImplementedError('debug=True (full recursion) is not available on the device')
            # nminus, n0, nplus -> totals and bin indices: ix = nminus, iy = nminus + n0
            n, ix, iy = a0 + a1 + a2, a0, a0 + a1
        if debug:
            raise NotImplementedError('debug=True (full recursion) is not available on the device')
        n, ix, iy = numpy.broadcast_arrays(n, ix, iy)
        p = n.shape[-1]
        kern = cls(alpha=alpha, beta=beta, maxd=maxd, gamma=gamma, splits=(numpy.zeros(p, numpy.int64), None),
                   pnt=pnt, intercept=intercept, weights=weights, reset=reset, indices=True)
        shape = n.shape[:-1]
        npairs = int(numpy.prod(shape, dtype=int))
        out = kern._pairs_indices(ix.reshape(npairs, p), iy.reshape(npairs, p), npair=n.reshape(npairs, p))
        return out.to_host().reshape(shape)

    # ---- evaluation ---------------------------------------------------------

    def _result_dtype(self, x, y):
        return self._flt

    def _to_indices(self, xs):
        xs = numpy.asarray(self._select(xs))
        if not xs.dtype.names:
            xs = xs[..., None]
        if self._indices:
            ix = self._check_x(xs)
            if not numpy.issubdtype(ix.dtype, numpy.integer):
                raise TypeError('indices=True requires integer inputs')
        else:
            ix = self._check_x(xs)
            if self._loc is not None:
                ix = ix - self._loc
            if self._scale is not None:
                ix = ix / self._scale
            ix = self._indices_from_coord(ix, self._splits)
        ix = numpy.ascontiguousarray(ix.reshape(-1, ix.shape[-1]), numpy.int32)
        if ix.shape[1] != len(self._splits[0]):
            raise ValueError(f'splitting grid is for {len(self._splits[0])} dimensions, found {ix.shape[1]}')
        return ix

    def _psi_table(self, maxn, ctx):
        memo = self.__dict__.setdefault('_psi_memo', {})
        key = (maxn, id(ctx))
        if key not in memo:
            k = numpy.arange(maxn + 2, dtype=float)
            k[0] = 1
            memo[key] = DeviceArray.from_host(_special.digamma(k).astype(self._flt), ctx=ctx)
        return memo[key]

    def _device_indices(self, xs, ctx):
        """ bin indices of a host point set on the device.  The last few point sets are remembered (by buffer address, layout and a
        strided sample of the bytes, so that an array modified in place is noticed): a model that rebuilds its GP at every
        hyperparameter step on the same points -- `bayestree.bart`, the C5 loop -- neither searches the splitting grid nor uploads
        the indices again """
        if isinstance(xs, DevicePoints):
            return xs.array
        a = numpy.asarray(xs)
        raw = a.view(numpy.uint8).reshape(-1) if a.flags.c_contiguous and a.size else None
        key = None
        if raw is not None:
            step = max(1, raw.size // 4096)
            key = (a.__array_interface__['data'][0], a.shape, str(a.dtype), id(ctx), raw[::step][:4096].tobytes())
            memo = self.__dict__.setdefault('_index_memo', [])
            for k, d in memo:
                if k == key:
                    return d
        d = DeviceArray.from_host(self._to_indices(xs), ctx=ctx)
        if key is not None:
            memo.append((key, d))
            del memo[:-4]
        return d

    def _run(self, name, ctx, p, ixd, iyd, n, m, nsplits, npair, out, ld):
        f, F = _lib.fn(name, self._flt)
        pairs = name.startswith('lgp_pairs')
        nsp = None if nsplits is None else (ctypes.c_int32 * max(p, 1))(*[int(v) for v in nsplits])
        w = (F * max(p, 1))(*[float(v) for v in self._weights])
        maxn = int(numpy.max(nsplits)) if nsplits is not None and p else 0
        if npair is not None and npair.size:
            maxn = max(maxn, int(npair.max()))
        psi = self._psi_table(maxn, ctx)
        npd = None
        if npair is not None:
            npd = DeviceArray.from_host(numpy.ascontiguousarray(npair, numpy.int32), ctx=ctx)
        gamma_ptr = None
        first = True
        for t, b, repeat in reversed(self._brackets):                # :434-455
            probs = _bart_probs(self._pnt, t, b, repeat)
            size = len(probs) // repeat
            pr = (F * len(probs))(*[float(v) for v in probs])
            last = (t, b, repeat) == self._brackets[0]
            nops, codes, vals = _ops_arrays(self._outops if last else (), F)
            gptr = None if first else out.ptr
            if pairs:
                _lib.check(f(ctx.handle, p, ixd.ptr, iyd.ptr, n, nsp, None if npd is None else npd.ptr, w, size, repeat, pr,
                             F(float(self._gamma0)), gptr, psi.ptr, psi.shape[0], nops, codes, vals, out.ptr), name)
            else:
                _lib.check(f(ctx.handle, p, ixd.ptr, n, iyd.ptr, m, nsp, w, size, repeat, pr,
                             F(float(self._gamma0)), gptr, psi.ptr, psi.shape[0], nops, codes, vals, out.ptr, ld), name)
            first = False
        return out

    def prepare(self, xs, *, ctx=None, dtype=None):
        """ bin indices of a flat point set, uploaded once: `DevicePoints` accepted by `gram` """
        ctx = ctx or Context.default()
        return DevicePoints('indices', DeviceArray.from_host(self._to_indices(xs), ctx=ctx), self._flt, ctx)

    def _host_indices(self, xs):
        return xs.array.to_host() if isinstance(xs, DevicePoints) else self._to_indices(xs)

    def _vector_eval(self, ix, iy, shape, ctx, out=None):
        """ vectorised parameters: `correlation` on the bin indices (it groups the pairs by distinct parameters and runs the scalar
        kernels on the device), the parameters' leading axes broadcast against `shape` like in the reference, where they meet
        the shape of the kernel's output (numpy rules: trailing axes first) """
        v = self._vector
        nsp = numpy.asarray(self._splits[0]).reshape((1,) * (ix.ndim - 1) + (-1,))
        res = self.correlation(nsp, ix, iy, alpha=v['alpha'], beta=v['beta'], gamma=v['gamma'], maxd=v['maxd'], pnt=v['pnt'],
                               intercept=v['intercept'], weights=v['weights'], reset=v['reset'], altinput=True)
        res = numpy.asarray(res)
        if res.shape != tuple(shape):
            raise NotImplementedError(f'vectorised parameters broadcast the kernel output to {res.shape}: only parameter shapes that '
                                      f'broadcast INTO the output shape {tuple(shape)} are evaluated here')
        for op, c in self._outops:
            res = res * res.dtype.type(c) if op == 'mul' else res + res.dtype.type(c)
        dev = DeviceArray.from_host(numpy.ascontiguousarray(res, self._flt), ctx=ctx)
        if out is None:
            return dev
        assert out.shape == dev.shape and out.dtype == dev.dtype
        out.assign(dev)
        return out

    def gram(self, xs, ys, *, ctx=None, out=None):
        ctx = ctx or Context.default()
        if self._vector is not None:
            ix, iy = self._host_indices(xs), self._host_indices(ys)
            if ix.shape[1] != len(self._splits[0]) or iy.shape[1] != len(self._splits[0]):
                raise ValueError(f'splitting grid is for {len(self._splits[0])} dimensions, found {ix.shape[1]}')
            return self._vector_eval(ix[:, None, :], iy[None, :, :], (ix.shape[0], iy.shape[0]), ctx, out)
        same = ys is xs
        ixd = self._device_indices(xs, ctx)
        iyd = ixd if same else self._device_indices(ys, ctx)
        if not (ixd.contiguous and iyd.contiguous):
            raise ValueError('index arrays must be contiguous row blocks')
        n, m, p = ixd.shape[0], iyd.shape[0], ixd.shape[1]
        if p != len(self._splits[0]):
            raise ValueError(f'splitting grid is for {len(self._splits[0])} dimensions, found {p}')
        if out is None:
            out = DeviceArray.empty((n, m), self._flt, ctx)
        assert out.shape == (n, m) and out.dtype == self._flt
        if n == 0 or m == 0:
            return out
        return self._run('lgp_gram_bart', ctx, p, ixd, iyd, n, m, self._splits[0], None, out, out.ld)

    def gram_jvp(self, xs, ys, dpnt=None, *, ctx=None):
        """
        The Gram matrix and its forward-mode tangents: ``dpnt`` is a sequence of at most two arrays like ``pnt``
        (derivatives of the splitting probabilities with respect to a parameter); default: the derivatives with
        respect to ``alpha`` and ``beta`` of ``pnt_d = alpha / (1 + d)^beta`` (``_bart.py:379-382``).  Returns
        ``(K, [dK_0, dK_1])`` as device matrices -- what ``jax.jacfwd`` of the reference's ``makegp`` yields for the
        BART hyperparameters of ``bayestree.bart`` (``bayestree/_bart.py:187-205``), computed in the same pass as K.
        """
        ctx = ctx or Context.default()
        if self._vector is not None:
            raise NotImplementedError('tangents of a BART kernel with vectorised parameters')
        if dpnt is None:
            if self._dpnt_ab is None:
                raise ValueError('the kernel was built from `pnt`: pass the derivatives of pnt explicitly')
            dpnt = self._dpnt_ab
        dpnt = [numpy.asarray(t, self._flt) for t in dpnt]
        if not 1 <= len(dpnt) <= 2 or any(t.shape != self._pnt.shape for t in dpnt):
            raise ValueError(f'one or two tangent arrays of shape {self._pnt.shape} expected')
        if self._deep:
            raise NotImplementedError('tangents of recursion slices deeper than two levels are not available: use `reset`')
        ix = self._to_indices(xs)
        iy = ix if ys is xs else self._to_indices(ys)
        n, m, p = ix.shape[0], iy.shape[0], ix.shape[1]
        out = DeviceArray.empty((n, m), self._flt, ctx)
        tans = [DeviceArray.zeros((n, m), self._flt, ctx) for _ in dpnt]
        if n == 0 or m == 0:
            return out, tans
        ixd = DeviceArray.from_host(ix, ctx=ctx)
        iyd = ixd if iy is ix else DeviceArray.from_host(iy, ctx=ctx)
        f, F = _lib.fn('lgp_gram_bart_jvp', self._flt)
        nsplits = self._splits[0]
        nsp = (ctypes.c_int32 * max(p, 1))(*[int(v) for v in nsplits])
        w = (F * max(p, 1))(*[float(v) for v in self._weights])
        psi = self._psi_table(int(numpy.max(nsplits)) if p else 0, ctx)
        ntan = len(dpnt)
        tptr = (ctypes.c_void_p * ntan)(*[t.ptr for t in tans])
        first = True
        for t, b, repeat in reversed(self._brackets):                # :434-455, tangents chained like the values
            probs = _bart_probs(self._pnt, t, b, repeat)
            size = len(probs) // repeat
            dprobs = []
            for tan in dpnt:
                # same layout as the probabilities; the entries that `_bart_probs` sets to the constant 1 have zero derivative
                const = _bart_probs(numpy.full_like(self._pnt, 2), t, b, repeat) == 1
                dprobs.append(numpy.where(const, 0, _bart_probs(tan, t, b, repeat)))
            pr = (F * len(probs))(*[float(v) for v in probs])
            dpr = (F * (ntan * len(probs)))(*[float(v) for dp in dprobs for v in dp])
            last = (t, b, repeat) == self._brackets[0]
            nops, codes, vals = _ops_arrays(self._outops if last else (), F)
            _lib.check(f(ctx.handle, p, ixd.ptr, n, iyd.ptr, m, nsp, w, size, repeat, pr, F(float(self._gamma0)),
                         None if first else out.ptr, psi.ptr, psi.shape[0], nops, codes, vals, out.ptr, out.ld,
                         ntan, dpr, tptr, None if first else tptr), 'lgp_gram_bart_jvp')
            first = False
        return out, tans

    def tangent_keys(self):
        keys = super().tangent_keys()
        for d in (self._dalpha, self._dbeta):
            for k in d:
                if k not in keys:
                    keys.append(k)
        return keys

    def gram_tangents(self, xs, ys, *, ctx=None):
        """ the Gram matrix and its tangents for ``alpha`` / ``beta`` given as `_dual.Dual` (through `gram_jvp`) """
        if self._dscale or self._dloc or any(self._doutops):
            raise NotImplementedError('tangents of BART with respect to scale, loc or multiplied / added constants')
        if not (self._dalpha or self._dbeta):
            return self.gram(xs, ys, ctx=ctx), {}
        K, (dKa, dKb) = self.gram_jvp(xs, ys, ctx=ctx)
        f, F = _lib.fn('lgp_axpby', K.dtype)
        tangents = {}
        for key in self.tangent_keys():
            T = DeviceArray.empty(K.shape, K.dtype, K.ctx)
            if K.size:
                _lib.check(f(K.ctx.handle, K.shape[0], K.shape[1], F(float(self._dalpha.get(key, 0.))), dKa.ptr, dKa.ld,
                             F(float(self._dbeta.get(key, 0.))), dKb.ptr, dKb.ld, T.ptr, T.ld), 'lgp_axpby')
            tangents[key] = T
        return K, tangents

    def _pairs_indices(self, ix, iy, npair=None, ctx=None):
        ctx = ctx or Context.default()
        n, p = ix.shape
        out = DeviceArray.empty((n,), self._flt, ctx)
        if n == 0:
            return out
        ixd = DeviceArray.from_host(numpy.ascontiguousarray(ix, numpy.int32), ctx=ctx)
        iyd = DeviceArray.from_host(numpy.ascontiguousarray(iy, numpy.int32), ctx=ctx)
        return self._run('lgp_pairs_bart', ctx, p, ixd, iyd, n, n, None if npair is not None else self._splits[0], npair, out, n)

    def pairs(self, xs, ys, *, ctx=None):
        if self._vector is not None:
            ix, iy = self._host_indices(xs), self._host_indices(ys)
            return self._vector_eval(ix, iy, (ix.shape[0],), ctx or Context.default())
        return self._pairs_indices(self._to_indices(xs), self._to_indices(ys), ctx=ctx)

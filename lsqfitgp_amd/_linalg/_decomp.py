"""
Host-side mirror of ``src/lsqfitgp/_linalg/_decomp.py``: the abstract
``Decomposition`` API (``:126-278``) and its Cholesky implementation ``Chol``
(``:376-658``), with every array operation executed by the HIP library
(``include/lsqfitgp_hip.h``) on an MI355X.  Same names, argument meaning and
error behaviour as the reference; inputs may be numpy arrays (uploaded) or
``DeviceArray`` (used in place); results are numpy arrays unless
``device=True`` is passed.
"""

import abc
import ctypes

import numpy

from .. import _lib
from .._device import Context, DeviceArray, asdevice


def _float_dtype(*arrays):
    """ ``_jaxext.float_type`` (``src/lsqfitgp/_jaxext/__init__.py:140-144``):
    float32 stays float32, everything else computes in float64 (x64 mode,
    ``_patch_jax.py:27``) """
    t = numpy.result_type(*[a.dtype for a in arrays])
    return numpy.dtype(numpy.float32) if t == numpy.float32 else numpy.dtype(numpy.float64)


class Decomposition(abc.ABC):
    """
    Abstract base class for decompositions of positive semidefinite matrices
    (``_decomp.py:126-278``).
    """

    @abc.abstractmethod
    def __init__(self, *args, **kw):  # pragma: no cover
        """ Decompose the input matrix """

    @abc.abstractmethod
    def matrix(self):  # pragma: no cover
        """ The input matrix """

    @abc.abstractmethod
    def ginv_linear(self, X):  # pragma: no cover
        """ Compute K⁻X """

    @abc.abstractmethod
    def pinv_bilinear(self, A, r):  # pragma: no cover
        """ Compute A'K⁺r """

    @abc.abstractmethod
    def pinv_bilinear_robj(self, A, r):  # pragma: no cover
        """ Compute A'K⁺r, where r can be an array of objects """

    @abc.abstractmethod
    def ginv_quad(self, A):  # pragma: no cover
        """ Compute A'K⁻A """

    @abc.abstractmethod
    def ginv_diagquad(self, A):  # pragma: no cover
        """ Compute diag(A'K⁻A) """

    @abc.abstractmethod
    def correlate(self, x):  # pragma: no cover
        """ Compute Zx where K = ZZ' """

    @abc.abstractmethod
    def back_correlate(self, X):  # pragma: no cover
        """ Compute Z'X """

    @abc.abstractmethod
    def pinv_correlate(self, x):  # pragma: no cover
        """ Compute Z⁺x """

    @abc.abstractmethod
    def minus_log_normal_density(self, r, *, dr_vjp=None, dK_vjp=None,
        dr_jvp_vec=None, dK_jvp_vec=None, dr=None, dK=None, value=False,
        gradrev=False, gradfwd=False, fisher=False, fishvec=False):  # pragma: no cover
        """ Minus log of a Normal density with covariance K, and derivatives
        (``_decomp.py:181-243``) """

    @classmethod
    def make_derivs(cls, K_fun, r_fun, primal, *, args=(), kw={}, vec=None, value=False, gradrev=False, gradfwd=False, fisher=False,
                    fishvec=False):
        """
        Prepares the derivative arguments of `minus_log_normal_density` (``_decomp.py:588-652``: same signature, same returns
        ``(K, r, out)`` with ``out`` holding ``dK_vjp`` / ``dr_vjp`` for ``gradrev`` and ``fishvec``, ``dK_jvp_vec`` /
        ``dr_jvp_vec`` for ``fishvec``, ``dK`` (n, n, p) / ``dr`` (n, p) for ``gradfwd`` and ``fisher``).

        The reference gets them from jax (``vjp`` / ``jvp`` / ``jacfwd`` of the traced functions).  Here ``K_fun(primal, *args,
        **kw)`` and ``r_fun`` are evaluated ONCE on a `Dual` primal that carries the p unit tangents (forward mode through the
        Python operators and numpy ufuncs the functions use, ``lsqfitgp_amd/_dual.py``); the Jacobians are formed from the
        tangents, and the vector-Jacobian products are contractions with them -- with p hyperparameters and n^2 outputs that is
        what reverse mode would compute as well.  Functions whose arithmetic `Dual` does not cover (a ``NotImplementedError`` /
        ``TypeError`` / ``ValueError`` on the dual primal, or tangents that disagree with one directional central difference
        because a dual number was turned into a float on the way) are differentiated by central differences with a step of
        6e-6 max(1, |primal_k|): slower and good to ~1e-9 relative instead of to rounding.
        """
        from .._dual import Dual, split
        primal = numpy.asarray(primal, float)
        if primal.ndim != 1:
            raise ValueError(f'primal must be a 1d array, found shape {primal.shape}')
        p = primal.size
        f = lambda fun: (lambda x: fun(x, *args, **kw))
        K_fun, r_fun = f(K_fun), f(r_fun)

        steps = 6e-6 * numpy.maximum(1., numpy.abs(primal))

        def central(fun, direction):
            return (numpy.asarray(fun(primal + steps * direction), float) - numpy.asarray(fun(primal - steps * direction), float)) / 2

        def jacobian(fun):
            """ value and Jacobian (value.shape + (p,)) """
            try:
                out = fun(Dual(primal, {k: numpy.eye(p)[k] for k in range(p)}))
                val, tan = split(out)
                val = numpy.asarray(val, float)
                jac = numpy.zeros(val.shape + (p,))
                for k, t in tan.items():
                    jac[..., k] = numpy.broadcast_to(numpy.asarray(t, float), val.shape)
                # arithmetic that turns a dual number into a float on the way (an assignment into a float array, ``float(x)``)
                # loses its tangent without an error: one directional central difference says whether that happened
                direction = numpy.random.default_rng(0).choice([-1., 1.], p)
                fd = central(fun, direction)
                if not numpy.allclose(jac @ (steps * direction), fd, rtol=1e-4, atol=1e-7 * max(1., float(numpy.max(numpy.abs(val), initial=0.))) * float(steps.max())):
                    raise ValueError('the tangents of the dual evaluation disagree with a central difference')
                return val, jac
            except (NotImplementedError, TypeError, ValueError):
                val = numpy.asarray(fun(primal), float)
                jac = numpy.empty(val.shape + (p,))
                for k in range(p):
                    e = numpy.zeros(p)
                    e[k] = 1.
                    jac[..., k] = central(fun, e) / steps[k]
                return val, jac

        out = {}
        if gradrev or fishvec or gradfwd or fisher:
            K, dK = jacobian(K_fun)
            r, dr = jacobian(r_fun)
        else:
            K, r = numpy.asarray(K_fun(primal), float), numpy.asarray(r_fun(primal), float)
        if gradrev or fishvec:
            out['dK_vjp'] = lambda x: numpy.einsum('ij,ijk->k', numpy.asarray(x), dK)
            out['dr_vjp'] = lambda x: numpy.asarray(x) @ dr
        if fishvec:
            vec = numpy.asarray(vec, float)
            out['dK_jvp_vec'] = dK @ vec
            out['dr_jvp_vec'] = dr @ vec
        if gradfwd or fisher:
            out['dK'] = dK
            out['dr'] = dr
        return K, r, out

    @property
    def eps(self):
        """ The threshold below which eigenvalues are too small to be determined """
        return self._eps

    @property
    @abc.abstractmethod
    def n(self):  # pragma: no cover
        """ Number of rows/columns of the matrix """

    @property
    @abc.abstractmethod
    def m(self):  # pragma: no cover
        """ Number of columns of Z """

    def ginv(self):
        """ Compute K⁻ (``_decomp.py:276-278``) """
        return self.ginv_quad(numpy.eye(self.n, dtype=self._dtype))


def solve_triangular_python(a, b, *, lower=False):
    """
    Triangular solve ``a x = b`` by substitution with plain Python arithmetic,
    so that ``b`` may hold arbitrary objects (same contract as the reference's
    helper, ``_decomp.py:280-309``; host code by nature, used only by
    `pinv_bilinear_robj`).  ``b`` is a vector or a matrix of right-hand sides.
    """
    a = numpy.asarray(a)
    b = numpy.asarray(b)
    n = a.shape[-1]
    if b.shape[0] != n:
        raise ValueError(f'right-hand side has {b.shape[0]} rows, the matrix has {n}')
    cols = b.reshape(n, -1)
    out = numpy.empty(cols.shape, dtype=b.dtype if b.dtype == object else numpy.result_type(a.dtype, b.dtype, float))
    order = range(n) if lower else range(n - 1, -1, -1)
    for c in range(cols.shape[1]):
        done = []                                   # indices already solved, in elimination order
        for i in order:
            acc = cols[i, c]
            for j in done:
                acc = acc - a[i, j] * out[j, c]
            out[i, c] = acc / a[i, i]
            done.append(i)
    return out.reshape(b.shape)


def solve_batched(a, b, *, assume_a='gen', ctx=None, **kw):
    """ ``a^-1 b`` batching matmul-like (``_decomp.py:330-347``): ``a`` (..., n, n), ``b`` (n,) or (..., n, k), the leading axes
    broadcast against each other.  On the device only the case the reference itself uses (``assume_a='pos'``, symmetric positive
    definite ``a``: tests/linalg/test_decomp.py:246,257): every distinct matrix of the batch is factored once, without
    regularisation (`Chol` with eps = 0), and all the right-hand sides that go with it are solved in one call """
    if assume_a not in ('pos', 'positive definite'):
        raise NotImplementedError(f"solve_batched(assume_a={assume_a!r}): only positive definite matrices have a device solver (Cholesky)")
    if kw.get('lower') or kw.get('transposed'):
        raise NotImplementedError('solve_batched: lower= / transposed= are not supported')
    a = numpy.asarray(a)
    b = numpy.asarray(b)
    if a.ndim < 2 or a.shape[-1] != a.shape[-2]:
        raise ValueError(f'solve_batched: square matrices expected, found shape {a.shape}')
    vec = b.ndim < 2
    if vec:
        b = b[:, None]
    n = a.shape[-1]
    if b.shape[-2] != n:
        raise ValueError(f'solve_batched: right-hand side has {b.shape[-2]} rows, the matrices {n}')
    batch = numpy.broadcast_shapes(a.shape[:-2], b.shape[:-2])
    k = b.shape[-1]
    dtype = _float_dtype(a, b)
    bb = numpy.broadcast_to(b, batch + (n, k))
    out = numpy.empty(batch + (n, k), dtype)
    # the batch axes along which `a` really varies (size 1 in a, or missing: one factorisation serves the whole axis)
    apad = a.reshape((1,) * (len(batch) - (a.ndim - 2)) + a.shape)
    for idx in numpy.ndindex(*apad.shape[:-2]):
        sel = tuple(slice(None) if apad.shape[ax] == 1 else i for ax, i in enumerate(idx))
        rhs = bb[sel]                                         # (free axes ..., n, k)
        free = rhs.shape[:-2]
        flat = numpy.moveaxis(rhs.reshape((-1, n, k)), 1, 0).reshape(n, -1)          # n x (free * k)
        dec = Chol(numpy.ascontiguousarray(apad[idx], dtype), epsrel=0, epsabs=0, ctx=ctx)
        sol = numpy.asarray(dec.ginv_linear(numpy.ascontiguousarray(flat, dtype))).reshape(n, -1, k)
        out[sel] = numpy.moveaxis(sol, 0, 1).reshape(free + (n, k))
    return out[..., 0] if vec else out


class Chol(Decomposition):
    """
    Cholesky decomposition. The matrix is regularized adding a small multiple
    of the identity (``_decomp.py:376-393``).  The factor L, the inverses of
    its 128x128 diagonal blocks and the original K stay resident in HBM.
    """

    def __init__(self, K, *, epsrel='auto', epsabs=0, ctx=None, rhs=None):
        """ ``rhs`` (optional, (n, q) host or device matrix): right-hand sides carried through the factorisation;
        ``L^-1 rhs`` is then available from `solved` at no extra cost (the forward solve that ``pinv_bilinear`` /
        ``ginv_quad`` / ``minus_log_normal_density`` would do, done by the MFMA updates of the factorisation) """
        if not isinstance(K, DeviceArray):
            K = numpy.asarray(K)
            if K.ndim != 2 or K.shape[0] != K.shape[1] or K.shape[0] == 0:
                raise ValueError(f'matrix to decompose must be square and nonempty, found shape {K.shape}')
            K = DeviceArray.from_host(K, dtype=_float_dtype(K), ctx=ctx)
        if K.ndim != 2 or K.shape[0] != K.shape[1] or K.shape[0] == 0:
            raise ValueError(f'matrix to decompose must be square and nonempty, found shape {K.shape}')
        if K.dtype.kind != 'f':
            raise TypeError(f'device matrix must be floating point, found {K.dtype}')
        self._K = K                                                    # :383
        self._ctx = K.ctx
        self._dtype = K.dtype
        n = K.shape[0]
        flags = 0
        if isinstance(epsrel, str):
            if epsrel != 'auto':
                raise ValueError(f'epsrel={epsrel!r}')
            flags |= _lib.LGP_EPS_AUTO_REL                             # :248-249
            epsrel = 0.
        if isinstance(epsabs, str):
            if epsabs != 'auto':
                raise ValueError(f'epsabs={epsabs!r}')
            flags |= _lib.LGP_EPS_AUTO_ABS                             # :250-251
            epsabs = 0.
        dblk = DeviceArray.empty((_lib.load().lgp_chol_dblk_size(n),), self._dtype, self._ctx)
        self._solved = None
        self._solved_for = None
        if rhs is None:
            f, F = _lib.fn('lgp_chol_factor', self._dtype)
            L = DeviceArray.empty((n, n), self._dtype, self._ctx)
            eps = F()
            info = _lib.check(f(self._ctx.handle, n, K.ptr, K.ld, F(epsrel), F(epsabs), flags, L.ptr, L.ld, dblk.ptr, ctypes.byref(eps)), 'lgp_chol_factor')
        else:
            B, _ = self._rhs_any(rhs, n)
            q = B.shape[1]
            nx = -(-q // 128) * 128          # whole 128-row tiles: the extra rows take the kernels' aligned fast path
            f, F = _lib.fn('lgp_chol_factor_rhs', self._dtype)
            Lfull = DeviceArray.empty((n + nx, n), self._dtype, self._ctx)
            Lfull.block(n + q, n + nx, 0, n).fill_zero()
            if q:
                ft, _ = _lib.fn('lgp_transpose', self._dtype)
                Bt = Lfull.block(n, n + q, 0, n)
                _lib.check(ft(self._ctx.handle, Bt.ptr, Bt.ld, B.ptr, B.ld, n, q), 'lgp_transpose')
            eps = F()
            info = _lib.check(f(self._ctx.handle, n, K.ptr, K.ld, F(epsrel), F(epsabs), flags, Lfull.ptr, Lfull.ld, nx, dblk.ptr,
                                ctypes.byref(eps)), 'lgp_chol_factor_rhs')
            L = Lfull.block(0, n, 0, n)
            self._solved = Lfull.block(n, n + q, 0, n)                 # (q, n) = (L^-1 rhs)^T
            self._solved_for = rhs
        if info != 0:                                                  # :389-391
            err = numpy.linalg.LinAlgError('cholesky decomposition not finite, probably matrix not pos def numerically')
            err.info = int(info)                                       # LAPACK-style: 1-based index of the first bad pivot
            raise err
        self._L = L
        self._dblk = dblk
        self._eps = self._dtype.type(eps.value)                        # :393

    @classmethod
    def _adopt(cls, L, dblk, eps, *, solved=None, matrix=None, lower_only=False):
        """ a decomposition around a factor that the library has already computed (the fused Gram + factorisation entry point never
        materialises K): ``L`` (n, n) device block, ``dblk`` its diagonal-block workspace, ``solved`` (q, n) the rows carried
        through the factorisation, ``matrix`` a callable that builds K on the device if somebody asks for it; ``lower_only``: the
        factor was made with ``LGP_FACTOR_LOWER_ONLY`` (no zeros above its diagonal: `_full_factor` writes them on first use) """
        self = cls.__new__(cls)
        self._upper_unset = bool(lower_only)
        self._K = None
        self._K_builder = matrix
        self._ctx = L.ctx
        self._dtype = L.dtype
        self._L = L
        self._dblk = dblk
        self._eps = self._dtype.type(eps)
        self._solved = solved
        self._solved_for = True if solved is not None else None
        return self

    def solved(self, *, device=False):
        """ ``L^-1 rhs`` for the ``rhs`` given at construction, shape (n, q) """
        if self._solved is None:
            raise ValueError('the decomposition was built without rhs')
        out = self._solved.transpose()
        return out if device else out.to_host()

    # ---- helpers ---------------------------------------------------------

    def _rhs_any(self, X, n):
        """ (n, q) device matrix of the decomposition's dtype from a host or device operand (no copy of device data) """
        if isinstance(X, DeviceArray):
            dev = X.as2d()
            if dev.dtype != self._dtype:
                raise TypeError(f'device operand has dtype {dev.dtype}, decomposition is {self._dtype}')
        else:
            X = numpy.asarray(X)
            dev = DeviceArray.from_host(X.reshape(len(X), -1), dtype=self._dtype, ctx=self._ctx)
        if dev.shape[0] != n:
            raise ValueError(f'operand has {dev.shape[0]} rows, the decomposed matrix has {n}')
        return dev, None

    def _rhs(self, X, copy):
        """ right-hand side as a device (n, k) matrix; returns (dev, was_vector) """
        if isinstance(X, DeviceArray):
            vec = X.ndim < 2
            dev = X.as2d()
            if dev.dtype != self._dtype:
                raise TypeError(f'device operand has dtype {dev.dtype}, decomposition is {self._dtype}')
            if copy:
                dev = dev.copy()
        else:
            X = numpy.asarray(X)
            if X.dtype == object:
                raise TypeError('object arrays are supported only by pinv_bilinear_robj')
            vec = X.ndim < 2
            if X.ndim > 2:
                raise NotImplementedError('batched right-hand sides (ndim > 2) are not supported')
            dev = DeviceArray.from_host(X.reshape(len(X), -1), dtype=self._dtype, ctx=self._ctx)
        if dev.shape[0] != self.n:
            raise ValueError(f'operand has {dev.shape[0]} rows, the decomposed matrix has {self.n}')
        return dev, vec

    def _out(self, dev, vec, device):
        if device:
            return dev
        a = dev.to_host()
        return a[..., 0] if vec and a.ndim == 2 else a

    def _trsm(self, B, trans):
        """ B <- L^-1 B (trans=0) or L^-T B (trans=1) in place """
        f, _ = _lib.fn('lgp_trsm', self._dtype)
        _lib.check(f(self._ctx.handle, self.n, self._L.ptr, self._L.ld, self._dblk.ptr, int(trans), B.ptr, B.ld, B.shape[1]), 'lgp_trsm')
        return B

    def _gemm(self, ta, tb, A, B, alpha=1., beta=0., C=None):
        f, F = _lib.fn('lgp_gemm', self._dtype)
        m = A.shape[1] if ta else A.shape[0]
        k = A.shape[0] if ta else A.shape[1]
        n = B.shape[0] if tb else B.shape[1]
        kb = B.shape[1] if tb else B.shape[0]
        assert k == kb, (A.shape, B.shape, ta, tb)
        if C is None:
            C = DeviceArray.empty((m, n), self._dtype, self._ctx)
        _lib.check(f(self._ctx.handle, int(ta), int(tb), m, n, k, F(alpha), A.ptr, A.ld, B.ptr, B.ld, F(beta), C.ptr, C.ld), 'lgp_gemm')
        return C

    # ---- Decomposition API -----------------------------------------------

    def matrix(self):
        """ The input matrix (``:395-396``), device resident """
        if self._K is None and getattr(self, '_K_builder', None) is not None:
            self._K = self._K_builder()
        return self._K

    def _full_factor(self):
        """ L as a full matrix: the solves and the density read the lower triangle and the diagonal blocks only, a product with L
        (``correlate``) or a download reads the zeros above the diagonal as well """
        if getattr(self, '_upper_unset', False):
            f, _ = _lib.fn('lgp_zero_upper', self._dtype)
            _lib.check(f(self._ctx.handle, self.n, self._L.ptr, self._L.ld), 'lgp_zero_upper')
            self._upper_unset = False
        return self._L

    def factor(self):
        """ the lower factor L (device resident); not part of the reference API """
        return self._full_factor()

    def ginv_linear(self, X, *, device=False):
        # = K⁻¹X = L'⁻¹(L⁻¹X)                                  (:398-403)
        B, vec = self._rhs(X, copy=True)
        self._trsm(B, 0)
        self._trsm(B, 1)
        return self._out(B, vec, device)

    def pinv_bilinear(self, A, r, *, device=False):
        # = A'K⁻¹r = (L⁻¹A)'(L⁻¹r)                             (:405-409)
        invLr, rvec = self._rhs(r, copy=True)
        invLA, avec = self._rhs(A, copy=True)
        self._trsm(invLr, 0)
        self._trsm(invLA, 0)
        out = self._gemm(True, False, invLA, invLr)
        res = self._out(out, False, device)
        if not device:
            if avec and rvec:
                res = res[0, 0]
            elif rvec:
                res = res[:, 0]
            elif avec:
                res = res[0, :]
        return res

    def pinv_bilinear_robj(self, A, r):
        # = A'K⁻¹r with r an array of objects                  (:411-415)
        Lh = self._full_factor().to_host()
        invLr = solve_triangular_python(Lh, numpy.asarray(r), lower=True)
        invLA, avec = self._rhs(A, copy=True)
        self._trsm(invLA, 0)
        invLA = self._out(invLA, avec, False)
        return numpy.asarray(invLA).T @ invLr

    def ginv_quad(self, A, *, device=False):
        # = A'K⁻¹A = (L⁻¹A)'(L⁻¹A)                             (:417-420)
        invLA, vec = self._rhs(A, copy=True)
        self._trsm(invLA, 0)
        out = self._gemm(True, False, invLA, invLA)
        res = self._out(out, False, device)
        if vec and not device:
            res = res[0, 0]
        return res

    def ginv_diagquad(self, A, *, device=False):
        # = diag(A'K⁻¹A) = sum_j (L⁻¹A)_ji^2                   (:422-427)
        invLA, vec = self._rhs(A, copy=True)
        self._trsm(invLA, 0)
        out = DeviceArray.empty((invLA.shape[1],), self._dtype, self._ctx)
        f, _ = _lib.fn('lgp_colsumsq', self._dtype)
        _lib.check(f(self._ctx.handle, invLA.shape[0], invLA.shape[1], invLA.ptr, invLA.ld, out.ptr), 'lgp_colsumsq')
        if device:
            return out
        res = out.to_host()
        return res[0] if vec else res

    def correlate(self, x, *, device=False):
        # = Lx                                                   (:429-431)
        X, vec = self._rhs(x, copy=False)
        return self._out(self._gemm(False, False, self._full_factor(), X), vec, device)

    def back_correlate(self, X, *, device=False):
        # = L'X                                                  (:433-435)
        Xd, vec = self._rhs(X, copy=False)
        return self._out(self._gemm(True, False, self._full_factor(), Xd), vec, device)

    def pinv_correlate(self, x, *, device=False):
        # = L⁻¹x                                                 (:437-439)
        B, vec = self._rhs(x, copy=True)
        self._trsm(B, 0)
        return self._out(B, vec, device)

    def _dK_list(self, dK):
        """ derivative matrices as a list of (n, n) device matrices: accepts the reference's (n, n, p) array
        (``∂K_ij/∂p_k``) or a sequence of host / device matrices """
        if dK is None:
            return None
        if isinstance(dK, DeviceArray):
            mats = [dK]
        elif isinstance(dK, (list, tuple)):
            mats = list(dK)
        else:
            arr = numpy.asarray(dK)
            if arr.ndim == 2:
                arr = arr[:, :, None]
            if arr.ndim != 3:
                raise ValueError(f'dK must have shape (n, n, p), found {arr.shape}')
            mats = [numpy.ascontiguousarray(arr[:, :, k]) for k in range(arr.shape[2])]
        out = []
        zero = None
        for m in mats:
            if m is None:                          # a parameter the matrix does not depend on
                if zero is None:
                    zero = DeviceArray.zeros((self.n, self.n), self._dtype, self._ctx)
                m = zero
            if not isinstance(m, DeviceArray):
                m = DeviceArray.from_host(numpy.asarray(m), dtype=self._dtype, ctx=self._ctx)
            if m.shape != (self.n, self.n) or m.dtype != self._dtype:
                raise ValueError(f'derivative matrix has shape {m.shape} / dtype {m.dtype}, expected {(self.n, self.n)} / {self._dtype}')
            out.append(m)
        return out

    def normal_density_derivs(self, r, dK=None, dr=None, *, fisher=False):
        """
        Value, forward-mode gradient and (optionally) Fisher matrix of the minus log density in ONE device call
        (``lgp_mlnd_grad``; ``_decomp.py:461-561``): ``dK`` as for `_dK_list`, ``dr`` an (n, p) host or device
        matrix.  Returns ``(value, grad, fisher_or_None)`` as host scalars / arrays.  Nothing n x n is downloaded.
        """
        mats = self._dK_list(dK)
        drd = None
        if dr is not None:
            drd = dr.as2d() if isinstance(dr, DeviceArray) else DeviceArray.from_host(numpy.asarray(dr).reshape(self.n, -1), dtype=self._dtype, ctx=self._ctx)
            if drd.shape[0] != self.n:
                raise ValueError(f'dr has {drd.shape[0]} rows, the decomposed matrix has {self.n}')
        p = len(mats) if mats is not None else (drd.shape[1] if drd is not None else 0)
        if mats is not None and drd is not None and drd.shape[1] != p:
            raise ValueError(f'dK has {p} parameters, dr has {drd.shape[1]}')
        rd, _ = self._rhs(r, copy=True)
        if rd.shape[1] != 1:
            raise ValueError('r must be a vector')
        f, F = _lib.fn('lgp_mlnd_grad', self._dtype)
        val = F()
        grad = (F * max(p, 1))()
        fish = (F * max(p * p, 1))()
        ptrs = None
        ldk = 0
        if mats is not None and p:
            lds = {m.ld for m in mats}
            if len(lds) != 1:
                mats = [m.copy() for m in mats]
            ptrs = (ctypes.c_void_p * p)(*[m.ptr for m in mats])
            ldk = mats[0].ld
        _lib.check(f(self._ctx.handle, self.n, self._L.ptr, self._L.ld, self._dblk.ptr, rd.ptr, p, ptrs, ldk,
                     None if drd is None else drd.ptr, 0 if drd is None else drd.ld,
                     ctypes.byref(val), grad if p else None, fish if (fisher and p) else None), 'lgp_mlnd_grad')
        t = self._dtype.type
        g = numpy.array([t(grad[k]) for k in range(p)], self._dtype)
        Fm = numpy.array([t(v) for v in fish[:p * p]], self._dtype).reshape(p, p) if fisher else None
        return t(val.value), g, Fm

    def minus_log_normal_density(self, r, *, dr_vjp=None, dK_vjp=None,
        dr_jvp_vec=None, dK_jvp_vec=None, dr=None, dK=None, value=False,
        gradrev=False, gradfwd=False, fisher=False, fishvec=False):
        """
        Minus log of a Normal density with covariance K and its derivatives, same arguments and 5-tuple
        ``(value, gradrev, gradfwd, fisher, fishvec)`` as ``_decomp.py:441-586``.

        Array-valued derivative inputs (``dK``, ``dr``: forward mode, what `empbayes_fit` uses) are contracted on the
        device by `normal_density_derivs`.  The reverse-mode inputs are *host callables* by contract (``dK_vjp(X)``
        takes an (n, n) array), so K^-1 is downloaded for them once.
        """
        val = rev = fwd = fis = fvec = None
        have_fwd_in = dK is not None or dr is not None

        if (gradfwd or fisher) and have_fwd_in:
            v, g, Fm = self.normal_density_derivs(r, dK, dr, fisher=fisher)
            if value:
                val = v
            if gradfwd:
                fwd = g
            if fisher:
                fis = Fm
        if gradfwd and fwd is None:
            fwd = 0
        if fisher and fis is None:
            fis = 0
        if value and val is None:
            memo = getattr(self, '_density_memo', None)
            if memo is not None and isinstance(r, numpy.ndarray) and r.dtype == memo[0].dtype and r.shape == memo[0].shape and numpy.array_equal(r, memo[0]):
                # the posterior of the same data solved L^-1 r on the way (`GP.pred` -> `lgp_posterior`), and kept the density
                val = memo[1]
        if value and val is None:
            rd, _ = self._rhs(r, copy=True)
            if rd.shape[1] != 1:
                raise ValueError('r must be a vector')
            f, F = _lib.fn('lgp_mlnd_value', self._dtype)
            out = F()
            _lib.check(f(self._ctx.handle, self.n, self._L.ptr, self._L.ld, self._dblk.ptr, rd.ptr, ctypes.byref(out)), 'lgp_mlnd_value')
            val = self._dtype.type(out.value)

        if gradrev:
            rev = 0
            if dK_vjp is not None or dr_vjp is not None:
                invKr = self.ginv_linear(r)
                if dK_vjp is not None:
                    invK = self.ginv()
                    rev = rev + (dK_vjp(invK) - dK_vjp(numpy.outer(invKr, invKr))) / 2
                if dr_vjp is not None:
                    rev = rev + dr_vjp(invKr)

        if fishvec:
            fvec = 0
            if dK_jvp_vec is not None or dK_vjp is not None:
                if dK_jvp_vec is None or dK_vjp is None:
                    raise ValueError('fishvec needs both dK_jvp_vec and dK_vjp')
                half = self.ginv_linear(numpy.asarray(dK_jvp_vec))          # K^-1 (dK v)
                fvec = fvec + dK_vjp(self.ginv_linear(half.T)) / 2          # K^-1 (dK v) K^-1, the matrix being symmetric
            if dr_jvp_vec is not None or dr_vjp is not None:
                if dr_jvp_vec is None or dr_vjp is None:
                    raise ValueError('fishvec needs both dr_jvp_vec and dr_vjp')
                fvec = fvec + dr_vjp(self.ginv_linear(numpy.asarray(dr_jvp_vec)))

        return val, rev, fwd, fis, fvec

    @property
    def n(self):
        return self._L.shape[0]

    m = n

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
    Substitution on object arrays (``_decomp.py:280-309``); host Python by
    nature, used only by `pinv_bilinear_robj`.
    """
    a = numpy.asarray(a)
    x = numpy.copy(b)
    vec = x.ndim < 2
    if vec:
        x = x[:, None]
    n = a.shape[-1]
    assert x.shape[-2] == n
    if not lower:
        a = a[..., ::-1, ::-1]
        x = x[..., ::-1, :]
    x[..., 0, :] /= a[..., 0, 0, None]
    for i in range(1, n):
        x[..., i:, :] -= x[..., None, i - 1, :] * a[..., i:, i - 1, None]
        x[..., i, :] /= a[..., i, i, None]
    if not lower:
        x = x[..., ::-1, :]
    if vec:
        x = numpy.squeeze(x, -1)
    return x


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
            raise numpy.linalg.LinAlgError('cholesky decomposition not finite, probably matrix not pos def numerically')
        self._L = L
        self._dblk = dblk
        self._eps = self._dtype.type(eps.value)                        # :393

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
        return self._K

    def factor(self):
        """ the lower factor L (device resident); not part of the reference API """
        return self._L

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
        Lh = self._L.to_host()
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
        return self._out(self._gemm(False, False, self._L, X), vec, device)

    def back_correlate(self, X, *, device=False):
        # = L'X                                                  (:433-435)
        Xd, vec = self._rhs(X, copy=False)
        return self._out(self._gemm(True, False, self._L, Xd), vec, device)

    def pinv_correlate(self, x, *, device=False):
        # = L⁻¹x                                                 (:437-439)
        B, vec = self._rhs(x, copy=True)
        self._trsm(B, 0)
        return self._out(B, vec, device)

    def minus_log_normal_density(self, r, *, dr_vjp=None, dK_vjp=None,
        dr_jvp_vec=None, dK_jvp_vec=None, dr=None, dK=None, value=False,
        gradrev=False, gradfwd=False, fisher=False, fishvec=False):
        """
        ``_decomp.py:441-586``.  The value is one fused device call; the
        derivative pieces are compositions of the device solves with small host
        contractions (the derivative inputs are host callables / arrays).
        """
        out = {}
        grad = (
            (gradrev and (dK_vjp is not None or dr_vjp is not None))
            or (gradfwd and (dK is not None or dr is not None))
        )
        if grad:
            invKr = self.ginv_linear(r)                                   # :468-469
        if (gradrev and dK_vjp is not None) or (gradfwd and dK is not None):
            invK = self.ginv()                                            # :470-472

        if value:                                                         # :474-488
            rd, _ = self._rhs(r, copy=True)
            if rd.shape[1] != 1:
                raise ValueError('r must be a vector')
            f, F = _lib.fn('lgp_mlnd_value', self._dtype)
            val = F()
            _lib.check(f(self._ctx.handle, self.n, self._L.ptr, self._L.ld, self._dblk.ptr, rd.ptr, ctypes.byref(val)), 'lgp_mlnd_value')
            out['value'] = self._dtype.type(val.value)
        else:
            out['value'] = None

        if gradrev:                                                       # :492-514
            out['gradrev'] = 0
            if dK_vjp is not None:
                tr_invK_dK = dK_vjp(invK)
                r_invK_dK_invK_r = dK_vjp(numpy.outer(invKr, invKr))
                out['gradrev'] += 1/2 * (tr_invK_dK - r_invK_dK_invK_r)
            if dr_vjp is not None:
                out['gradrev'] += dr_vjp(invKr)
        else:
            out['gradrev'] = None

        if gradfwd:                                                       # :516-534
            out['gradfwd'] = 0
            if dK is not None:
                dK = numpy.asarray(dK)
                tr_invK_dK = numpy.einsum('ij,ijk->k', invK, dK)
                r_invK_dK_invK_r = numpy.einsum('i,ijk,j->k', invKr, dK, invKr)
                out['gradfwd'] += 1/2 * (tr_invK_dK - r_invK_dK_invK_r)
            if dr is not None:
                out['gradfwd'] += invKr @ numpy.asarray(dr)
        else:
            out['gradfwd'] = None

        if fisher:                                                        # :536-561
            out['fisher'] = 0
            if dK is not None:
                dKk = numpy.moveaxis(numpy.asarray(dK), 2, 0)
                invL_dK = [self.pinv_correlate(m) for m in dKk]           # kim: L⁻¹_il dK_lmk
                invL_dK_invL = numpy.stack([self.pinv_correlate(m.T) for m in invL_dK])
                out['fisher'] += 1/2 * numpy.einsum('kij,qij->kq', invL_dK_invL, invL_dK_invL)
            if dr is not None:
                out['fisher'] += self.ginv_quad(numpy.asarray(dr))
        else:
            out['fisher'] = None

        if fishvec:                                                       # :563-584
            out['fishvec'] = 0
            if not (dK_jvp_vec is None and dK_vjp is None):
                invK_dKv = self.ginv_linear(numpy.asarray(dK_jvp_vec))
                invK_dKv_invK = self.ginv_linear(invK_dKv.T)
                out['fishvec'] += 1/2 * dK_vjp(invK_dKv_invK)
            if not (dr_jvp_vec is None and dr_vjp is None):
                out['fishvec'] += dr_vjp(self.ginv_linear(numpy.asarray(dr_jvp_vec)))
        else:
            out['fishvec'] = None

        return tuple(out.values())

    @property
    def n(self):
        return self._L.shape[0]

    m = n

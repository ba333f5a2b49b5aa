"""
CPU restatement (numpy/scipy) of ``src/lsqfitgp/_linalg/_decomp.py`` -- test
infrastructure only, see ``oracle/__init__.py``.  parity unpinned vs JAX-CPU.
"""

import abc

import numpy
from scipy import linalg


def float_type(*args):
    """ ``src/lsqfitgp/_jaxext/__init__.py:140-144``: result type of the
    arguments, promoted to at least a floating type """
    t = numpy.result_type(*args)
    return t if numpy.issubdtype(t, numpy.floating) else numpy.result_type(t, numpy.float32 if t == numpy.float32 else numpy.float64)


def eigval_bound(K):
    """ ``_decomp.py:349-354``: Gershgorin bound, max_i sum_j |K_ij| """
    return numpy.max(numpy.sum(numpy.abs(K), axis=1))


def diag_scale_pow2(K):
    """ ``_decomp.py:356-361``: s_i = 2^rint(log2(K_ii)/2), 1 where K_ii == 0 """
    d = numpy.diag(K)
    with numpy.errstate(divide='ignore', invalid='ignore'):
        s = numpy.exp2(numpy.rint(0.5 * numpy.log2(d)))
    return numpy.where(d, s, 1).astype(K.dtype)


def solve_triangular_python(a, b, *, lower=False):
    """ ``_decomp.py:280-309``: substitution that also works on object arrays """
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


class Decomposition(abc.ABC):
    """ ``_decomp.py:126-278`` abstract surface """

    @abc.abstractmethod
    def __init__(self, *args, **kw): pass
    @abc.abstractmethod
    def matrix(self): pass
    @abc.abstractmethod
    def ginv_linear(self, X): pass
    @abc.abstractmethod
    def pinv_bilinear(self, A, r): pass
    @abc.abstractmethod
    def pinv_bilinear_robj(self, A, r): pass
    @abc.abstractmethod
    def ginv_quad(self, A): pass
    @abc.abstractmethod
    def ginv_diagquad(self, A): pass
    @abc.abstractmethod
    def correlate(self, x): pass
    @abc.abstractmethod
    def back_correlate(self, X): pass
    @abc.abstractmethod
    def pinv_correlate(self, x): pass
    @abc.abstractmethod
    def minus_log_normal_density(self, r, **kw): pass

    def _parseeps(self, K, epsrel, epsabs, maxeigv=None):
        """ ``_decomp.py:245-255`` """
        machine_eps = numpy.finfo(K.dtype).eps
        if isinstance(epsrel, str) and epsrel == 'auto':
            epsrel = len(K) * machine_eps
        if isinstance(epsabs, str) and epsabs == 'auto':
            epsabs = machine_eps
        if maxeigv is None:
            maxeigv = eigval_bound(K)
        self._eps = epsrel * maxeigv + epsabs
        return self._eps

    @property
    def eps(self):
        return self._eps

    @property
    @abc.abstractmethod
    def n(self): pass

    @property
    @abc.abstractmethod
    def m(self): pass

    def ginv(self):
        """ ``_decomp.py:276-278`` """
        return self.ginv_quad(numpy.eye(self.n, dtype=self._L.dtype))


def _st(L, B, lower=True, trans=0):
    return linalg.solve_triangular(L, B, lower=lower, trans=trans, check_finite=False)


class Chol(Decomposition):
    """ ``_decomp.py:376-658`` (value path) """

    def __init__(self, K, *, epsrel='auto', epsabs=0):
        K = numpy.asarray(K)
        if not numpy.issubdtype(K.dtype, numpy.floating):
            K = K.astype(float)
        self._K = K                                     # :383
        s = diag_scale_pow2(K)                          # :384
        K = K / s / s[:, None]                          # :385
        eps = self._parseeps(K, epsrel, epsabs)         # :386
        K = K.copy()
        K[numpy.diag_indices_from(K)] += K.dtype.type(eps)   # :387
        # :388 jlinalg.cholesky(K, lower=True): non-finite instead of raising
        try:
            L = linalg.cholesky(K, lower=True, check_finite=False)
        except linalg.LinAlgError:
            L = numpy.full_like(K, numpy.nan)
        if not numpy.all(numpy.isfinite(L)):            # :389-391
            raise numpy.linalg.LinAlgError('cholesky decomposition not finite, probably matrix not pos def numerically')
        self._L = L * s[:, None]                        # :392
        self._eps = eps * numpy.min(s * s)              # :393

    def matrix(self):
        return self._K

    def ginv_linear(self, X):                           # :398-403
        invLX = _st(self._L, X)
        return _st(self._L, invLX, trans=1)

    def pinv_bilinear(self, A, r):                      # :405-409
        invLr = _st(self._L, r)
        invLA = _st(self._L, A)
        return invLA.T @ invLr

    def pinv_bilinear_robj(self, A, r):                 # :411-415
        invLr = solve_triangular_python(self._L, r, lower=True)
        invLA = _st(self._L, A)
        return numpy.asarray(invLA).T @ invLr

    def ginv_quad(self, A):                             # :417-420
        invLA = _st(self._L, A)
        return invLA.T @ invLA

    def ginv_diagquad(self, A):                         # :422-427
        invLA = _st(self._L, A)
        return numpy.einsum('ji,ji->i', invLA, invLA)

    def correlate(self, x):                             # :429-431
        return self._L @ x

    def back_correlate(self, X):                        # :433-435
        return self._L.T @ X

    def pinv_correlate(self, x):                        # :437-439
        return _st(self._L, x)

    def minus_log_normal_density(self, r, *, dr_vjp=None, dK_vjp=None,
        dr_jvp_vec=None, dK_jvp_vec=None, dr=None, dK=None, value=False,
        gradrev=False, gradfwd=False, fisher=False, fishvec=False):
        """ ``_decomp.py:441-586``; value and the forward-mode pieces
        (gradfwd, fisher) which need only arrays; the vjp-callable pieces
        follow the same formulas with user callables """
        L = self._L
        out = {}
        grad = (
            (gradrev and (dK_vjp is not None or dr_vjp is not None))
            or (gradfwd and (dK is not None or dr is not None))
        )
        if value or grad:
            invLr = _st(L, r)
        if grad:
            invKr = _st(L, invLr, trans=1)
        if (gradrev and dK_vjp is not None) or (gradfwd and dK is not None):
            invL = _st(L, numpy.eye(len(L), dtype=L.dtype))
            invK = invL.T @ invL

        if value:                                       # :484-488
            out['value'] = 1/2 * (
                len(L) * numpy.log(2 * numpy.pi) +
                2 * numpy.sum(numpy.log(numpy.diag(L))) +
                invLr @ invLr
            )
        else:
            out['value'] = None

        if gradrev:                                     # :492-514
            out['gradrev'] = 0
            if dK_vjp is not None:
                tr_invK_dK = dK_vjp(invK)
                r_invK_dK_invK_r = dK_vjp(numpy.outer(invKr, invKr))
                out['gradrev'] += 1/2 * (tr_invK_dK - r_invK_dK_invK_r)
            if dr_vjp is not None:
                out['gradrev'] += dr_vjp(invKr)
        else:
            out['gradrev'] = None

        if gradfwd:                                     # :516-534
            out['gradfwd'] = 0
            if dK is not None:
                tr_invK_dK = numpy.einsum('ij,ijk->k', invK, dK)
                r_invK_dK_invK_r = numpy.einsum('i,ijk,j->k', invKr, dK, invKr)
                out['gradfwd'] += 1/2 * (tr_invK_dK - r_invK_dK_invK_r)
            if dr is not None:
                out['gradfwd'] += invKr @ dr
        else:
            out['gradfwd'] = None

        if fisher:                                      # :536-561
            out['fisher'] = 0
            if dK is not None:
                dKk = numpy.moveaxis(dK, 2, 0)
                invL_dK = numpy.stack([_st(L, m) for m in dKk])
                invL_dK_invL = numpy.stack([_st(L, m.T) for m in invL_dK])
                out['fisher'] += 1/2 * numpy.einsum('kij,qij->kq', invL_dK_invL, invL_dK_invL)
            if dr is not None:
                invLdr = _st(L, dr)
                out['fisher'] += invLdr.T @ invLdr
        else:
            out['fisher'] = None

        if fishvec:                                     # :563-584
            out['fishvec'] = 0
            if not (dK_jvp_vec is None and dK_vjp is None):
                invL_dKv = _st(L, dK_jvp_vec)
                invK_dKv = _st(L, invL_dKv, trans=1)
                invL_dKv_invK = _st(L, invK_dKv.T)
                invK_dKv_invK = _st(L, invL_dKv_invK, trans=1)
                out['fishvec'] += 1/2 * dK_vjp(invK_dKv_invK)
            if not (dr_jvp_vec is None and dr_vjp is None):
                invL_drv = _st(L, dr_jvp_vec)
                invK_drv = _st(L, invL_drv, trans=1)
                out['fishvec'] += dr_vjp(invK_drv)
        else:
            out['fishvec'] = None

        return tuple(out.values())

    @property
    def n(self):
        return len(self._L)

    m = n

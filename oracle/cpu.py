"""
ctypes face of the C++ / OpenMP CPU restatement of the hot path (``oracle/cpu_backend.cpp``, built by ``oracle/Makefile`` into
``oracle/_build/liblsqfitgp_cpu.so``): TEST / MEASUREMENT INFRASTRUCTURE.  Same C-ABI as the HIP library (same names and
argument lists, ``include/lsqfitgp_hip.h``) on host arrays; used by the CPU tests (pinned against the numpy oracle) and by
``bench.py``'s ``cpu_baseline`` leg.  Never imported by the product.
"""

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(HERE, '_build', 'liblsqfitgp_cpu.so')
_lib = None

c_int, c_double, c_size_t, c_void_p = ctypes.c_int, ctypes.c_double, ctypes.c_size_t, ctypes.c_void_p
P = ctypes.POINTER


def build():
    subprocess.run(['make', '-C', HERE, 'cpu'], check=True, stdout=subprocess.DEVNULL)
    return LIBPATH


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            build()
        lib = ctypes.CDLL(LIBPATH, mode=ctypes.RTLD_LOCAL)
        S, D = c_size_t, c_double
        sig = {
            'lgp_ctx_create': (c_int, [c_int, P(c_void_p)]),
            'lgp_ctx_destroy': (c_int, [c_void_p]),
            'lgp_gram_isotropic_f64': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, S, S, c_void_p, S, S, c_int, D, c_int, D,
                                               c_int, c_void_p, c_void_p, c_int, c_void_p, S]),
            'lgp_add_diag_f64': (c_int, [c_void_p, S, c_void_p, S, c_void_p, D]),
            'lgp_chol_factor_f64': (c_int, [c_void_p, S, c_void_p, S, D, D, c_int, c_void_p, S, c_void_p, P(D)]),
            'lgp_chol_factor_rhs_f64': (c_int, [c_void_p, S, c_void_p, S, D, D, c_int, c_void_p, S, S, c_void_p, P(D)]),
            'lgp_potrf_f64': (c_int, [c_void_p, S, c_void_p, S, c_void_p]),
            'lgp_trsm_f64': (c_int, [c_void_p, S, c_void_p, S, c_void_p, c_int, c_void_p, S, S]),
            'lgp_gemm_f64': (c_int, [c_void_p, c_int, c_int, S, S, S, D, c_void_p, S, c_void_p, S, D, c_void_p, S]),
            'lgp_posterior_f64': (c_int, [c_void_p, S, S, c_void_p, S, c_void_p, c_void_p, S, c_void_p, S, c_void_p, c_void_p, S, P(D)]),
            'lgp_mlnd_value_f64': (c_int, [c_void_p, S, c_void_p, S, c_void_p, c_void_p, P(D)]),
        }
        for name, (res, args) in sig.items():
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
        lib.lgp_last_error.restype = ctypes.c_char_p
        _lib = lib
    return _lib


def _ptr(a):
    return a.ctypes.data_as(c_void_p)


class CpuBackend:
    """ a context (= thread count) of the CPU backend and the calls of one fit + predict pass """

    EXPQUAD, MATERNP = 0, 1

    def __init__(self, threads=0):
        self.lib = load()
        h = c_void_p()
        assert self.lib.lgp_ctx_create(int(threads), ctypes.byref(h)) == 0
        self.h = h
        self.threads = threads

    def __del__(self):
        try:
            self.lib.lgp_ctx_destroy(self.h)
        except Exception:
            pass

    def gram(self, kernel, p, x, y, scale=None, out=None):
        """ x: (d, n), y: (d, m) SoA covariates -> (n, m); out: a C-contiguous (n, m) array to fill instead of a new one """
        x = np.ascontiguousarray(np.atleast_2d(x), float); y = np.ascontiguousarray(np.atleast_2d(y), float)
        d, n = x.shape
        m = y.shape[1]
        if out is None:
            out = np.empty((n, m))
        assert out.shape == (n, m) and out.flags.c_contiguous and out.dtype == np.float64
        st = self.lib.lgp_gram_isotropic_f64(self.h, kernel, p, 0, None, d, _ptr(x), n, n, _ptr(y), m, m, 0, 0., int(scale is not None),
                                             float(scale or 1.), 0, None, None, 0, _ptr(out), m)
        assert st == 0, self.lib.lgp_last_error()
        return out

    def chol(self, K, rhs=None, epsrel='auto', epsabs=0.):
        """ Chol.__init__ of the reference; returns (L with the carried rows (L^-1 rhs)^T below it, eps) """
        n = len(K)
        q = 0 if rhs is None else rhs.shape[1]
        L = np.empty((n + q, n))
        if q:
            L[n:] = rhs.T
        flags = 0
        if isinstance(epsrel, str):
            flags |= 1; epsrel = 0.
        if isinstance(epsabs, str):
            flags |= 2; epsabs = 0.
        eps = c_double()
        K = np.ascontiguousarray(K, float)
        info = self.lib.lgp_chol_factor_rhs_f64(self.h, n, _ptr(K), n, epsrel, epsabs, flags, _ptr(L), n, q, None, ctypes.byref(eps))
        if info != 0:
            raise np.linalg.LinAlgError(f'cholesky decomposition not finite, probably matrix not pos def numerically (info {info})')
        return L, eps.value

    def chol_inplace(self, L, n, epsrel='auto', epsabs=0.):
        """ the same on ONE buffer (for matrices that fit the host only once): L is (n + q, n), rows 0 .. n-1 hold K on entry, the rows
        below it rhs^T; on exit the factor and (L^-1 rhs)^T.  (chol_factor of cpu_backend.cpp reads a row of K before it writes that
        row of L, and nothing else of K afterwards.) """
        assert L.flags.c_contiguous and L.dtype == np.float64 and L.shape[1] == n and L.shape[0] >= n
        q = L.shape[0] - n
        flags = 0
        if isinstance(epsrel, str):
            flags |= 1; epsrel = 0.
        if isinstance(epsabs, str):
            flags |= 2; epsabs = 0.
        eps = c_double()
        info = self.lib.lgp_chol_factor_rhs_f64(self.h, n, _ptr(L), n, epsrel, epsabs, flags, _ptr(L), n, q, None, ctypes.byref(eps))
        if info != 0:
            raise np.linalg.LinAlgError(f'cholesky decomposition not finite, probably matrix not pos def numerically (info {info})')
        return eps.value

    def trsm(self, L, B, trans=False):
        B = np.array(B, float, order='C', ndmin=2)
        n = len(L)
        assert self.lib.lgp_trsm_f64(self.h, n, _ptr(L), L.strides[0] // 8, None, int(trans), _ptr(B), B.shape[1], B.shape[1]) == 0
        return B

    def gemm(self, A, B, ta=False, tb=False):
        A = np.ascontiguousarray(A, float); B = np.ascontiguousarray(B, float)
        m = A.shape[1] if ta else A.shape[0]
        k = A.shape[0] if ta else A.shape[1]
        n = B.shape[0] if tb else B.shape[1]
        C = np.empty((m, n))
        assert self.lib.lgp_gemm_f64(self.h, int(ta), int(tb), m, n, k, 1., _ptr(A), A.shape[1], _ptr(B), B.shape[1], 0., _ptr(C), n) == 0
        return C

    def posterior(self, L, W, Kss):
        """ W = [Kxxs | y] (n, m + 1), overwritten; returns mean, cov, minus log density """
        n, k = W.shape
        m = k - 1
        mean = np.empty(m); cov = np.empty((m, m)); val = c_double()
        assert self.lib.lgp_posterior_f64(self.h, n, m, _ptr(L), L.strides[0] // 8, None, _ptr(W), k, _ptr(Kss), m, _ptr(mean), _ptr(cov), m,
                                          ctypes.byref(val)) == 0
        return mean, cov, val.value

"""
ctypes binding of ``liblsqfitgp_hip.so`` (``include/lsqfitgp_hip.h``).

This is the thin C-ABI layer BASELINE.json's north star asks for: Python host
code -> ctypes -> hand-written HIP kernels.  There is NO CPU fallback: if the
library is missing, cannot be loaded, or no gfx950 GPU is visible, every
operation of the package raises ``LibraryError``.
"""

import ctypes
import os
from ctypes import (POINTER, c_char_p, c_double, c_float, c_int, c_int32,
                    c_size_t, c_void_p)

import numpy

from . import _build


class LibraryError(RuntimeError):
    """ the HIP library is missing / not loadable / reported an error """


LGP_OK = 0
LGP_ENOMEM = -4
LGP_KERNEL_EXPQUAD = 0
LGP_KERNEL_MATERNP = 1
LGP_KERNEL_GAMMAEXP = 2
LGP_KERNEL_CAUCHY = 3
LGP_KERNEL_EXPQUAD_DIFF = 4
LGP_KERNEL_MATERNP_DIFF = 5
LGP_KERNEL_WENDLAND = 6
LGP_KERNEL_CAUSALEXPQUAD = 7
LGP_KERNEL_LOG = 8
LGP_KERNEL_CONSTANT = 9
LGP_KERNEL_WHITE = 10
LGP_KERNEL_EXPON = 11
LGP_KERNEL_HOLEEFFECT = 12
LGP_KERNEL_PERIODIC = 13
LGP_KERNEL_COS = 14
LGP_KERNEL_SINC = 15
LGP_KERNEL_LINEAR = 16
LGP_KERNEL_WIENER = 17
LGP_KERNEL_ORNSTEINUHLENBECK = 18
LGP_KERNEL_BROWNIANBRIDGE = 19
LGP_KERNEL_DECAYING = 20
LGP_KERNEL_FRACBROWNIAN = 21
LGP_KERNEL_WIENERINTEGRAL = 22
LGP_KERNEL_STATIONARYFRACBROWNIAN = 23
LGP_KERNEL_CELERITE = 24
LGP_KERNEL_HARMONIC = 25
LGP_KERNEL_CIRCULAR = 26
LGP_KERNEL_TAYLOR = 27
LGP_KERNEL_GIBBS = 28
LGP_KERNEL_MATERN = 29
LGP_KERNEL_LAGTABLE = 30
LGP_KERNEL_COLOR = 31
LGP_KERNEL_PINK = 32
LGP_KERNEL_BESSEL = 33
LGP_KERNEL_PSERIES = 34
LGP_KERNEL_NNKERNEL = 35
LGP_OP_MUL = 0
LGP_OP_ADD = 1
LGP_EPS_AUTO_REL = 1
LGP_EPS_AUTO_ABS = 2
LGP_FACTOR_LOWER_ONLY = 4
LGP_COMM_ID_BYTES = 128

_P = c_void_p          # device pointer
_S = c_size_t


def _signatures():
    """ name -> (restype, argtypes); mirrors include/lsqfitgp_hip.h one to one """
    sig = {
        'lgp_version': (c_int, []),
        'lgp_last_error': (c_char_p, []),
        'lgp_device_count': (c_int, [POINTER(c_int)]),
        'lgp_ctx_create': (c_int, [c_int, POINTER(c_void_p)]),
        'lgp_ctx_destroy': (c_int, [c_void_p]),
        'lgp_sync': (c_int, [c_void_p]),
        'lgp_device_info': (c_int, [c_void_p, c_char_p, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_size_t)]),
        'lgp_malloc': (c_int, [c_void_p, _S, POINTER(c_void_p)]),
        'lgp_free': (c_int, [c_void_p, _P]),
        'lgp_memset': (c_int, [c_void_p, _P, c_int, _S]),
        'lgp_upload': (c_int, [c_void_p, _P, c_void_p, _S]),
        'lgp_download': (c_int, [c_void_p, c_void_p, _P, _S]),
        'lgp_copy2d': (c_int, [c_void_p, _P, _S, _P, _S, _S, _S, _S]),
        'lgp_convert_f32_f64': (c_int, [c_void_p, _S, _S, _P, _S, _P, _S]),
        'lgp_convert_f64_f32': (c_int, [c_void_p, _S, _S, _P, _S, _P, _S]),
        'lgp_event_create': (c_int, [c_void_p, POINTER(c_void_p)]),
        'lgp_event_destroy': (c_int, [c_void_p, c_void_p]),
        'lgp_event_record': (c_int, [c_void_p, c_void_p]),
        'lgp_event_elapsed_ms': (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_float)]),
        'lgp_chol_dblk_size': (c_size_t, [_S]),
        'lgp_profile_begin': (c_int, [c_void_p]),
        'lgp_profile_end': (c_int, [c_void_p, POINTER(c_size_t), POINTER(c_double), POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
        'lgp_predict_isotropic_f64': (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_double), c_int, c_void_p, _S, _S, c_int, c_double, c_int, c_double,
                                              c_int, POINTER(c_int), POINTER(c_double), c_void_p, c_void_p, _P, _P, _S, _S, _P, _P, c_void_p, POINTER(c_double)]),
        'lgp_comm_unique_id': (c_int, [c_void_p]),
        'lgp_comm_init': (c_int, [c_void_p, c_int, c_int, c_void_p]),
        'lgp_comm_init_host': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
        'lgp_comm_set_bcast': (c_int, [c_void_p, c_int, c_size_t]),
        'lgp_bcast_plan': (ctypes.c_long, [c_size_t, c_int, c_int, c_void_p, c_size_t, c_void_p]),
        'lgp_comm_destroy': (c_int, [c_void_p]),
        'lgp_comm_info': (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
        'lgp_bcast': (c_int, [c_void_p, _P, _S, c_int]),
        'lgp_allreduce_sum_f64': (c_int, [c_void_p, _P, _S]),
        'lgp_allreduce_max_f64': (c_int, [c_void_p, _P, _S]),
        'lgp_panel_factor_f64': (c_int, [c_void_p, _S, _S, _P, _S, _P, _S]),
        'lgp_chain_schedule': (ctypes.c_long, [_S, _P, _S]),
        'lgp_update_nt_cyclic_f64': (c_int, [c_void_p, _S, _S, _S, _P, _S, _P, _S, _S, _S]),
        'lgp_update_nt_f64': (c_int, [c_void_p, _S, _S, _S, _P, _S, _P, _S, _P, _S, c_int]),
        'lgp_get_info': (c_int, [c_void_p, POINTER(c_int)]),
        'lgp_select_stream': (c_int, [c_void_p, c_int]),
        'lgp_stream_wait_event': (c_int, [c_void_p, c_void_p]),
        'lgp_colabssum_f64': (c_int, [c_void_p, _S, _S, _P, _S, _P]),
        'lgp_chol_factor_dist_f64': (c_int, [c_void_p, _S, _S, _S, _P, _S, _P, _P, _S, _S, _P, POINTER(c_int)]),
        'lgp_chol_dist_plan': (ctypes.c_long, [_S, _S, _S, _S, _S, _P, _S, POINTER(c_size_t), POINTER(c_size_t)]),
        'lgp_dist_stats': (c_int, [c_void_p, c_int, _P]),
        'lgp_chol_dist_trace': (ctypes.c_long, [_S, _S, _S, c_int, c_int, _S, c_int, _P, _S]),
        'lgp_update_nt_slab_f64': (c_int, [c_void_p, _S, _S, _S, _P, _S, _P, _S, _P, _S, _S, _S, _S]),
        'lgp_max_f64': (c_int, [c_void_p, _S, _P, _P]),
        'lgp_lanczos_f64': (c_int, [c_void_p, _S, _P, _S, _S, _P, _P, _P]),
        'lgp_lanczos_f32': (c_int, [c_void_p, _S, _P, _S, _S, _P, _P, _P]),
        'lgp_scale_rows_cols_f64': (c_int, [c_void_p, _S, _S, _P, _S, _P, _P, c_int]),
        'lgp_diag_scale_pow2_f64': (c_int, [c_void_p, _S, _P, _S, _P]),
    }
    for suf, F in (('f64', c_double), ('f32', c_float)):
        sig.update({
            'lgp_transpose_' + suf: (c_int, [c_void_p, _P, _S, _P, _S, _S, _S]),
            'lgp_gram_isotropic_' + suf: (c_int, [c_void_p, c_int, c_int, c_int, POINTER(F), c_int, _P, _S, _S, _P, _S, _S,
                                                 c_int, F, c_int, F, c_int, POINTER(c_int), POINTER(F), c_int, _P, _S]),
            'lgp_gram_reg_cols_' + suf: (c_int, [c_void_p, c_int, c_int, c_int, POINTER(F), c_int, _P, _S, _S, _S, _S, _S, _S, c_int, F, c_int, F,
                                                c_int, POINTER(c_int), POINTER(F), F, _P, _P, _S, _P, _P]),
            'lgp_gram_pseries_' + suf: (c_int, [c_void_p, c_int, c_int, _P, _S, POINTER(F), _P, _S, _S, _P, _S, _S, c_int, F, c_int, F,
                                               c_int, POINTER(c_int), POINTER(F), c_int, _P, _S]),
            'lgp_pairs_pseries_' + suf: (c_int, [c_void_p, c_int, c_int, _P, _S, POINTER(F), _P, _S, _P, _S, _S, c_int, F, c_int, F,
                                                c_int, POINTER(c_int), POINTER(F), _P]),
            'lgp_gram_lagtable_' + suf: (c_int, [c_void_p, _P, _S, F, _P, _S, _S, _P, _S, _S, c_int, F, c_int, F,
                                                c_int, POINTER(c_int), POINTER(F), c_int, _P, _S]),
            'lgp_pairs_lagtable_' + suf: (c_int, [c_void_p, _P, _S, F, _P, _S, _P, _S, _S, c_int, F, c_int, F,
                                                 c_int, POINTER(c_int), POINTER(F), _P]),
            'lgp_gram_bart_' + suf: (c_int, [c_void_p, c_int, _P, _S, _P, _S, POINTER(c_int32), POINTER(F),
                                            c_int, c_int, POINTER(F), F, _P, _P, _S,
                                            c_int, POINTER(c_int), POINTER(F), _P, _S]),
            'lgp_gram_deriv_' + suf: (c_int, [c_void_p, c_int, c_int, c_int, POINTER(F), c_int, POINTER(F), c_int, _P, _S, _S, _P, _S, _S,
                                      c_int, F, c_int, F, F, c_int, c_int, c_int, c_int, _P, _S]),
            'lgp_gram_isotropic_dscale_' + suf: (c_int, [c_void_p, c_int, c_int, c_int, POINTER(F), c_int, _P, _S, _S, _P, _S, _S,
                                                        c_int, F, c_int, F, c_int, POINTER(c_int), POINTER(F), _P, _P, _S]),
            'lgp_gram_bart_jvp_' + suf: (c_int, [c_void_p, c_int, _P, _S, _P, _S, POINTER(c_int32), POINTER(F),
                                                c_int, c_int, POINTER(F), F, _P, _P, _S,
                                                c_int, POINTER(c_int), POINTER(F), _P, _S, c_int, POINTER(F), POINTER(c_void_p), POINTER(c_void_p)]),
            'lgp_pairs_isotropic_' + suf: (c_int, [c_void_p, c_int, c_int, c_int, POINTER(F), c_int, _P, _S, _P, _S, _S,
                                                  c_int, F, c_int, F, c_int, POINTER(c_int), POINTER(F), _P]),
            'lgp_pairs_bart_' + suf: (c_int, [c_void_p, c_int, _P, _P, _S, POINTER(c_int32), _P, POINTER(F),
                                             c_int, c_int, POINTER(F), F, _P, _P, _S,
                                             c_int, POINTER(c_int), POINTER(F), _P]),
            'lgp_allclose_' + suf: (c_int, [c_void_p, _S, _S, _P, _S, _P, _S, F, F, POINTER(c_int)]),
            'lgp_axpby_' + suf: (c_int, [c_void_p, _S, _S, F, _P, _S, F, _P, _S, _P, _S]),
            'lgp_add_diag_' + suf: (c_int, [c_void_p, _S, _P, _S, _P, F]),
            'lgp_add_scalar_' + suf: (c_int, [c_void_p, _S, _S, _P, _S, F]),
            'lgp_mul_' + suf: (c_int, [c_void_p, _S, _S, F, _P, _S, _P, _S, F, _P, _S, _P, _S]),
            'lgp_unary_' + suf: (c_int, [c_void_p, _S, _S, _P, _S, c_int, F, F]),
            'lgp_check_matrix_' + suf: (c_int, [c_void_p, _S, _S, _P, _S, c_int, F, F, POINTER(c_int)]),
            'lgp_check_matrix_dev_' + suf: (c_int, [c_void_p, _S, _S, _P, _S, c_int, F, F, _P]),
            'lgp_chol_factor_' + suf: (c_int, [c_void_p, _S, _P, _S, F, F, c_int, _P, _S, _P, POINTER(F)]),
            'lgp_chol_factor_rhs_' + suf: (c_int, [c_void_p, _S, _P, _S, F, F, c_int, _P, _S, _S, _P, POINTER(F)]),
            'lgp_gram_chol_factor_rhs_' + suf: (c_int, [c_void_p, c_int, c_int, c_int, POINTER(F), c_int, _P, _S, _S, c_int, F, c_int, F,
                                                       c_int, POINTER(c_int), POINTER(F), F, _P, F, F, c_int, _P, _S, _S, _P, POINTER(F)]),
            'lgp_potrf_' + suf: (c_int, [c_void_p, _S, _P, _S, _P]),
            'lgp_zero_upper_' + suf: (c_int, [c_void_p, _S, _P, _S]),
            'lgp_trsm_' + suf: (c_int, [c_void_p, _S, _P, _S, _P, c_int, _P, _S, _S]),
            'lgp_gemm_' + suf: (c_int, [c_void_p, c_int, c_int, _S, _S, _S, F, _P, _S, _P, _S, F, _P, _S]),
            'lgp_colsumsq_' + suf: (c_int, [c_void_p, _S, _S, _P, _S, _P]),
            'lgp_logdet_diag_' + suf: (c_int, [c_void_p, _S, _P, _S, POINTER(F)]),
            'lgp_dot_' + suf: (c_int, [c_void_p, _S, _P, _P, POINTER(F)]),
            'lgp_posterior_' + suf: (c_int, [c_void_p, _S, _S, _P, _S, _P, _P, _S, _P, _S, _P, _P, _S, POINTER(F)]),
            'lgp_posterior_solved_' + suf: (c_int, [c_void_p, _S, _S, _P, _S, _P, _S, _P, _S, _P, _P, _S, POINTER(F)]),
            'lgp_mlnd_value_' + suf: (c_int, [c_void_p, _S, _P, _S, _P, _P, POINTER(F)]),
            'lgp_mlnd_grad_' + suf: (c_int, [c_void_p, _S, _P, _S, _P, _P, _S, POINTER(c_void_p), _S, _P, _S, POINTER(F), POINTER(F), POINTER(F)]),
        })
    return sig


SIGNATURES = _signatures()

_lib = None


def library_path():
    """ the in-tree library; LGP_LIBRARY names another BUILD of it (same sources, other compile-time switches: the same-box A/B
    comparisons of tools/ab_lib.sh), never another implementation """
    return os.environ.get('LGP_LIBRARY') or _build.LIBPATH


def load():
    """ load the shared library (once) and declare every prototype """
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise LibraryError(
            f'{path} not found: the HIP extension has not been built. Run '
            '`python -c "import __graft_entry__ as g; g.build()"` (needs hipcc); '
            'there is no CPU fallback.')
    try:
        lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except OSError as exc:
        raise LibraryError(f'cannot load {path}: {exc}') from exc
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise LibraryError(f'{path} does not export {name}') from exc
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error():
    msg = load().lgp_last_error()
    return msg.decode('utf8', 'replace') if msg else ''


_on_enomem = []         # callables that give cached device memory back (the contexts' block caches register here)


def check(status, what=''):
    """ negative status -> LibraryError; returns nonnegative status (info).  When the library itself ran out of device memory
    (a workspace of its own: LGP_ENOMEM), the blocks that the host classes keep for reuse are given back first, so that the
    caller's next attempt finds them free """
    if status < 0:
        msg = last_error()
        if status == LGP_ENOMEM:
            for release in list(_on_enomem):
                try:
                    release()
                except Exception:
                    pass
            msg += ' (the cached device blocks of the host classes have been released: retry)'
        raise LibraryError(f'{what or "HIP call"} failed (status {status}): {msg}')
    return status


def suffix(dtype):
    dtype = numpy.dtype(dtype)
    if dtype == numpy.float64:
        return 'f64', c_double
    if dtype == numpy.float32:
        return 'f32', c_float
    raise TypeError(f'unsupported floating type {dtype}')


def fn(name, dtype):
    """ typed entry point, e.g. fn('lgp_gemm', numpy.float32) """
    suf, ctype = suffix(dtype)
    return getattr(load(), f'{name}_{suf}'), ctype

/*
 * lsqfitgp_hip.h -- C-ABI of the MI355X (gfx950) dense-GP hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or framework
 * types.  Every entry point names the reference interface it replaces
 * (paths relative to the lsqfitgp checkout, v0.22.dev0).  The reference has no
 * FFI of its own (it is pure Python over jax); the binding a maintainer would
 * add is a ctypes stub, shown in INTEGRATION.md and implemented in
 * lsqfitgp_amd/_lib.py.
 *
 * Conventions
 *  - all matrices are dense ROW-MAJOR (C order, like the numpy/jax arrays of the
 *    reference) with an explicit leading dimension in ELEMENTS (ld >= cols);
 *  - `*_f64` entry points compute in IEEE double, `*_f32` in IEEE float;
 *  - pointers named d* / "device" are HIP device pointers obtained from
 *    lgp_malloc (or any hipMalloc allocation of the same device); pointers
 *    named h* are host pointers;
 *  - every function returns an int status: 0 = ok; >0 = LAPACK-style info
 *    (1-based index of the first non-positive / non-finite pivot);
 *    <0 = error (LGP_E*), message in lgp_last_error();
 *  - all work is enqueued on the context's stream; functions that return
 *    scalars to the host synchronise that stream, the others do not (call
 *    lgp_sync);
 *  - ownership: the caller owns every buffer and handle it allocates and frees
 *    it explicitly; the library never frees caller memory.  Re-entrant per
 *    context; no global mutable state except the thread-local error string.
 */
#ifndef LSQFITGP_HIP_H
#define LSQFITGP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LGP_OK 0
#define LGP_EINVAL (-1)   /* bad argument */
#define LGP_EHIP (-2)     /* HIP runtime error */
#define LGP_ECOMM (-3)    /* RCCL error */
#define LGP_ENOMEM (-4)   /* device allocation failed */

typedef struct lgp_ctx lgp_ctx;

/* ---- context, memory, timing ------------------------------------------------ */

int lgp_version(void);                                  /* ABI version */
const char *lgp_last_error(void);                       /* thread-local message */
int lgp_device_count(int *count);
int lgp_ctx_create(int device, lgp_ctx **ctx);          /* one stream + workspace on `device` */
int lgp_ctx_destroy(lgp_ctx *ctx);
int lgp_sync(lgp_ctx *ctx);                             /* wait for the context's stream(s) */
int lgp_device_info(lgp_ctx *ctx, char *name, int namelen, int *cus, int *clock_khz, size_t *mem_bytes);

int lgp_malloc(lgp_ctx *ctx, size_t bytes, void **dptr);
int lgp_free(lgp_ctx *ctx, void *dptr);
int lgp_memset(lgp_ctx *ctx, void *dptr, int byte, size_t bytes);
int lgp_upload(lgp_ctx *ctx, void *ddst, const void *hsrc, size_t bytes);     /* synchronous */
int lgp_download(lgp_ctx *ctx, void *hdst, const void *dsrc, size_t bytes);   /* synchronous */
/* strided device-to-device copy of a rows x cols block (elemsize 4 or 8); used to assemble
 * block matrices like jnp.block in _GP/_elements.py:642-649 */
int lgp_copy2d(lgp_ctx *ctx, void *ddst, size_t ldd, const void *dsrc, size_t lds,
               size_t rows, size_t cols, size_t elemsize);
/* dst (cols x rows) = src^T; the transposed block stored "for free" by _elements.py:637 */
int lgp_transpose_f64(lgp_ctx *ctx, double *ddst, size_t ldd, const double *dsrc, size_t lds, size_t rows, size_t cols);
int lgp_transpose_f32(lgp_ctx *ctx, float *ddst, size_t ldd, const float *dsrc, size_t lds, size_t rows, size_t cols);

/* dst = (type of dst) src, rows x cols: the dtype promotion of mixed-precision block assembly -- the reference's
 * float32 BART blocks meet float64 zero / noise blocks in jnp.block and tensordot (_GP/_elements.py:332-344,642-649) */
int lgp_convert_f32_f64(lgp_ctx *ctx, size_t rows, size_t cols, double *ddst, size_t ldd, const float *dsrc, size_t lds);
int lgp_convert_f64_f32(lgp_ctx *ctx, size_t rows, size_t cols, float *ddst, size_t ldd, const double *dsrc, size_t lds);

/* HIP-event timing on the context's stream (bench.py measures kernels with these) */
int lgp_event_create(lgp_ctx *ctx, void **event);
int lgp_event_destroy(lgp_ctx *ctx, void *event);
int lgp_event_record(lgp_ctx *ctx, void *event);
int lgp_event_elapsed_ms(lgp_ctx *ctx, void *start, void *stop, float *ms);   /* synchronises on stop */

/* Live timing of the dominant kernel, the syrk trailing update of the blocked Cholesky: between
 * lgp_profile_begin and lgp_profile_end every trailing-update launch is bracketed by a HIP event pair on the
 * stream it is launched on.  lgp_profile_end synchronises and returns the number of launches, the sum of
 * their durations, the sum of their algorithmic flops (2*k per produced lower-triangle entry) and the
 * flops / duration of the largest launch. */
int lgp_profile_begin(lgp_ctx *ctx);
int lgp_profile_end(lgp_ctx *ctx, size_t *launches, double *total_ms, double *total_flops,
                    double *max_flops, double *max_ms);

/* ---- Gram build: _GP/_elements.py:554-579 -> _Kernel/_crosskernel.py:192-200 ------- */

#define LGP_KERNEL_EXPQUAD 0   /* _kernels/_basic.py:61-75   exp(-1/2 r2)                        */
#define LGP_KERNEL_MATERNP 1   /* _kernels/_matern.py:29-49 + _special/_bessel.py:101-110 (p)  */
#define LGP_KERNEL_GAMMAEXP 2  /* _kernels/_basic.py:94-115   exp(-r2^(gamma/2)); hpar = {gamma, gamma/2}                              */
#define LGP_KERNEL_CAUCHY 3    /* _kernels/_basic.py:315-344  (1 + r2^(alpha/2)/beta)^(-beta/alpha); hpar = {alpha, beta, -beta/alpha}  */
#define LGP_KERNEL_EXPQUAD_DIFF 4 /* _Kernel/_ops.py:115-211 (`diff`) on ExpQuad: along field f, a_f derivatives in x and b_f in y; hpar[f]
                                   * = a_f + b_f <= 4 for each of the d <= 8 fields (npar = 0, d = 1: p is the order); the core is
                                   * prod_f He_{hpar[f]}(u_f) exp(-|u|^2/2), u = (x - y)/scale; the factor prod_f (-1)^{a_f} / scale^order
                                   * of the chain rule is an output operation of the caller                                       */
#define LGP_KERNEL_MATERNP_DIFF 5 /* the same on Maternp (_kernels/_matern.py:29-49, derivable p times), one-dimensional points: the
                                   * derivative of total order q of exp(-t) P_p(t), t = sqrt(2p+1) |u|, is (c sign u)^q exp(-t) Q_q(t);
                                   * hpar = {q mod 2, the p + 1 coefficients of Q_q in ascending powers}; c^q and the chain-rule
                                   * factors are an output operation of the caller                                                */
#define LGP_KERNEL_WENDLAND 6     /* _kernels/_wendland.py:32-84 (1 - r)_+^(nu+k) poly(r), input='posabs'; p = k in [0, 3],
                                   * hpar = {nu + k, the k + 1 coefficients of poly, highest power first}                         */
#define LGP_KERNEL_CAUSALEXPQUAD 7 /* _kernels/_basic.py:346-360 erfc(alpha/4 r) exp(-r^2/2), input='posabs'; hpar = {alpha / 4}  */
#define LGP_KERNEL_LOG 8          /* _kernels/_basic.py:381-392 log1p(r) / r, input='posabs'                                     */
#define LGP_KERNEL_CONSTANT 9     /* _kernels/_basic.py:34-46   1                                                                 */
#define LGP_KERNEL_WHITE 10       /* _kernels/_basic.py:48-59   1 where every field of x equals that of y, else 0                */
/* stationary kernels of ONE variable (maxdim = 1, d must be 1): functions of the signed difference delta = x - y */
#define LGP_KERNEL_EXPON 11       /* _kernels/_basic.py:254-269 exp(-|delta|), input='abs'                                       */
#define LGP_KERNEL_HOLEEFFECT 12  /* _kernels/_basic.py:299-310 (1 - |delta|) exp(-|delta|), input='abs'                          */
#define LGP_KERNEL_PERIODIC 13    /* _kernels/_basic.py:187-208 exp(-2 (sin(delta / 2) / outerscale)^2); hpar = {outerscale}        */
#define LGP_KERNEL_COS 14         /* _kernels/_spectral.py:25-38 cos(delta)                                                       */
#define LGP_KERNEL_SINC 15        /* _kernels/_spectral.py:113-122 + _special/_sinc.py:25-32 sinc(|delta| + eps), input='posabs'   */
#define LGP_KERNEL_LINEAR 16      /* _kernels/_basic.py:76-91   sum_f x_f y_f (any fields)                                        */
/* kernels of x and y themselves, ONE variable (maxdim = 1, d must be 1) */
#define LGP_KERNEL_WIENER 17            /* _kernels/_randomwalk.py:26-41   min(x, y)                                             */
#define LGP_KERNEL_ORNSTEINUHLENBECK 18 /* _kernels/_randomwalk.py:112-132 exp(-|x - y|) - exp(-(x + y))                          */
#define LGP_KERNEL_BROWNIANBRIDGE 19    /* _kernels/_randomwalk.py:134-151 min(x, y) - x y                                       */
#define LGP_KERNEL_DECAYING 20          /* _kernels/_basic.py:362-379      1 / (x + y + 1)^alpha; hpar = {alpha}                 */
#define LGP_KERNEL_FRACBROWNIAN 21      /* _kernels/_randomwalk.py:46-68   2^-K ((|x|^2H + |y|^2H)^K - |x - y|^2HK); hpar = {2H, K, 2^-K, 2HK} */
#define LGP_KERNEL_WIENERINTEGRAL 22    /* _kernels/_randomwalk.py:93-110  a^2 (b - a / 3) / 2, a = min(x, y), b = max(x, y)      */
#define LGP_KERNEL_STATIONARYFRACBROWNIAN 23 /* _kernels/_randomwalk.py:156-172 (|d + 1|^2H + |d - 1|^2H - 2 |d|^2H) / 2, d = x - y; hpar = {2H} */
#define LGP_KERNEL_CELERITE 24    /* _kernels/_celerite.py:34-52  exp(-gamma |d|) (cos d + B sin |d|); hpar = {gamma, B}, one variable    */
#define LGP_KERNEL_HARMONIC 25    /* _kernels/_celerite.py:54-121 damped driven oscillator; hpar = {Q, eta Q, sqrt(1 - Q^2) - 1}, p = branch */
#define LGP_KERNEL_CIRCULAR 26    /* _kernels/_wendland.py:88-113 W_c of the geodesic distance on the unit circle; hpar = {tau, c}          */
#define LGP_KERNEL_TAYLOR 27      /* _kernels/_basic.py:393-408  I0(2 sqrt(x y)) for x y >= 0, J0(2 sqrt(-x y)) otherwise; one variable                   */
#define LGP_KERNEL_GIBBS 28       /* _kernels/_basic.py:156-185  sqrt(2 s s' / (s^2 + s'^2)) exp(-|x - y|^2 / (s^2 + s'^2)): the LAST coordinate row holds s = scalefun(x), evaluated by the caller; no loc / scale */
#define LGP_KERNEL_MATERN 29      /* _kernels/_matern.py:56-76  2^(1-nu)/Gamma(nu) x^nu K_nu(x), x = sqrt(2 nu) r, real nu in [0, 100]: hpar = {nu}; p ignored.  K_nu by Temme's series / Steed's continued fraction + upward recurrence; nu = 0 is 1 where r = 0.  (Half-integer nu: LGP_KERNEL_MATERNP is the closed form.)  Has a scale tangent. */
#define LGP_KERNEL_LAGTABLE 30    /* _kernels/_arma.py:31-63,180-189  values tabulated over the integer lag |x - y| (MA, AR): only through lgp_gram_lagtable / lgp_pairs_lagtable, which take the table */
#define LGP_KERNEL_COLOR 31       /* _kernels/_spectral.py:73-111  (n - 1) Re E_n(-i |x - y|), integer n >= 2 passed as p: power spectrum 1/omega^n above omega = 1.  E_n by its power series (|x - y| <= 2) and its continued fraction.  Has a scale tangent (-(n - 1) Im E_{n-1}) */
#define LGP_KERNEL_PINK 32        /* _kernels/_spectral.py:41-66   (Ci(D (1 + dw)) - Ci(D)) / log(1 + dw), D = |x - y|; cos(D (1 + dw/2)) where D dw < sqrt(eps): hpar = {dw}.  Has a scale tangent (Ci' = cos x / x) */
#define LGP_KERNEL_BESSEL 33      /* _kernels/_matern.py:102-119   Gamma(nu + 1) (s r / 2)^-nu J_nu(s r), s = 2 + nu / 2, real nu in [0, 100]: hpar = {nu}; at most 2 (floor(nu) + 1) fields.  J_nu by series / Miller's recurrence / Hankel's expansion.  Has a scale tangent (through the kernel of order nu + 1) */
#define LGP_KERNEL_NNKERNEL 35    /* _kernels/_basic.py:122-154  2/pi asin(2 (q + x.y) / ((1 + 2 (q + x.x)) (1 + 2 (q + y.y)))), q = sigma0^2: hpar = {q}.  Derivative entry points only, lgp_gram_deriv_f64 / lgp_gram_deriv_f32: the values are a composite of Linear and per-point factors on the host side */
#define LGP_KERNEL_PSERIES 34     /* even 1-periodic functions of |x - y| from host-computed coefficients (the Zeta kernel, _kernels/_zeta.py:37-70): only through lgp_gram_pseries / lgp_pairs_pseries */
#define LGP_OP_MUL 0           /* kernel * c  (_Kernel/_alg.py:58-82)  */
#define LGP_OP_ADD 1           /* kernel + c  (_Kernel/_alg.py:32-56)  */

/* out[i,j] = ops(core(r2_ij)), r2_ij = sum_f (u_if - v_jf)^2 accumulated in field order
 * (_Kernel/_isotropic.py:61-81, _Kernel/_util.py:74-99), u = (x - loc)/scale, v = (y - loc)/scale
 * (_Kernel/_ops.py:292-326; has_loc/has_scale = 0 skips the operation like loc=None/scale=None).
 * dx: d field arrays of n values, SoA: field f at dx + f*ldx  (columnar StructuredArray, _array.py:104-105)
 * dy: same for the m column points.  out: n x m, ld ldo.
 * hpar (HOST array of npar values, NULL/0 for ExpQuad and Maternp): the real parameters of the kernel and the
 * exponents the reference forms from them as Python floats (so that float32 kernels round them the same way).
 * nops output algops (hops_code/hops_val are HOST arrays, nops <= 8) applied in order.
 * symmetric != 0 (requires dx == dy, n == m): evaluate tiles on/below the diagonal and mirror them
 * (the `halfmatrix` option of _elements.py:567-573). */
int lgp_gram_isotropic_f64(lgp_ctx *ctx, int kernel, int p, int npar, const double *hpar, int d,
                           const double *dx, size_t ldx, size_t n,
                           const double *dy, size_t ldy, size_t m,
                           int has_loc, double loc, int has_scale, double scale,
                           int nops, const int *hops_code, const double *hops_val,
                           int symmetric, double *dout, size_t ldo);
int lgp_gram_isotropic_f32(lgp_ctx *ctx, int kernel, int p, int npar, const float *hpar, int d,
                           const float *dx, size_t ldx, size_t n,
                           const float *dy, size_t ldy, size_t m,
                           int has_loc, float loc, int has_scale, float scale,
                           int nops, const int *hops_code, const float *hops_val,
                           int symmetric, float *dout, size_t ldo);

/* One block column of the REGULARISED Gram matrix, for the sharded factorisation (each rank builds its own block columns in place;
 * _GP/_elements.py:554-579 + _linalg/_decomp.py:384-387 in one pass): dblock (n x w, ld ldb) receives the columns [c0, c0 + w) of
 * tril(K~), K~ = (kernel(x, x) + diag(noise)) / s / s', zeros above the diagonal; dsvec (n) the scales s of ALL rows (every rank
 * computes the same vector: K_ii = k(0) + noise_i needs the diagonal alone); dsums (w) the Gershgorin sums sum_i |K~_ij| of these
 * columns over ALL n rows (the block column is evaluated in full for them: no other rank holds the mirrored entries).  Kernels:
 * LGP_KERNEL_EXPQUAD, MATERNP, GAMMAEXP, CAUCHY, as lgp_gram_chol_factor_rhs.  noise: scalar, or dnoise (device, n).
 * Block-cyclic storage in ONE launch (cyc_nb > 0, a multiple of the tile width 512 (f64) / 1024 (f32)): the w local columns are
 * blocks of cyc_nb columns, local block q holding the global columns [c0 + q cyc_stride, c0 + q cyc_stride + cyc_nb) -- a rank's
 * whole share of the matrix (c0 = rank nb, cyc_stride = nranks nb); cyc_nb = 0: the plain range [c0, c0 + w). */
int lgp_gram_reg_cols_f64(lgp_ctx *ctx, int kernel, int p, int npar, const double *hpar, int d,
                          const double *dx, size_t ldx, size_t n, size_t c0, size_t w, size_t cyc_nb, size_t cyc_stride,
                          int has_loc, double loc, int has_scale, double scale,
                          int nops, const int *hops_code, const double *hops_val,
                          double noise, const double *dnoise, double *dblock, size_t ldb, double *dsvec, double *dsums);
int lgp_gram_reg_cols_f32(lgp_ctx *ctx, int kernel, int p, int npar, const float *hpar, int d,
                          const float *dx, size_t ldx, size_t n, size_t c0, size_t w, size_t cyc_nb, size_t cyc_stride,
                          int has_loc, float loc, int has_scale, float scale,
                          int nops, const int *hops_code, const float *hops_val,
                          float noise, const float *dnoise, float *dblock, size_t ldb, float *dsvec, float *dsums);

/* Stationary kernels of one integer variable given by a table of their values (the reference's discrete-time kernels MA and AR,
 * _kernels/_arma.py:31-63 `cov.at[delta + len(w) - 1].get(mode='fill', fill_value=0)` and :180-189 `acf.at[delta].get(mode='fill',
 * fill_value=nan)`): out[i][j] = dtable[|x_i - y_j|] where the lag is below ntable, `fill` elsewhere.  dtable: DEVICE array.  The
 * lags are integers (the reference indexes with them); a fractional lag is truncated.  Points, loc / scale, output operations and
 * `symmetric` as for lgp_gram_isotropic with d = 1. */
int lgp_gram_lagtable_f64(lgp_ctx *ctx, const double *dtable, size_t ntable, double fill, const double *dx, size_t ldx, size_t n,
                          const double *dy, size_t ldy, size_t m, int has_loc, double loc, int has_scale, double scale,
                          int nops, const int *hops_code, const double *hops_val, int symmetric, double *dout, size_t ldo);
int lgp_gram_lagtable_f32(lgp_ctx *ctx, const float *dtable, size_t ntable, float fill, const float *dx, size_t ldx, size_t n,
                          const float *dy, size_t ldy, size_t m, int has_loc, float loc, int has_scale, float scale,
                          int nops, const int *hops_code, const float *hops_val, int symmetric, float *dout, size_t ldo);
int lgp_pairs_lagtable_f64(lgp_ctx *ctx, const double *dtable, size_t ntable, double fill, const double *dx, size_t ldx,
                           const double *dy, size_t ldy, size_t npairs, int has_loc, double loc, int has_scale, double scale,
                           int nops, const int *hops_code, const double *hops_val, double *dout);
int lgp_pairs_lagtable_f32(lgp_ctx *ctx, const float *dtable, size_t ntable, float fill, const float *dx, size_t ldx,
                           const float *dy, size_t ldy, size_t npairs, int has_loc, float loc, int has_scale, float scale,
                           int nops, const int *hops_code, const float *hops_val, float *dout);

/* Even, 1-periodic stationary kernels of one variable from host-computed coefficients -- the evaluation half of the reference's Zeta
 * kernel, k(D) = Re Li_s(e^{2 pi i D}) / zeta(s), s = 1 + 2 nu (_kernels/_zeta.py:37-70, _special/_zeta.py:76-177).  With
 * theta = 2 pi (|x - y| mod 1, reflected into [0, 1/2]):
 *   mode 0: sum_{j=1..ntable} dtable[j-1] cos(j theta)                  (the Fourier series itself: s >= 10, table j^-s / zeta(s))
 *   mode 1: sum_{n<ntable} dtable[n] theta^(2n) + theta^pw (A g + D),  g = 1 (uselog 0), log theta (uselog 1, eps = 0) or
 *           expm1(eps log theta) (uselog 1, eps != 0)                   (the polylogarithm's expansion on the unit circle,
 *           table (-1)^n zeta(s-2n) / ((2n)! zeta(s)), A theta^(s-1) its non-analytic term, regrouped near odd s)
 *   mode 2: 1 where theta = 0, else 0                                   (nu = 0)
 * hpar4 = {A, pw, eps, D} (host); dtable: DEVICE array.  lsqfitgp_amd/_zeta.py computes the coefficients for Zeta(nu). */
int lgp_gram_pseries_f64(lgp_ctx *ctx, int mode, int uselog, const double *dtable, size_t ntable, const double *hpar4,
                         const double *dx, size_t ldx, size_t n, const double *dy, size_t ldy, size_t m,
                         int has_loc, double loc, int has_scale, double scale,
                         int nops, const int *hops_code, const double *hops_val, int symmetric, double *dout, size_t ldo);
int lgp_gram_pseries_f32(lgp_ctx *ctx, int mode, int uselog, const float *dtable, size_t ntable, const float *hpar4,
                         const float *dx, size_t ldx, size_t n, const float *dy, size_t ldy, size_t m,
                         int has_loc, float loc, int has_scale, float scale,
                         int nops, const int *hops_code, const float *hops_val, int symmetric, float *dout, size_t ldo);
int lgp_pairs_pseries_f64(lgp_ctx *ctx, int mode, int uselog, const double *dtable, size_t ntable, const double *hpar4,
                          const double *dx, size_t ldx, const double *dy, size_t ldy, size_t npairs,
                          int has_loc, double loc, int has_scale, double scale,
                          int nops, const int *hops_code, const double *hops_val, double *dout);
int lgp_pairs_pseries_f32(lgp_ctx *ctx, int mode, int uselog, const float *dtable, size_t ntable, const float *hpar4,
                          const float *dx, size_t ldx, const float *dy, size_t ldy, size_t npairs,
                          int has_loc, float loc, int has_scale, float scale,
                          int nops, const int *hops_code, const float *hops_val, float *dout);

/* BART correlation Gram matrix, the `altinput=True` path of _kernels/_bart.py:301-455,628-757 for one
 * recursion slice ("bracket") of `size` = 1, 2 or 3 probabilities, iterated `repeat` times (the folding of
 * `reset` depths, :424-447; e.g. maxd=10, reset=[2,4,6,8] is ONE slice with size 3, repeat 5).
 * dix: n x p int32 row-major bin indices (one row per point), diy: m x p; hnsplits: HOST p totals
 * (splits[0]); hweights: HOST p weights; hprobs: HOST repeat*size probabilities laid out as _bart.py:436-447
 * does (rows evaluated first to last); gamma: scalar interpolation coefficient, or dgamma_mat (n x m, ld ldo,
 * may alias dout) = per-pair result of the previous, deeper slice (:434-455).
 * dpsi: device table psi[k] = digamma(k), k = 1..npsi-1, npsi >= max(nsplits)+2 (needed for size 3, :735-744).
 * The reference tests point equality by a 64-bit hash (:674-681); here the index vectors are compared exactly. */
int lgp_gram_bart_f32(lgp_ctx *ctx, int p, const int32_t *dix, size_t n, const int32_t *diy, size_t m,
                      const int32_t *hnsplits, const float *hweights,
                      int size, int repeat, const float *hprobs, float gamma, const float *dgamma_mat,
                      const float *dpsi, size_t npsi,
                      int nops, const int *hops_code, const float *hops_val,
                      float *dout, size_t ldo);
int lgp_gram_bart_f64(lgp_ctx *ctx, int p, const int32_t *dix, size_t n, const int32_t *diy, size_t m,
                      const int32_t *hnsplits, const double *hweights,
                      int size, int repeat, const double *hprobs, double gamma, const double *dgamma_mat,
                      const double *dpsi, size_t npsi,
                      int nops, const int *hops_code, const double *hops_val,
                      double *dout, size_t ldo);

/* Gram block of a DERIVED kernel: d^ax/dx^ax d^by/dy^by k(x, y) along field fx of x and field fy of y (-1 with order 0), orders
 * 0 .. 2 -- what `kernel.linop('diff', xderiv, yderiv)` evaluates through jax.grad of the core (_Kernel/_ops.py:115-211) when points
 * are added with `deriv=` (_GP/_elements.py:158-242).  The core runs once in truncated Taylor arithmetic (csrc/gram_jet.hip).
 * kernel: LGP_KERNEL_EXPQUAD, GAMMAEXP (gamma = 2), CAUCHY (alpha = 2; hpar = {alpha/2, beta, -beta/alpha}), MATERNP (hcoef: five
 * rows of 24 Laurent coefficients of t^-7 .. t^16), CONSTANT, LINEAR, and of one variable (d = 1): COS, PERIODIC (hpar as
 * lgp_gram_isotropic), SINC, CELERITE, HARMONIC, WENDLAND (hpar = {nu + k}, hcoef = the polynomial), CIRCULAR, DECAYING,
 * WIENERINTEGRAL, FRACBROWNIAN (H = K = 1), STATIONARYFRACBROWNIAN (H = 1).  factor: product of the multiplicative output
 * operations; the chain-rule factor 1 / scale^(ax + by) is applied by the library. */
int lgp_gram_deriv_f64(lgp_ctx *ctx, int kernel, int p, int npar, const double *hpar, int ncoef, const double *hcoef, int d,
                       const double *dx, size_t ldx, size_t n, const double *dy, size_t ldy, size_t m, int has_loc, double loc,
                       int has_scale, double scale, double factor, int fx, int ax, int fy, int by, double *dout, size_t ldo);
int lgp_gram_deriv_f32(lgp_ctx *ctx, int kernel, int p, int npar, const float *hpar, int ncoef, const float *hcoef, int d,
                       const float *dx, size_t ldx, size_t n, const float *dy, size_t ldy, size_t m, int has_loc, float loc,
                       int has_scale, float scale, float factor, int fx, int ax, int fy, int by, float *dout, size_t ldo);

/* The isotropic Gram matrix together with its derivative with respect to `scale` (dout and ddscale: n x m, ld ldo): the
 * forward-mode tangent of `(x - loc) / scale` (_Kernel/_ops.py:310-326) through the cores and the output algops -- what
 * jax.jacfwd of a gpfactory with a `scale` hyperparameter produces in the reference's fit (_fit.py:679-690), here in the
 * same pass over the pairs as the matrix itself.  scale=None: the derivative at scale = 1. */
int lgp_gram_isotropic_dscale_f64(lgp_ctx *ctx, int kernel, int p, int npar, const double *hpar, int d,
                                  const double *dx, size_t ldx, size_t n, const double *dy, size_t ldy, size_t m,
                                  int has_loc, double loc, int has_scale, double scale,
                                  int nops, const int *hops_code, const double *hops_val,
                                  double *dout, double *ddscale, size_t ldo);
int lgp_gram_isotropic_dscale_f32(lgp_ctx *ctx, int kernel, int p, int npar, const float *hpar, int d,
                                  const float *dx, size_t ldx, size_t n, const float *dy, size_t ldy, size_t m,
                                  int has_loc, float loc, int has_scale, float scale,
                                  int nops, const int *hops_code, const float *hops_val,
                                  float *dout, float *ddscale, size_t ldo);
/* The BART Gram matrix with ntan <= 2 forward-mode tangents of the splitting probabilities (alpha and beta of
 * bayestree.bart, bayestree/_bart.py:187-205): arguments as lgp_gram_bart_*, plus hdprobs (HOST, ntan x repeat*size:
 * d hprobs / d theta_t, laid out like hprobs), hdtan (HOST array of ntan DEVICE n x m outputs, ld ldo: d dout / d theta_t)
 * and hdgtan (HOST array of ntan DEVICE per-pair tangents of the deeper slice's result, may alias hdtan; NULL = zero). */
int lgp_gram_bart_jvp_f32(lgp_ctx *ctx, int p, const int32_t *dix, size_t n, const int32_t *diy, size_t m,
                          const int32_t *hnsplits, const float *hweights,
                          int size, int repeat, const float *hprobs, float gamma, const float *dgamma_mat,
                          const float *dpsi, size_t npsi, int nops, const int *hops_code, const float *hops_val,
                          float *dout, size_t ldo, int ntan, const float *hdprobs, float *const *hdtan,
                          const float *const *hdgtan);
int lgp_gram_bart_jvp_f64(lgp_ctx *ctx, int p, const int32_t *dix, size_t n, const int32_t *diy, size_t m,
                          const int32_t *hnsplits, const double *hweights,
                          int size, int repeat, const double *hprobs, double gamma, const double *dgamma_mat,
                          const double *dpsi, size_t npsi, int nops, const int *hops_code, const double *hops_val,
                          double *dout, size_t ldo, int ntan, const double *hdprobs, double *const *hdtan,
                          const double *const *hdgtan);

/* Elementwise ("pairs") forms of the same kernels: out[i] = k(x_i, y_i), i < npairs -- the broadcasting
 * call contract of CrossKernel.__call__ for equally shaped inputs (_Kernel/_crosskernel.py:192-200) and the
 * flat upper-triangle evaluation of `halfmatrix` (_GP/_elements.py:567-573).  Arguments as the Gram forms.
 * BART: dnpair (npairs x p int32, or NULL to use hnsplits for every pair) gives per-pair split totals, which
 * makes it the fully vectorised BART.correlation(..., altinput=True) of _kernels/_bart.py:301-455;
 * dgamma_vec (npairs, may alias dout) chains recursion slices. */
int lgp_pairs_isotropic_f64(lgp_ctx *ctx, int kernel, int p, int npar, const double *hpar, int d, const double *dx, size_t ldx,
                            const double *dy, size_t ldy, size_t npairs,
                            int has_loc, double loc, int has_scale, double scale,
                            int nops, const int *hops_code, const double *hops_val, double *dout);
int lgp_pairs_isotropic_f32(lgp_ctx *ctx, int kernel, int p, int npar, const float *hpar, int d, const float *dx, size_t ldx,
                            const float *dy, size_t ldy, size_t npairs,
                            int has_loc, float loc, int has_scale, float scale,
                            int nops, const int *hops_code, const float *hops_val, float *dout);
int lgp_pairs_bart_f32(lgp_ctx *ctx, int p, const int32_t *dix, const int32_t *diy, size_t npairs,
                       const int32_t *hnsplits, const int32_t *dnpair, const float *hweights,
                       int size, int repeat, const float *hprobs, float gamma, const float *dgamma_vec,
                       const float *dpsi, size_t npsi,
                       int nops, const int *hops_code, const float *hops_val, float *dout);
int lgp_pairs_bart_f64(lgp_ctx *ctx, int p, const int32_t *dix, const int32_t *diy, size_t npairs,
                       const int32_t *hnsplits, const int32_t *dnpair, const double *hweights,
                       int size, int repeat, const double *hprobs, double gamma, const double *dgamma_vec,
                       const double *dpsi, size_t npsi,
                       int nops, const int *hops_code, const double *hops_val, double *dout);

/* ---- elementwise helpers used by the GP facade ---------------------------------- */

/* C = alpha*A + beta*B elementwise (rows x cols); `Kxsxs - Q` of _GP/_compute.py:260, `Kxx + ycov` :85 */
int lgp_axpby_f64(lgp_ctx *ctx, size_t rows, size_t cols, double alpha, const double *dA, size_t lda,
                  double beta, const double *dB, size_t ldb, double *dC, size_t ldc);
int lgp_axpby_f32(lgp_ctx *ctx, size_t rows, size_t cols, float alpha, const float *dA, size_t lda,
                  float beta, const float *dB, size_t ldb, float *dC, size_t ldc);
/* A[i,i] += dvec[i]*alpha (dvec may be NULL: += alpha); diag noise of bayestree/_bart.py:196 */
int lgp_add_diag_f64(lgp_ctx *ctx, size_t n, double *dA, size_t lda, const double *dvec, double alpha);
int lgp_add_diag_f32(lgp_ctx *ctx, size_t n, float *dA, size_t lda, const float *dvec, float alpha);
/* A += c elementwise (`k_sigma_mu^2 * 11^T`) */
int lgp_add_scalar_f64(lgp_ctx *ctx, size_t rows, size_t cols, double *dA, size_t lda, double c);
int lgp_add_scalar_f32(lgp_ctx *ctx, size_t rows, size_t cols, float *dA, size_t lda, float c);
/* C = alpha (A o B) + beta C0, entry by entry (C0 may be NULL, or C itself): the product of two kernels evaluated on the same
 * points (`k1 * k2`, reference _Kernel/_alg.py:58-82) and, accumulated, the product rule of its hyperparameter tangents. */
int lgp_mul_f64(lgp_ctx *ctx, size_t rows, size_t cols, double alpha, const double *dA, size_t lda, const double *dB, size_t ldb,
                double beta, const double *dC0, size_t ldc0, double *dC, size_t ldc);
int lgp_mul_f32(lgp_ctx *ctx, size_t rows, size_t cols, float alpha, const float *dA, size_t lda, const float *dB, size_t ldb,
                float beta, const float *dC0, size_t ldc0, float *dC, size_t ldc);
/* A <- a f(A) + b entry by entry; op 0: f = asin.  The outer function of a kernel that is a function of another kernel's value
 * (NNKernel, reference _kernels/_basic.py:122-154). */
#define LGP_UNARY_ASIN 0
int lgp_unary_f64(lgp_ctx *ctx, size_t rows, size_t cols, double *dA, size_t lda, int op, double a, double b);
int lgp_unary_f32(lgp_ctx *ctx, size_t rows, size_t cols, float *dA, size_t lda, int op, float a, float b);
/* flags[0] = any non-finite entry, flags[1] = max |A - A^T| > atol + rtol*|A^T| anywhere (jnp.allclose),
 * checks of _GP/_elements.py:618-622 and _GP/_compute.py:369-381.  Synchronises. */
int lgp_check_matrix_f64(lgp_ctx *ctx, size_t rows, size_t cols, const double *dA, size_t lda,
                         int check_sym, double rtol, double atol, int *hflags);
int lgp_check_matrix_f32(lgp_ctx *ctx, size_t rows, size_t cols, const float *dA, size_t lda,
                         int check_sym, float rtol, float atol, int *hflags);
/* The same two checks without the read-back: the flags are OR-ed into two int words in DEVICE memory that the
 * caller has zeroed, so that a caller assembling a matrix from many blocks (_GP/_elements.py:642-649) reads all
 * the verdicts at once.  Does not synchronise. */
int lgp_check_matrix_dev_f64(lgp_ctx *ctx, size_t rows, size_t cols, const double *dA, size_t lda,
                             int check_sym, double rtol, double atol, int *dflags);
int lgp_check_matrix_dev_f32(lgp_ctx *ctx, size_t rows, size_t cols, const float *dA, size_t lda,
                             int check_sym, float rtol, float atol, int *dflags);

/* *hclose = 1 iff |A - B| <= atol + rtol*|B| everywhere (jnp.allclose(A, B), the off-diagonal symmetry
 * check `allclose(block.T, blockT)` of _GP/_elements.py:631-636).  Synchronises. */
int lgp_allclose_f64(lgp_ctx *ctx, size_t rows, size_t cols, const double *dA, size_t lda,
                     const double *dB, size_t ldb, double rtol, double atol, int *hclose);
int lgp_allclose_f32(lgp_ctx *ctx, size_t rows, size_t cols, const float *dA, size_t lda,
                     const float *dB, size_t ldb, float rtol, float atol, int *hclose);

/* ---- decomposition: _linalg/_decomp.py:376-393 ----------------------------------- */

#define LGP_EPS_AUTO_REL 1   /* epsrel='auto' -> n * finfo.eps   (_decomp.py:248-249) */
#define LGP_EPS_AUTO_ABS 2   /* epsabs='auto' -> finfo.eps       (_decomp.py:250-251) */
#define LGP_FACTOR_LOWER_ONLY 4 /* lgp_gram_chol_factor_rhs_* only: the entries of dL above the diagonal are left UNSPECIFIED outside the
                                 * 128 x 128 diagonal blocks (no stores of zeros: half of the Gram kernel's traffic).  Every entry point of this
                                 * library that takes a factor reads its lower triangle and the diagonal blocks only; lgp_zero_upper_* fills the
                                 * zeros in afterwards for a consumer that multiplies by the full matrix (Decomposition.correlate). */

/* Chol.__init__: s = diag_scale_pow2(K) (:356-361); Kt = K/s/s[:,None]; eps = epsrel*max_i sum_j|Kt_ij| + epsabs
 * (:245-255,349-354); Kt_ii += eps; Lt = cholesky(Kt) lower; L = Lt*s[:,None]; eps_out = eps*min(s^2).
 * dK: n x n symmetric (only read).  dL: n x n output, strict upper triangle zeroed.
 * ddblk: workspace/output, lgp_chol_dblk_size(n) elements: packed copies (block, transposed block, reciprocal
 * diagonal) of the 128x128 diagonal blocks of L, read by the substitution kernels of lgp_trsm_*.  Returns 0, or info > 0 = 1-based index of the first non-positive or non-finite
 * pivot (the reference's "cholesky decomposition not finite" LinAlgError, :389-391).  Synchronises. */
size_t lgp_chol_dblk_size(size_t n);
int lgp_chol_factor_f64(lgp_ctx *ctx, size_t n, const double *dK, size_t ldk,
                        double epsrel, double epsabs, int eps_flags,
                        double *dL, size_t ldl, double *ddblk, double *heps_out);
int lgp_chol_factor_f32(lgp_ctx *ctx, size_t n, const float *dK, size_t ldk,
                        float epsrel, float epsabs, int eps_flags,
                        float *dL, size_t ldl, float *ddblk, float *heps_out);
/* the bare blocked factorisation (jlinalg.cholesky, _decomp.py:388) of the lower triangle of dL in place */
int lgp_potrf_f64(lgp_ctx *ctx, size_t n, double *dL, size_t ldl, double *ddblk);
int lgp_potrf_f32(lgp_ctx *ctx, size_t n, float *dL, size_t ldl, float *ddblk);

/* ---- solves and products with L: _linalg/_decomp.py:398-439 ---------------------- */

/* B <- L^-1 B (trans=0) or L^-T B (trans=1), B n x nrhs in place: jlinalg.solve_triangular(L, B, lower=True)
 * (_decomp.py:402,407,408,414,419,426,439,467) and solve_triangular(L.T, ., lower=False) (:403,469) */
int lgp_trsm_f64(lgp_ctx *ctx, size_t n, const double *dL, size_t ldl, const double *ddblk,
                 int trans, double *dB, size_t ldb, size_t nrhs);
int lgp_trsm_f32(lgp_ctx *ctx, size_t n, const float *dL, size_t ldl, const float *ddblk,
                 int trans, float *dB, size_t ldb, size_t nrhs);
/* C (m x n) = alpha * op(A) op(B) + beta * C;  ta/tb = 0: as stored, 1: transposed.
 * op(A) is m x k, op(B) is k x n.  `invLA.T @ invLr` (:409), `invLA.T @ invLA` (:420),
 * `L @ x` (:431), `L.T @ X` (:435), `A.T @ ycov @ A` (_compute.py:269) */
int lgp_gemm_f64(lgp_ctx *ctx, int ta, int tb, size_t m, size_t n, size_t k, double alpha,
                 const double *dA, size_t lda, const double *dB, size_t ldb, double beta, double *dC, size_t ldc);
int lgp_gemm_f32(lgp_ctx *ctx, int ta, int tb, size_t m, size_t n, size_t k, float alpha,
                 const float *dA, size_t lda, const float *dB, size_t ldb, float beta, float *dC, size_t ldc);
/* out[j] = sum_i A[i,j]^2 : einsum('ji,ji->i') of _decomp.py:427 */
int lgp_colsumsq_f64(lgp_ctx *ctx, size_t rows, size_t cols, const double *dA, size_t lda, double *dout);
int lgp_colsumsq_f32(lgp_ctx *ctx, size_t rows, size_t cols, const float *dA, size_t lda, float *dout);
/* sum_i log L_ii (_decomp.py:486) and x.y (:487); results to host, synchronise */
int lgp_logdet_diag_f64(lgp_ctx *ctx, size_t n, const double *dL, size_t ldl, double *hout);
int lgp_logdet_diag_f32(lgp_ctx *ctx, size_t n, const float *dL, size_t ldl, float *hout);
int lgp_dot_f64(lgp_ctx *ctx, size_t n, const double *dx, const double *dy, double *hout);
int lgp_dot_f32(lgp_ctx *ctx, size_t n, const float *dx, const float *dy, float *hout);

/* ---- fused conveniences for GP.pred / GP.marginal_likelihood --------------------- */

/* _GP/_compute.py:259-260 with L^-1 Kxxs computed once:
 * dW: n x (m+1) workspace, on entry [Kxxs | y] (column m = y), overwritten with L^-1 [Kxxs | y];
 * mean (m) = (L^-1 Kxxs)^T (L^-1 y); cov (m x m, ld ldc) = Kxsxs - (L^-1 Kxxs)^T (L^-1 Kxxs).
 * dKxsxs may alias dcov.  dmean/dcov are device pointers.  hmlnd (host, may be NULL): if given, also returns
 * minus_log_normal_density(y, value=True) (_decomp.py:484-488) from the already solved column L^-1 y, so that a
 * fit+predict pass needs one forward solve in all; synchronises in that case. */
int lgp_posterior_f64(lgp_ctx *ctx, size_t n, size_t m, const double *dL, size_t ldl, const double *ddblk,
                      double *dW, size_t ldw, const double *dKxsxs, size_t ldk,
                      double *dmean, double *dcov, size_t ldc, double *hmlnd);
int lgp_posterior_f32(lgp_ctx *ctx, size_t n, size_t m, const float *dL, size_t ldl, const float *ddblk,
                      float *dW, size_t ldw, const float *dKxsxs, size_t ldk,
                      float *dmean, float *dcov, size_t ldc, float *hmlnd);
/* The same pass with the forward solve carried through the factorisation.  Chol.__init__ + the L^-1 products of
 * GPCompute.pred (_linalg/_decomp.py:380-393 + :408,:419,:467): dL has n + nrhs rows; rows n .. n+nrhs-1 hold B^T
 * (nrhs x n, e.g. [Kxxs | y]^T) on entry and (L^-1 B)^T on exit, L (rows 0..n-1) being exactly the factor that
 * lgp_chol_factor returns.  The extra rows are "rows below" for every panel of the blocked factorisation, so the
 * solve is done by its MFMA trailing updates instead of a chain of 2n/128 small triangular kernels. */
int lgp_chol_factor_rhs_f64(lgp_ctx *ctx, size_t n, const double *dK, size_t ldk, double epsrel, double epsabs, int eps_flags,
                            double *dL, size_t ldl, size_t nrhs, double *ddblk, double *heps_out);
int lgp_chol_factor_rhs_f32(lgp_ctx *ctx, size_t n, const float *dK, size_t ldk, float epsrel, float epsabs, int eps_flags,
                            float *dL, size_t ldl, size_t nrhs, float *ddblk, float *heps_out);
/* Gram build and Chol.__init__ fused (_GP/_elements.py:575-577 + _GP/_compute.py:85 + _linalg/_decomp.py:380-393): the
 * factor of K = kernel(x, x) + diag(noise) for an isotropic kernel, K itself never materialised.  One pass over the pairs
 * writes tril(K / s / s') into dL and leaves the Gershgorin sums (the separate path writes K, re-reads it for the sums and
 * again for the scaled lower copy); then eps, the blocked factorisation and the rescaling exactly as lgp_chol_factor_rhs_*.
 * Kernel arguments as lgp_gram_isotropic_* (x against itself); noise: scalar added to the diagonal, or dnoise (n device
 * values) if not NULL; dL: (n + nrhs) x n, rows n.. hold B^T on entry like for lgp_chol_factor_rhs_*; nrhs may be 0. */
int lgp_gram_chol_factor_rhs_f64(lgp_ctx *ctx, int kernel, int p, int npar, const double *hpar, int d,
                                 const double *dx, size_t ldx, size_t n,
                                 int has_loc, double loc, int has_scale, double scale,
                                 int nops, const int *hops_code, const double *hops_val,
                                 double noise, const double *dnoise, double epsrel, double epsabs, int eps_flags,
                                 double *dL, size_t ldl, size_t nrhs, double *ddblk, double *heps_out);
int lgp_gram_chol_factor_rhs_f32(lgp_ctx *ctx, int kernel, int p, int npar, const float *hpar, int d,
                                 const float *dx, size_t ldx, size_t n,
                                 int has_loc, float loc, int has_scale, float scale,
                                 int nops, const int *hops_code, const float *hops_val,
                                 float noise, const float *dnoise, float epsrel, float epsabs, int eps_flags,
                                 float *dL, size_t ldl, size_t nrhs, float *ddblk, float *heps_out);
/* The zeros above the diagonal of a factor made with LGP_FACTOR_LOWER_ONLY (dL: n x n).  Asynchronous on the context's stream. */
int lgp_zero_upper_f64(lgp_ctx *ctx, size_t n, double *dL, size_t ldl);
int lgp_zero_upper_f32(lgp_ctx *ctx, size_t n, float *dL, size_t ldl);
/* lgp_posterior for right-hand sides solved that way: dWt = (m+1) x n, rows 0..m-1 = (L^-1 Kxxs)^T, row m = (L^-1 y)^T
 * (typically dL + n * ldl); same outputs as lgp_posterior. */
int lgp_posterior_solved_f64(lgp_ctx *ctx, size_t n, size_t m, const double *dL, size_t ldl, const double *dWt, size_t ldw,
                             const double *dKxsxs, size_t ldk, double *dmean, double *dcov, size_t ldc, double *hmlnd);
int lgp_posterior_solved_f32(lgp_ctx *ctx, size_t n, size_t m, const float *dL, size_t ldl, const float *dWt, size_t ldw,
                             const float *dKxsxs, size_t ldk, float *dmean, float *dcov, size_t ldc, float *hmlnd);
/* GP.pred(fromdata=True, raw=True) (_GP/_compute.py:138-305) for the plain question -- one isotropic kernel, n points conditioned on,
 * m predicted, diagonal noise -- from HOST buffers to HOST results in one call (uploads, Kxxs^T into the carried rows, the fused
 * Gram build + factorisation of lgp_gram_chol_factor_rhs, the posterior of lgp_posterior_solved, one download).
 * hpts: HOST d x (n + m), row f = coordinate f of the n data points followed by the m prediction points.  hy: HOST n.
 * hnoise: HOST n (diagonal of the data covariance) or NULL.
 * dP: DEVICE d x (n + m), receives the points.  dL: DEVICE (n + nx) x n with leading dimension ldl, nx >= m + 1 (whole 128-row tiles
 * are fastest): on exit the factor L (as with LGP_FACTOR_LOWER_ONLY: nothing is promised above its diagonal), and below it rows
 * n .. n + m = (L^-1 [Kxxs | y])^T.  ddblk: lgp_chol_dblk_size(n).
 * dwork: DEVICE m m + m + 2 + n doubles.  hout: HOST m m + m + 2 doubles: the posterior covariance (m x m), the posterior mean (m),
 * minus_log_normal_density(y) (_decomp.py:484-488), and one scratch value.  heps_out: the jitter that was added (_decomp.py:393).
 * Returns 0, info > 0 (first non-positive pivot) or an error.  Synchronises. */
int lgp_predict_isotropic_f64(lgp_ctx *ctx, int kernel, int p, int npar, const double *hpar, int d, const double *hpts, size_t n, size_t m,
                              int has_loc, double loc, int has_scale, double scale, int nops, const int *hops_code, const double *hops_val,
                              const double *hy, const double *hnoise, double *dP, double *dL, size_t ldl, size_t nx, double *ddblk,
                              double *dwork, double *hout, double *heps_out);
/* minus_log_normal_density(r, value=True) (_decomp.py:466-488): 1/2 (n log 2pi + 2 sum log L_ii + |L^-1 r|^2).
 * dr (n) is overwritten with L^-1 r.  Synchronises. */
int lgp_mlnd_value_f64(lgp_ctx *ctx, size_t n, const double *dL, size_t ldl, const double *ddblk,
                       double *dr, double *hvalue);
int lgp_mlnd_value_f32(lgp_ctx *ctx, size_t n, const float *dL, size_t ldl, const float *ddblk,
                       float *dr, float *hvalue);

/* minus_log_normal_density(r, dK=..., dr=..., value, gradfwd, fisher) (_decomp.py:441-561): the value, its forward-mode
 * gradient 1/2 tr(K^-1 dK_k) - 1/2 r'K^-1 dK_k K^-1 r + r'K^-1 dr_k and the Fisher matrix 1/2 tr(K^-1 dK_k K^-1 dK_q) +
 * dr_k' K^-1 dr_q of the p parameters, computed on the device from the factor: K^-1 is formed once (as the reference
 * does, :470-472), every contraction is an MFMA gemm or a reduction kernel; nothing n x n leaves HBM.
 * dr (n, device) is overwritten with L^-1 r.  hdK: HOST array of p DEVICE pointers to the n x n matrices dK_k (ld lddk), or
 * NULL; ddr: n x p device matrix (ld lddr) of dr_k, or NULL.  Outputs (host, each may be NULL): hvalue, hgrad[p],
 * hfisher[p*p] row-major.  Synchronises. */
int lgp_mlnd_grad_f64(lgp_ctx *ctx, size_t n, const double *dL, size_t ldl, const double *ddblk, double *dr,
                      size_t p, const double *const *hdK, size_t lddk, const double *ddr, size_t lddr,
                      double *hvalue, double *hgrad, double *hfisher);
int lgp_mlnd_grad_f32(lgp_ctx *ctx, size_t n, const float *dL, size_t ldl, const float *ddblk, float *dr,
                      size_t p, const float *const *hdK, size_t lddk, const float *ddr, size_t lddr,
                      float *hvalue, float *hgrad, float *hfisher);

/* ---- multi-GPU: 1-D block-column-cyclic Cholesky over RCCL (no reference counterpart) -- */

#define LGP_COMM_ID_BYTES 128
int lgp_comm_unique_id(void *hid128);                                   /* ncclGetUniqueId on rank 0 */
int lgp_comm_init(lgp_ctx *ctx, int rank, int nranks, const void *hid128);  /* ncclCommInitRank */
int lgp_comm_destroy(lgp_ctx *ctx);
/* Rank and number of ranks as the communicator itself reports them (ncclCommUserRank / ncclCommCount), and which
 * transport the context's collectives use; 1 rank and LGP_TRANSPORT_NONE without a communicator. */
#define LGP_TRANSPORT_NONE 0
#define LGP_TRANSPORT_RCCL 1
#define LGP_TRANSPORT_HOST 2
int lgp_comm_info(lgp_ctx *ctx, int *rank, int *nranks, int *transport);
/* Alternative transport (bring your own collective, e.g. MPI or gloo; also how the multi-rank schedule is tested on
 * a single GPU): lgp_bcast / lgp_allreduce_* then stage the buffer through host memory on the context's current
 * stream and call fn(user, op, host_buffer, count, root), op 0 = broadcast `count` bytes from `root`, 1 = all-reduce
 * sum of `count` doubles, 2 = all-reduce max of `count` doubles; fn returns 0 on success.  Only when asked to by
 * lgp_comm_set_bcast(ctx, 2, .): op 5 = point to point, `count` bytes from rank (root & 0xffff) to rank (root >> 16); EVERY
 * rank is called with every transfer, in the same order, and a rank that is neither end leaves hbuf alone. */
int lgp_comm_init_host(lgp_ctx *ctx, int rank, int nranks, int (*fn)(void *user, int op, void *hbuf, size_t count, int root),
                       void *user);
/* Broadcast on the ctx stream.  With three or more ranks a large buffer travels as scatter + all-gather over the pairwise
 * xGMI links: the root sends one of nranks - 1 chunks to each peer, the peers exchange them (two groups of ncclSend /
 * ncclRecv), 2 S / ((G - 1) B_link) instead of the S / B_link of a broadcast that crosses one link per GPU.
 * lgp_comm_set_bcast: algo 0 = auto (RCCL: as above from min_bytes on, default 1 MiB; host transport: its own op 0),
 * 1 = always the transport's broadcast, 2 = always scatter + all-gather (>= 3 ranks).  Env LGP_BCAST_ALGO=auto|plain|scatter
 * and LGP_BCAST_MIN_BYTES set a new context's values.  lgp_bcast_plan (host only): the transfers of such a broadcast, four
 * values each (source, destination, byte offset, length), the *nscatter transfers of the first group first; returns their
 * number (negative: error). */
int lgp_bcast(lgp_ctx *ctx, void *dbuf, size_t bytes, int root);
int lgp_comm_set_bcast(lgp_ctx *ctx, int algo, size_t min_bytes);
long lgp_bcast_plan(size_t bytes, int nranks, int root, unsigned long long *xfers, size_t capacity, size_t *nscatter);
int lgp_allreduce_sum_f64(lgp_ctx *ctx, double *dbuf, size_t count);
int lgp_allreduce_max_f64(lgp_ctx *ctx, double *dbuf, size_t count);

/* The context owns two streams; entry points enqueue on the selected one (0 = main, default; 1 = auxiliary,
 * highest priority).  The distributed driver puts the RCCL broadcasts on the auxiliary stream so that panel
 * transfers overlap the trailing updates, ordering them with events (lgp_event_record / lgp_stream_wait_event). */
int lgp_select_stream(lgp_ctx *ctx, int which);
int lgp_stream_wait_event(lgp_ctx *ctx, void *event);      /* the selected stream waits for `event` */
/* out[j] = sum_i |A[i,j]|: Gershgorin sums (_decomp.py:349-354) of a locally owned block column of a symmetric
 * matrix (column sums = row sums), rows x cols */
int lgp_colabssum_f64(lgp_ctx *ctx, size_t rows, size_t cols, const double *dA, size_t lda, double *dout);

/* building blocks of the distributed factorisation (the single-GPU driver is built from the same ones):
 * factor one block-column panel held locally: dP is rows x nb (ld ldp), its top nb x nb block is the
 * diagonal block; on exit dP holds the L panel and ddblk its packed diagonal blocks (lgp_chol_dblk_size(nb)).
 * info offset `row0` is added to pivot indices. */
int lgp_panel_factor_f64(lgp_ctx *ctx, size_t rows, size_t nb, double *dP, size_t ldp, double *ddblk, size_t row0);
/* The order in which the resident workers of the factorisation execute the diagonal squares of all 1024-column panels
 * of an n x n matrix (n a multiple of 128): two words per task, word 0 = type | j << 8 | c << 16 | k << 24 with k the
 * panel, type (low 7 bits; bit 7 set: the task is on the dependent chain, which has workers of its own) 0: diagonal
 * block j of the panel factored, 1: the 64 rows of unit r of the square solved against block
 * j, 2: the 64 x 64 tile at row unit r, column unit c of the square updated with block j; word 1 = r.  Writes at most
 * `capacity` tasks and returns their total number (negative: error).  Host only, no device needed: the parity tests
 * check that every task comes after the tasks it waits for and that executing the list factors the matrix.  No
 * counterpart in the reference (LAPACK potrf inside jlinalg.cholesky, _linalg/_decomp.py:388, orders its own blocks). */
long lgp_chain_schedule(size_t n, unsigned int *tasks, size_t capacity);
/* The same update for ALL block columns a rank owns in a 1-D block-column-cyclic layout, in one launch: dC is the
 * rank's local storage from its first not yet finished block column on (rows x cols, local block q of width nb =
 * global block (q0 + q) * nranks + rank), dP the panel rows from the global row of C(0,0) on (rows x k).  Element (i, j)
 * is updated with P[i] . P[j + nb (nranks-1) (j / nb)] where that global column is <= i ("staircase" lower). */
int lgp_update_nt_cyclic_f64(lgp_ctx *ctx, size_t rows, size_t cols, size_t k, const double *dP, size_t ldp,
                             double *dC, size_t ldc, size_t nb, size_t nranks);
/* C (rows x cols, ldc) -= A (rows x k) * B (cols x k)^T, `lower` != 0 restricts to row >= col + diag_offset */
int lgp_update_nt_f64(lgp_ctx *ctx, size_t rows, size_t cols, size_t k, const double *dA, size_t lda,
                      const double *dB, size_t ldb, double *dC, size_t ldc, int lower);
int lgp_get_info(lgp_ctx *ctx, int *hinfo);    /* read & reset the pivot info word; synchronises */

/* The whole sharded factorisation in one call (no host language between the panels): dLloc holds this rank's block
 * columns (block b = columns [b nb, (b+1) nb) of the n x n matrix lives on rank b % nranks at local columns
 * [(b / nranks) nb, ...), all n + nx rows, ld ldl; lower triangle + regularisation already in place; rows n .. n+nx-1 are
 * right-hand sides carried through the factorisation as in lgp_chol_factor_rhs).  Right-looking; the panels are factored,
 * broadcast (ncclBroadcast on a stream of its own) and consumed in row PIECES -- the diagonal square, then slabs of rows --
 * that every stream of every rank takes in one common wavefront order (lgp_chol_dist_plan), so that the dependent chain is that
 * of one slab and the slabs below it follow as a pipeline; each rank updates its remaining block columns with a piece in one
 * staircase MFMA launch.  ddblk: nlocal * lgp_chol_dblk_size(nb) workspace/outputs (packed diagonal blocks of the owned block
 * columns); dpanels: workspace of nslots (n + nx) ldp doubles, nslots from lgp_chol_dist_plan, ldp >= nb (pad it: ldp = nb
 * puts every operand row of the trailing update on one HBM channel).  dsrow (n values, may be NULL): the power-of-two row
 * scaling s undone on the finished factor, L = L~ s[:, None] (_decomp.py:392).  *hinfo: pivot info of THIS rank's panels
 * (0 = ok).  Uses the context's communicator (lgp_comm_init / lgp_comm_init_host; none: single rank).  Synchronises.  A
 * failure on one rank still issues the rest of that rank's broadcasts, so that its peers are not left inside a collective.
 * No counterpart in the reference (LAPACK potrf on one host, _linalg/_decomp.py:388). */
int lgp_chol_factor_dist_f64(lgp_ctx *ctx, size_t n, size_t nb, size_t nx, double *dLloc, size_t ldl, double *ddblk,
                             double *dpanels, size_t ldp, size_t nslots, const double *dsrow, int *hinfo);
/* The plan of that factorisation (host only, no device needed): the pieces in the order in which every stream of every rank
 * takes them, four values each -- panel b, piece index i (0 = the diagonal square, which stays on its owner), first row,
 * end row -- at most `capacity` of them written to `pieces` (may be NULL); returns their number (negative: error).  slab:
 * rows per slab below the squares (rounded up to a multiple of nb), 0 = one piece with all the rows below, (size_t)-1 = the
 * library's default for `nranks` (what lgp_chol_factor_dist uses: none for one rank, 8 nb otherwise, env LGP_DIST_SLAB).
 * *nslots: panel slots the factorisation needs; *slab_out: the slab height used.  The CPU tests execute this list over
 * numpy + gloo (tests/test_dist_gloo.py). */
long lgp_chol_dist_plan(size_t n, size_t nb, size_t nx, size_t nranks, size_t slab, unsigned long long *pieces, size_t capacity,
                        size_t *nslots, size_t *slab_out);
/* The SCHEDULE of that factorisation for one rank, without a device: lgp_chol_factor_dist_f64 itself runs with a tracer that
 * writes every enqueue down instead of performing it -- event waits and records, kernel launches, copies and broadcasts, each
 * with its stream and the operand ranges it was given.  20 int64 words per record (kind, stream, 18 arguments: layout in
 * csrc/common.h, decoder and checker in tests/dist_trace.py), at most `capacity` records written to `rec` (may be NULL);
 * returns their number (negative: error).  slab as in lgp_chol_dist_plan; flags: -1 = the library's defaults for `nranks`,
 * else bit 0 = a rank's own pieces are broadcast from the communication stream behind an event (default with several ranks)
 * instead of from the stream that packed them, bit 1 = squares go to the chain kernel where the library would use it, bit 2 =
 * broadcasts are issued with one rank too.  The same records are written by a live call under LGP_DIST_TRACE=<path> (one file
 * <path>.rank<r> per rank); LGP_DIST_JITTER=<seed> puts spin kernels of random length in front of randomly chosen enqueues.
 * The CPU suite checks that every pair of conflicting accesses of the schedule is ordered by stream order and event edges, and
 * that a simulator running it in any order those edges allow computes the factor (tests/test_dist_trace.py).  No reference
 * counterpart. */
long lgp_chol_dist_trace(size_t n, size_t nb, size_t nx, int rank, int nranks, size_t slab, int flags, long long *rec, size_t capacity);
/* The last lgp_chol_factor_dist_f64 of this context measured from inside (for the multi-GPU line of bench.py: is a step bound
 * by compute, by the links or by the chain of slabs?).  enable: 1 / 0 switch the measurement on / off for the calls that follow
 * (pairs of timing events around the compute stream's waits for arriving pieces and around the packs: keep it off in timed
 * runs), -1 leaves it as it is.  out (8 doubles, may be NULL): wall ms of the call; ms the compute stream was idle in front of a
 * piece that had not arrived (or, for a panel of its own, was not solved); ms of packs; panel bytes its updates read from other
 * ranks; panel bytes of its own; pieces; broadcast algorithm (0 none, 1 the transport's broadcast, 2 scatter + all-gather);
 * 1 if own pieces were handed to the communication stream.  No reference counterpart. */
int lgp_dist_stats(lgp_ctx *ctx, int enable, double *out);
/* One horizontal slab of the staircase update above: rows [roff, roff + rows) below the diagonal of the first block column
 * of dC (which points at the slab's first row, as dA does; dB points at the panel row of that diagonal). */
int lgp_update_nt_slab_f64(lgp_ctx *ctx, size_t rows, size_t cols, size_t k, const double *dA, size_t lda, const double *dB, size_t ldb,
                           double *dC, size_t ldc, size_t nb, size_t nranks, size_t roff);
/* dout[0] = max_i dx[i] (NaN if any element is NaN, like jnp.max): the Gershgorin bound over a rank's column sums */
int lgp_max_f64(lgp_ctx *ctx, size_t n, const double *dx, double *dout);
/* `steps` Lanczos iterations on the symmetric n x n device matrix dK from the unit host vector hq0 (n doubles), entirely on the
 * device: halpha, hbeta (steps values each, host) are the diagonal and the off-diagonal of the tridiagonal matrix whose
 * eigenvalues lie inside the spectrum of K; hbeta[j] = 0 marks an invariant subspace (cut the matrix there).  Three-term
 * recurrence, vectors and sums in double for both matrix types.  Replaces the two LOBPCG iterations on a host copy behind
 * GP(checkpos=True) (_GP/_elements.py:651-668).  Synchronises. */
int lgp_lanczos_f64(lgp_ctx *ctx, size_t n, const double *dK, size_t ldk, size_t steps, const double *hq0, double *halpha, double *hbeta);
int lgp_lanczos_f32(lgp_ctx *ctx, size_t n, const float *dK, size_t ldk, size_t steps, const double *hq0, double *halpha, double *hbeta);
/* A_ij <- A_ij / drow_i / dcol_j (divide != 0) or * drow_i * dcol_j: the power-of-two scaling of a locally owned block
 * column and its undoing on the factor (_decomp.py:385,392); drow / dcol may be NULL */
int lgp_scale_rows_cols_f64(lgp_ctx *ctx, size_t rows, size_t cols, double *dA, size_t lda,
                            const double *drow, const double *dcol, int divide);
/* ds_j = 2^rint(log2(A_jj) / 2), or 1 where A_jj == 0 (_decomp.py:356-361) */
int lgp_diag_scale_pow2_f64(lgp_ctx *ctx, size_t n, const double *dA, size_t lda, double *ds);

#ifdef __cplusplus
}
#endif
#endif /* LSQFITGP_HIP_H */

// cpu_backend.cpp -- TEST / MEASUREMENT INFRASTRUCTURE, not part of the product (the product has no CPU fallback).
//
// A C++ / OpenMP restatement of the dense hot path behind the SAME C-ABI as the HIP library (include/lsqfitgp_hip.h: same
// names, same argument lists, row-major matrices with leading dimensions), operating on HOST pointers: SURVEY.md 8(d)(ii) asks
// for it as the CPU baseline that is "always" timed beside the GPU, next to the numpy / scipy oracle (whose LAPACK ran at
// 30 GFLOP/s on 64 threads of the GPU box in round 2: 64 BLAS threads on a 16-CPU share of the host).  Built by
// oracle/Makefile into oracle/_build/liblsqfitgp_cpu.so; loaded only by tests/ and by bench.py's cpu_baseline leg.
//
// What it follows (reference lines): Chol.__init__ _linalg/_decomp.py:380-393 (power-of-two scaling, Gershgorin jitter, blocked
// lower Cholesky = LAPACK potrf in jax), solve_triangular :402-439, the Normal density :484-488, the posterior
// _GP/_compute.py:254-260, the isotropic kernels _Kernel/_isotropic.py:61-81, _kernels/_basic.py:75, _kernels/_matern.py:29-49.
//
// One GEMM core does all the O(n^3) work: C = beta C + alpha op(A) op(B) with strided operands, BLIS-style (operands packed
// into micro-panels, a 16 x 12 register tile of 8-wide fp64 vectors = 24 accumulators, broadcast of B from memory), AVX-512 /
// AVX2 / generic clones dispatched at load time; parallel over the row blocks of C.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <omp.h>

extern "C" {
struct lgp_ctx { int threads; };
}

namespace {

thread_local char g_err[256] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Denormals: the far off-diagonal entries of an ExpQuad Gram matrix (exp(-r^2 / 2), r up to 500 at C2) are subnormal numbers,
// and x86 cores take a microcode assist for every operation on one: the same factorisation ran at 79 instead of 254 GFLOP/s
// (8 threads, N = 8192; LAPACK: 24 instead of 68).  The worker threads therefore flush subnormals to zero (MXCSR FTZ | DAZ) for
// the duration of a call, as the XLA CPU runtime does for its compute threads; values below 2.2e-308 become 0.
#if defined(__x86_64__)
#include <xmmintrin.h>
struct FtzScope {
  unsigned saved;
  FtzScope() : saved(_mm_getcsr()) { _mm_setcsr(saved | 0x8040u); }
  ~FtzScope() { _mm_setcsr(saved); }
};
#else
struct FtzScope {};
#endif

constexpr int MR = 16, NR = 12, KC = 256, MC = 192, NC = 3072;
typedef double v8d __attribute__((vector_size(64), aligned(8)));

// Ap: kc x MR (column k: MR contiguous values), Bp: kc x NR; acc += Ap Bp'
__attribute__((target_clones("avx512f", "avx2,fma", "default")))
void micro_kernel(int kc, const double *__restrict Ap, const double *__restrict Bp, double *__restrict acc /* NR x MR, column j contiguous */) {
  v8d c[NR][2];
  for (int j = 0; j < NR; ++j) c[j][0] = c[j][1] = (v8d){0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = 0; k < kc; ++k) {
    const v8d a0 = *(const v8d *)(Ap + (size_t)k * MR), a1 = *(const v8d *)(Ap + (size_t)k * MR + 8);
    const double *b = Bp + (size_t)k * NR;
#pragma GCC unroll 12
    for (int j = 0; j < NR; ++j) {
      const v8d bj = {b[j], b[j], b[j], b[j], b[j], b[j], b[j], b[j]};
      c[j][0] += a0 * bj;
      c[j][1] += a1 * bj;
    }
  }
  for (int j = 0; j < NR; ++j) {
    *(v8d *)(acc + j * MR) = c[j][0];
    *(v8d *)(acc + j * MR + 8) = c[j][1];
  }
}

// element (i, k) of op(A) at A[i * rs + k * cs]
void pack_a(int mc, int kc, const double *A, long rs, long cs, double *Ap) {
  for (int i0 = 0; i0 < mc; i0 += MR) {
    const int mr = std::min(MR, mc - i0);
    double *dst = Ap + (size_t)i0 * kc;
    for (int k = 0; k < kc; ++k) {
      for (int i = 0; i < mr; ++i) dst[(size_t)k * MR + i] = A[(long)(i0 + i) * rs + (long)k * cs];
      for (int i = mr; i < MR; ++i) dst[(size_t)k * MR + i] = 0.0;
    }
  }
}
// element (k, j) of op(B) at B[k * rs + j * cs]
void pack_b(int kc, int nc, const double *B, long rs, long cs, double *Bp) {
#pragma omp for schedule(static)
  for (int j0 = 0; j0 < nc; j0 += NR) {
    const int nr = std::min(NR, nc - j0);
    double *dst = Bp + (size_t)j0 * kc;
    for (int k = 0; k < kc; ++k) {
      for (int j = 0; j < nr; ++j) dst[(size_t)k * NR + j] = B[(long)k * rs + (long)(j0 + j) * cs];
      for (int j = nr; j < NR; ++j) dst[(size_t)k * NR + j] = 0.0;
    }
  }
}

// C (m x n, row-major ldc) = beta C + alpha op(A) (m x k) op(B) (k x n); lower: only entries with row + roff >= col are required
// (tiles entirely above are skipped, the others are computed in full: callers never read above the diagonal)
void gemm(int threads, int m, int n, int k, double alpha, const double *A, long ars, long acs, const double *B, long brs, long bcs, double beta,
          double *C, long ldc, bool lower = false, long roff = 0) {
  if (m <= 0 || n <= 0) return;
  if (k <= 0 || alpha == 0.0) {
    if (beta != 1.0)
      for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) C[(long)i * ldc + j] = beta == 0.0 ? 0.0 : beta * C[(long)i * ldc + j];
    return;
  }
  std::vector<double> Bp((size_t)KC * (std::min(n, NC) + NR));
  const int nth = std::max(1, std::min(threads, (m + MR - 1) / MR));
#pragma omp parallel num_threads(nth)
  {
    FtzScope ftz;
    std::vector<double> Ap((size_t)MC * KC + 64);
    alignas(64) double acc[MR * NR];
    for (int jc = 0; jc < n; jc += NC) {
      const int nc = std::min(NC, n - jc);
      for (int pc = 0; pc < k; pc += KC) {
        const int kc = std::min(KC, k - pc);
        const double bet = pc == 0 ? beta : 1.0;
        pack_b(kc, nc, B + (long)pc * brs + (long)jc * bcs, brs, bcs, Bp.data());      // (omp for + implicit barrier)
#pragma omp for schedule(dynamic, 1)
        for (int ic = 0; ic < m; ic += MC) {
          const int mc = std::min(MC, m - ic);
          if (lower && ic + mc - 1 + roff < jc) continue;
          pack_a(mc, kc, A + (long)ic * ars + (long)pc * acs, ars, acs, Ap.data());
          for (int jr = 0; jr < nc; jr += NR) {
            const int nr = std::min(NR, nc - jr);
            for (int ir = 0; ir < mc; ir += MR) {
              const int mr = std::min(MR, mc - ir);
              if (lower && ic + ir + mr - 1 + roff < jc + jr) continue;
              micro_kernel(kc, Ap.data() + (size_t)ir * kc, Bp.data() + (size_t)jr * kc, acc);
              double *Ct = C + (long)(ic + ir) * ldc + (jc + jr);
              for (int i = 0; i < mr; ++i)
                for (int j = 0; j < nr; ++j) {
                  const double v = alpha * acc[j * MR + i];
                  Ct[(long)i * ldc + j] = bet == 0.0 ? v : (bet == 1.0 ? Ct[(long)i * ldc + j] + v : bet * Ct[(long)i * ldc + j] + v);
                }
            }
          }
        }
      }
    }
  }
}

// ---- Cholesky -----------------------------------------------------------------------------------------------------------
constexpr int NB = 256, IB = 32;

// unblocked lower Cholesky of an nb x nb block (row-major); returns 0 or the 1-based index of the first bad pivot
int potf2(int nb, double *A, long lda) {
  for (int j = 0; j < nb; ++j) {
    double d = A[(long)j * lda + j];
    for (int k = 0; k < j; ++k) d -= A[(long)j * lda + k] * A[(long)j * lda + k];
    if (!(d > 0.0) || !std::isfinite(d)) return j + 1;
    d = std::sqrt(d);
    A[(long)j * lda + j] = d;
    for (int i = j + 1; i < nb; ++i) {
      double s = A[(long)i * lda + j];
      const double *ai = A + (long)i * lda, *aj = A + (long)j * lda;
      for (int k = 0; k < j; ++k) s -= ai[k] * aj[k];
      A[(long)i * lda + j] = s / d;
    }
  }
  return 0;
}
// blocked (IB) lower Cholesky of an nb x nb diagonal block, single thread (on the dependent chain)
int potrf_block(int nb, double *A, long lda) {
  for (int j0 = 0; j0 < nb; j0 += IB) {
    const int jb = std::min(IB, nb - j0);
    int info = potf2(jb, A + (long)j0 * lda + j0, lda);
    if (info) return j0 + info;
    const int below = nb - j0 - jb;
    if (below <= 0) continue;
    double *X = A + (long)(j0 + jb) * lda + j0, *D = A + (long)j0 * lda + j0;
    for (int i = 0; i < below; ++i)                 // X <- X D^-T
      for (int c = 0; c < jb; ++c) {
        double s = X[(long)i * lda + c];
        for (int k = 0; k < c; ++k) s -= X[(long)i * lda + k] * D[(long)c * lda + k];
        X[(long)i * lda + c] = s / D[(long)c * lda + c];
      }
    gemm(1, below, below, jb, -1.0, X, lda, 1, X, 1, lda, 1.0, A + (long)(j0 + jb) * lda + (j0 + jb), lda, true);
  }
  return 0;
}
// rows x nb panel X <- X D^-T for a factored nb x nb block D: blocked over IB columns, the rest by the GEMM core; parallel over rows
void trsm_right_lt(int threads, long rows, int nb, const double *D, long ldd, double *X, long ldx) {
  const long chunk = 96;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
  for (long r0 = 0; r0 < rows; r0 += chunk) {
    FtzScope ftz;
    const int mr = (int)std::min(chunk, rows - r0);
    double *Xr = X + r0 * ldx;
    for (int c0 = 0; c0 < nb; c0 += IB) {
      const int cb = std::min(IB, nb - c0);
      if (c0 > 0) gemm(1, mr, cb, c0, -1.0, Xr, ldx, 1, D + (long)c0 * ldd, 1, ldd, 1.0, Xr + c0, ldx);
      for (int i = 0; i < mr; ++i)
        for (int c = 0; c < cb; ++c) {
          double s = Xr[(long)i * ldx + c0 + c];
          for (int k = 0; k < c; ++k) s -= Xr[(long)i * ldx + c0 + k] * D[(long)(c0 + c) * ldd + c0 + k];
          Xr[(long)i * ldx + c0 + c] = s / D[(long)(c0 + c) * ldd + c0 + c];
        }
    }
  }
}
// in-place lower Cholesky of the n x n matrix L (lower triangle read), with `extra` rows below carried along; info as LAPACK
int potrf(int threads, long n, double *L, long ldl, long extra) {
  FtzScope ftz;
  for (long k0 = 0; k0 < n; k0 += NB) {
    const int kb = (int)std::min<long>(NB, n - k0);
    double *D = L + k0 * ldl + k0;
    const int info = potrf_block(kb, D, ldl);
    if (info) return (int)(k0 + info);
    const long below = n + extra - (k0 + kb);
    if (below <= 0) continue;
    double *P = L + (k0 + kb) * ldl + k0;
    trsm_right_lt(threads, below, kb, D, ldl, P, ldl);
    const long rem = n - (k0 + kb);
    if (rem > 0) gemm(threads, (int)below, (int)rem, kb, -1.0, P, ldl, 1, P, 1, ldl, 1.0, L + (k0 + kb) * ldl + (k0 + kb), ldl, true);
  }
  return 0;
}
// B (n x nrhs) <- L^-1 B (trans = 0) or L^-T B (trans != 0)
void trsm_left(int threads, long n, const double *L, long ldl, int trans, double *B, long ldb, long nrhs) {
  if (!trans) {
    for (long k0 = 0; k0 < n; k0 += NB) {
      const int kb = (int)std::min<long>(NB, n - k0);
#pragma omp parallel for schedule(static) num_threads(threads)
      for (long c0 = 0; c0 < nrhs; c0 += 64) {
        FtzScope ftz;
        const long cb = std::min<long>(64, nrhs - c0);
        for (int i = 0; i < kb; ++i) {
          double *bi = B + (k0 + i) * ldb + c0;
          const double *li = L + (k0 + i) * ldl + k0;
          for (int k = 0; k < i; ++k) {
            const double l = li[k];
            const double *bk = B + (k0 + k) * ldb + c0;
            for (long c = 0; c < cb; ++c) bi[c] -= l * bk[c];
          }
          const double d = li[i];
          for (long c = 0; c < cb; ++c) bi[c] /= d;
        }
      }
      const long below = n - (k0 + kb);
      if (below > 0) gemm(threads, (int)below, (int)nrhs, kb, -1.0, L + (k0 + kb) * ldl + k0, ldl, 1, B + k0 * ldb, ldb, 1, 1.0, B + (k0 + kb) * ldb, ldb);
    }
  } else {
    const long nblk = (n + NB - 1) / NB;
    for (long blk = nblk; blk-- > 0;) {
      const long k0 = blk * NB;
      const int kb = (int)std::min<long>(NB, n - k0);
#pragma omp parallel for schedule(static) num_threads(threads)
      for (long c0 = 0; c0 < nrhs; c0 += 64) {
        FtzScope ftz;
        const long cb = std::min<long>(64, nrhs - c0);
        for (int i = kb; i-- > 0;) {
          double *bi = B + (k0 + i) * ldb + c0;
          for (int k = i + 1; k < kb; ++k) {
            const double l = L[(k0 + k) * ldl + k0 + i];
            const double *bk = B + (k0 + k) * ldb + c0;
            for (long c = 0; c < cb; ++c) bi[c] -= l * bk[c];
          }
          const double d = L[(k0 + i) * ldl + k0 + i];
          for (long c = 0; c < cb; ++c) bi[c] /= d;
        }
      }
      if (k0 > 0) gemm(threads, (int)k0, (int)nrhs, kb, -1.0, L + k0 * ldl, 1, ldl, B + k0 * ldb, ldb, 1, 1.0, B, ldb);
    }
  }
}

// Chol.__init__ (_decomp.py:380-393): s = 2^rint(log2(diag)/2), K~ = K/s/s', eps = epsrel max_i sum_j |K~_ij| + epsabs, L~, L = s L~
int chol_factor(lgp_ctx *c, size_t n, const double *K, size_t ldk, double epsrel, double epsabs, int flags, double *L, size_t ldl, size_t nrhs,
                double *heps) {
  const int T = c->threads;
  std::vector<double> s(n), rows(n);
  double mins2 = INFINITY;
  for (size_t i = 0; i < n; ++i) {
    const double d = K[i * ldk + i];
    s[i] = d != 0.0 ? std::exp2(std::rint(0.5 * std::log2(d))) : 1.0;      // (rint: round half to even, :356-361)
    mins2 = std::min(mins2, s[i] * s[i]);
  }
#pragma omp parallel for schedule(static) num_threads(T)
  for (size_t i = 0; i < n; ++i) {
    FtzScope ftz;
    double acc = 0.0;
    const double si = s[i];
    for (size_t j = 0; j < n; ++j) {
      const double v = K[i * ldk + j] / s[j] / si;
      acc += std::fabs(v);
      if (j <= i) L[i * ldl + j] = v;
      else L[i * ldl + j] = 0.0;
    }
    rows[i] = acc;
  }
  const double meps = 2.220446049250313e-16;
  if (flags & 1) epsrel = (double)n * meps;       // LGP_EPS_AUTO_REL
  if (flags & 2) epsabs = meps;                   // LGP_EPS_AUTO_ABS
  double mx = 0.0;
  for (size_t i = 0; i < n; ++i) mx = std::max(mx, rows[i]);
  const double eps = epsrel * mx + epsabs;
  for (size_t i = 0; i < n; ++i) L[i * ldl + i] += eps;
  for (size_t r = 0; r < nrhs; ++r)               // carried rows B^T: columns / s
    for (size_t j = 0; j < n; ++j) L[(n + r) * ldl + j] /= s[j];
  const int info = potrf(T, (long)n, L, (long)ldl, (long)nrhs);
  if (info) return info;
  bool unit = true;
  for (size_t i = 0; i < n; ++i) unit = unit && s[i] == 1.0;
  // L = L~ s[:, None] (:392); the strict upper triangle (scratch of the tiles that straddle the diagonal) is zero in the result
#pragma omp parallel for schedule(static) num_threads(T)
  for (size_t i = 0; i < n; ++i) {
    if (!unit)
      for (size_t j = 0; j <= i; ++j) L[i * ldl + j] *= s[i];
    for (size_t j = i + 1; j < n; ++j) L[i * ldl + j] = 0.0;
  }
  if (heps) *heps = eps * mins2;
  return 0;
}

// ---- isotropic Gram kernels (ExpQuad = 0, Maternp = 1; _kernels/_basic.py:75, _matern.py:29-49, _bessel.py:101-110) ------------
#pragma GCC push_options
#pragma GCC optimize("fast-math")       // (the vector exp of libmvec needs it; the reference's XLA loop is vectorised too)
void gram_rows(int threads, int kernel, int p, int d, const double *x, size_t ldx, size_t n, const double *y, size_t ldy, size_t m, int has_loc,
               double loc, int has_scale, double scale, double *out, size_t ldo) {
  std::vector<double> u((size_t)d * n), v((size_t)d * m);
  for (int f = 0; f < d; ++f) {
    for (size_t i = 0; i < n; ++i) {
      double t = x[f * ldx + i];
      if (has_loc) t -= loc;
      if (has_scale) t /= scale;
      u[(size_t)f * n + i] = t;
    }
    for (size_t j = 0; j < m; ++j) {
      double t = y[f * ldy + j];
      if (has_loc) t -= loc;
      if (has_scale) t /= scale;
      v[(size_t)f * m + j] = t;
    }
  }
  double coef[20];
  for (int k = 0; k < p; ++k) coef[k] = (double)(p - k) / ((double)(2 * p - k) * (double)(k + 1)) * 2.0;
#pragma omp parallel for schedule(static) num_threads(threads)
  for (size_t i = 0; i < n; ++i) {
    FtzScope ftz;
    double *o = out + i * ldo;
#pragma omp simd
    for (size_t j = 0; j < m; ++j) {
      double r2 = 0.0;
      for (int f = 0; f < d; ++f) {
        const double t = u[(size_t)f * n + i] - v[(size_t)f * m + j];
        r2 += t * t;
      }
      o[j] = r2;
    }
    if (kernel == 0) {
#pragma omp simd
      for (size_t j = 0; j < m; ++j) o[j] = std::exp(-0.5 * o[j]);
    } else {
#pragma omp simd
      for (size_t j = 0; j < m; ++j) {
        const double t = std::sqrt((double)(2 * p + 1) * o[j] + 1e-30);
        double poly = 1.0;
        for (int k = p - 1; k >= 0; --k) poly = 1.0 + poly * coef[k] * t;
        o[j] = std::exp(-t) * poly;
      }
    }
  }
}
#pragma GCC pop_options

}  // namespace

extern "C" {

int lgp_version(void) { return 1; }
const char *lgp_last_error(void) { return g_err; }
int lgp_device_count(int *count) { if (count) *count = 0; return 0; }
int lgp_ctx_create(int threads, lgp_ctx **out) {
  lgp_ctx *c = new lgp_ctx();
  c->threads = threads > 0 ? threads : omp_get_max_threads();
  *out = c;
  return 0;
}
int lgp_ctx_destroy(lgp_ctx *c) { delete c; return 0; }
size_t lgp_chol_dblk_size(size_t) { return 1; }

int lgp_gram_isotropic_f64(lgp_ctx *ctx, int kernel, int p, int npar, const double *hpar, int d, const double *dx, size_t ldx, size_t n, const double *dy,
                           size_t ldy, size_t m, int has_loc, double loc, int has_scale, double scale, int nops, const int *hops_code,
                           const double *hops_val, int symmetric, double *dout, size_t ldo) {
  (void)npar; (void)hpar; (void)symmetric;
  if (kernel != 0 && kernel != 1) { set_error("cpu backend: kernel %d not implemented", kernel); return -1; }
  if (p > 16) { set_error("Maternp order above 16"); return -1; }
  gram_rows(ctx->threads, kernel, p, d, dx, ldx, n, dy, ldy, m, has_loc, loc, has_scale, scale, dout, ldo);
  for (int q = 0; q < nops; ++q)
    for (size_t i = 0; i < n; ++i)
      for (size_t j = 0; j < m; ++j) dout[i * ldo + j] = hops_code[q] == 0 ? dout[i * ldo + j] * hops_val[q] : dout[i * ldo + j] + hops_val[q];
  return 0;
}
int lgp_add_diag_f64(lgp_ctx *, size_t n, double *A, size_t lda, const double *v, double a) {
  for (size_t i = 0; i < n; ++i) A[i * lda + i] += v ? a * v[i] : a;
  return 0;
}
int lgp_chol_factor_f64(lgp_ctx *c, size_t n, const double *K, size_t ldk, double epsrel, double epsabs, int flags, double *L, size_t ldl, double *,
                        double *heps) {
  return chol_factor(c, n, K, ldk, epsrel, epsabs, flags, L, ldl, 0, heps);
}
int lgp_chol_factor_rhs_f64(lgp_ctx *c, size_t n, const double *K, size_t ldk, double epsrel, double epsabs, int flags, double *L, size_t ldl,
                            size_t nrhs, double *, double *heps) {
  return chol_factor(c, n, K, ldk, epsrel, epsabs, flags, L, ldl, nrhs, heps);
}
int lgp_potrf_f64(lgp_ctx *c, size_t n, double *L, size_t ldl, double *) { return potrf(c->threads, (long)n, L, (long)ldl, 0); }
int lgp_trsm_f64(lgp_ctx *c, size_t n, const double *L, size_t ldl, const double *, int trans, double *B, size_t ldb, size_t nrhs) {
  trsm_left(c->threads, (long)n, L, (long)ldl, trans, B, (long)ldb, (long)nrhs);
  return 0;
}
int lgp_gemm_f64(lgp_ctx *c, int ta, int tb, size_t m, size_t n, size_t k, double alpha, const double *A, size_t lda, const double *B, size_t ldb,
                 double beta, double *C, size_t ldc) {
  gemm(c->threads, (int)m, (int)n, (int)k, alpha, A, ta ? 1 : (long)lda, ta ? (long)lda : 1, B, tb ? 1 : (long)ldb, tb ? (long)ldb : 1, beta, C, (long)ldc);
  return 0;
}
// W (n x (m+1)) = [Kxxs | y] on entry, L^-1 W on exit; mean = X' x_y, cov = K - X'X, *hmlnd = 1/2 (n log 2pi + 2 sum log L_ii + |x_y|^2)
int lgp_posterior_f64(lgp_ctx *c, size_t n, size_t m, const double *L, size_t ldl, const double *, double *W, size_t ldw, const double *K, size_t ldk,
                      double *mean, double *cov, size_t ldc, double *hmlnd) {
  const size_t k = m + 1;
  trsm_left(c->threads, (long)n, L, (long)ldl, 0, W, (long)ldw, (long)k);
  std::vector<double> G(k * k);
  gemm(c->threads, (int)k, (int)k, (int)n, 1.0, W, 1, (long)ldw, W, (long)ldw, 1, 0.0, G.data(), (long)k);
  for (size_t i = 0; i < m; ++i) {
    mean[i] = G[i * k + m];
    for (size_t j = 0; j < m; ++j) cov[i * ldc + j] = K[i * ldk + j] - G[i * k + j];
  }
  double ld = 0.0;
  for (size_t i = 0; i < n; ++i) ld += std::log(L[i * ldl + i]);
  if (hmlnd) *hmlnd = 0.5 * ((double)n * std::log(2 * M_PI) + 2.0 * ld + G[m * k + m]);
  return 0;
}
int lgp_mlnd_value_f64(lgp_ctx *c, size_t n, const double *L, size_t ldl, const double *, double *r, double *h) {
  trsm_left(c->threads, (long)n, L, (long)ldl, 0, r, 1, 1);
  double q = 0.0, ld = 0.0;
  for (size_t i = 0; i < n; ++i) { q += r[i] * r[i]; ld += std::log(L[i * ldl + i]); }
  *h = 0.5 * ((double)n * std::log(2 * M_PI) + 2.0 * ld + q);
  return 0;
}

}  // extern "C"

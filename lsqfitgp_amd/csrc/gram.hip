// gram.hip -- Gram-matrix build for gfx950: the `kernel(x[:,None], y[None,:])` call of
// src/lsqfitgp/_GP/_elements.py:575-577 for the three kernels of the hot path.
//
//   isotropic  : u = (x - loc)/scale per field (_Kernel/_ops.py:292-326), r2 = sum_f (u_f - v_f)^2 in field order
//                (_Kernel/_isotropic.py:61-81, _Kernel/_util.py:74-99), then
//                  ExpQuad  exp(-1/2 * r2)                                     (_kernels/_basic.py:75)
//                  Maternp  x = sqrt((2p+1) r2 + 1e-30); exp(-x) * Horner(x)   (_kernels/_matern.py:48-49, _special/_bessel.py:103-110)
//   BART       : depth <= 2 closed forms of the altinput correlation with `repeat` folding
//                (_kernels/_bart.py:628-757), digamma at integers from a harmonic-number table.
//
// Both are HBM-write bound (8 or 4 bytes per pair out, O(d) bytes per point in): a workgroup owns a
// 32 x 512 output tile, the transformed coordinates of its 32 rows and 512 columns are staged once in LDS
// (so (x-loc)/scale is evaluated once per point, exactly as the reference does before differencing), every
// lane produces 16 contiguous bytes per row so a wave stores 1 KiB (fp64: 128 columns) contiguously.
// This file is compiled with -ffp-contract=off: the reference evaluates square-then-add without fused
// multiply-add and the entries are compared to a few ulp.
#include "common.h"

namespace {

constexpr int GT_ROWS = 32;
constexpr int GT_THREADS = 256;
constexpr int MAX_OPS = 8;
constexpr int MAX_P = 16;

template <typename T>
struct OutOps {
  int n;
  int code[MAX_OPS];
  T val[MAX_OPS];
  __device__ __forceinline__ T apply(T v) const {
    for (int i = 0; i < n; ++i) v = (code[i] == LGP_OP_MUL) ? v * val[i] : v + val[i];
    return v;
  }
};

template <typename T>
struct IsoArgs {
  const T *x;
  const T *y;
  T *out;
  long ldx, ldy, ldo;
  int n, m, d;
  int kernel, p;
  int has_loc, has_scale;
  T loc, scale;
  T coef[MAX_P];   // Horner coefficients c_{k+1}/c_k, k = p-1..0 in evaluation order
  T twop1;         // 2p+1
  T par[3];        // GammaExp: gamma, gamma/2;  Cauchy: alpha/2, beta, -beta/alpha  (host-evaluated like the reference's Python floats)
  int flag;        // GammaExp: gamma == 2;  Cauchy: alpha == 2  (the branches that act on r2 itself)
  OutOps<T> ops;
};

template <typename T> struct VecOf;
template <> struct VecOf<double> { typedef double type __attribute__((ext_vector_type(2))); static constexpr int W = 2; };
template <> struct VecOf<float> { typedef float type __attribute__((ext_vector_type(4))); static constexpr int W = 4; };

template <typename T>
__device__ __forceinline__ T iso_core(const IsoArgs<T> &g, T r2) {
  if (g.kernel == LGP_KERNEL_EXPQUAD) {
    return exp(T(-0.5) * r2);
  } else if (g.kernel == LGP_KERNEL_GAMMAEXP) {          // _kernels/_basic.py:111-113
    return g.flag ? exp(-r2) : exp(-pow(r2, g.par[1]));
  } else if (g.kernel == LGP_KERNEL_CAUCHY) {            // _kernels/_basic.py:340-344
    const T power = g.flag ? r2 : pow(r2, g.par[0]);
    return pow(T(1) + power / g.par[1], g.par[2]);
  } else {
    T r2s = g.twop1 * r2;
    T x = sqrt(r2s + T(1e-30));
    T poly = T(1);
    for (int k = 0; k < g.p; ++k) poly = T(1) + poly * g.coef[k] * T(2) * x;
    return exp(-x) * poly;
  }
}

// D > 0: compile-time field count (coordinates of the lane's columns live in registers); D == 0: runtime d
template <typename T, int D>
__global__ void __launch_bounds__(GT_THREADS) gram_iso_kernel(const IsoArgs<T> g) {
  constexpr int W = VecOf<T>::W;
  constexpr int GT_COLS = GT_THREADS * W;
  typedef typename VecOf<T>::type vec_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *su = reinterpret_cast<T *>(smem_raw);            // [d][GT_ROWS]
  const int d = D > 0 ? D : g.d;
  T *sv = su + d * GT_ROWS;                            // [d][GT_COLS]

  const int ntc = (g.m + GT_COLS - 1) / GT_COLS;
  const int tr = blockIdx.x / ntc, tc = blockIdx.x % ntc;
  const int row0 = tr * GT_ROWS, col0 = tc * GT_COLS;
  const int tid = threadIdx.x;

  for (int idx = tid; idx < d * GT_ROWS; idx += GT_THREADS) {
    const int f = idx / GT_ROWS, r = idx % GT_ROWS;
    T v = T(0);
    if (row0 + r < g.n) {
      v = g.x[(long)f * g.ldx + row0 + r];
      if (g.has_loc) v = v - g.loc;
      if (g.has_scale) v = v / g.scale;
    }
    su[idx] = v;
  }
  for (int idx = tid; idx < d * GT_COLS; idx += GT_THREADS) {
    const int f = idx / GT_COLS, c = idx % GT_COLS;
    T v = T(0);
    if (col0 + c < g.m) {
      v = g.y[(long)f * g.ldy + col0 + c];
      if (g.has_loc) v = v - g.loc;
      if (g.has_scale) v = v / g.scale;
    }
    sv[idx] = v;
  }
  __syncthreads();

  const int c0 = tid * W;
  T vreg[D > 0 ? D : 1][W];
  if (D > 0) {
#pragma unroll
    for (int f = 0; f < D; ++f)
#pragma unroll
      for (int e = 0; e < W; ++e) vreg[f][e] = sv[f * GT_COLS + c0 + e];
  }
  const bool colfull = (col0 + c0 + W <= g.m) && ((g.ldo % W) == 0) && ((reinterpret_cast<uintptr_t>(g.out) % 16) == 0);
  const int rmax = (g.n - row0) < GT_ROWS ? (g.n - row0) : GT_ROWS;
  for (int r = 0; r < rmax; ++r) {
    T res[W];
#pragma unroll
    for (int e = 0; e < W; ++e) {
      T r2 = T(0);
      if (D > 0) {
#pragma unroll
        for (int f = 0; f < D; ++f) {
          const T df = su[f * GT_ROWS + r] - vreg[f][e];
          const T sq = df * df;
          r2 = (f == 0) ? sq : r2 + sq;
        }
      } else {
        for (int f = 0; f < d; ++f) {
          const T df = su[f * GT_ROWS + r] - sv[f * GT_COLS + c0 + e];
          const T sq = df * df;
          r2 = (f == 0) ? sq : r2 + sq;
        }
      }
      res[e] = g.ops.apply(iso_core<T>(g, r2));
    }
    T *po = g.out + (long)(row0 + r) * g.ldo + col0 + c0;
    if (colfull) {
      vec_t v;
#pragma unroll
      for (int e = 0; e < W; ++e) v[e] = res[e];
      *reinterpret_cast<vec_t *>(po) = v;
    } else {
#pragma unroll
      for (int e = 0; e < W; ++e)
        if (col0 + c0 + e < g.m) po[e] = res[e];
    }
  }
}

// out[i] = ops(core(r2(x_i, y_i))): the elementwise (broadcast) form of CrossKernel.__call__
// (_Kernel/_crosskernel.py:192-200) for equally shaped x and y, also used by `halfmatrix`
template <typename T>
__global__ void __launch_bounds__(256) gram_iso_pairs_kernel(const IsoArgs<T> g) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= g.n) return;
  T r2 = T(0);
  for (int f = 0; f < g.d; ++f) {
    T u = g.x[(long)f * g.ldx + i], v = g.y[(long)f * g.ldy + i];
    if (g.has_loc) { u = u - g.loc; v = v - g.loc; }
    if (g.has_scale) { u = u / g.scale; v = v / g.scale; }
    const T df = u - v;
    const T sq = df * df;
    r2 = (f == 0) ? sq : r2 + sq;
  }
  g.out[i] = g.ops.apply(iso_core<T>(g, r2));
}

template <typename T>
int gram_iso_launch(lgp_ctx *ctx, int kernel, int p, int npar, const T *hpar, int d, const T *dx, size_t ldx, size_t n, const T *dy, size_t ldy, size_t m,
                    int has_loc, T loc, int has_scale, T scale, int nops, const int *hops_code, const T *hops_val, T *dout,
                    size_t ldo, bool pairs = false) {
  LGP_REQUIRE(ctx != nullptr, "null context");
  if (n == 0 || m == 0) return LGP_OK;
  if (pairs) { LGP_REQUIRE(n == m, "pairs mode needs as many x as y points"); ldo = m; }
  LGP_REQUIRE(dx != nullptr && dy != nullptr && dout != nullptr, "null pointer");
  LGP_REQUIRE(kernel == LGP_KERNEL_EXPQUAD || kernel == LGP_KERNEL_MATERNP || kernel == LGP_KERNEL_GAMMAEXP || kernel == LGP_KERNEL_CAUCHY,
              "unknown isotropic kernel");
  LGP_REQUIRE(d >= 1 && d <= 64, "field count must be in [1,64]");
  LGP_REQUIRE(p >= 0 && p <= MAX_P, "Maternp order must be in [0,16]");
  LGP_REQUIRE(nops >= 0 && nops <= MAX_OPS, "at most 8 output operations");
  LGP_REQUIRE(n < (1u << 30) && m < (1u << 30), "too many points");
  LGP_REQUIRE(ldx >= n && ldy >= m && ldo >= m, "leading dimension too small");
  LGP_REQUIRE(!has_scale || (scale > T(0) && scale < T(INFINITY)), "scale must be positive and finite");   // _ops.py:324-325
  LGP_REQUIRE(!has_loc || (loc > T(-INFINITY) && loc < T(INFINITY)), "loc must be finite");                 // _ops.py:306-307
  DeviceGuard dg(ctx->device);
  IsoArgs<T> g;
  g.x = dx; g.y = dy; g.out = dout;
  g.ldx = (long)ldx; g.ldy = (long)ldy; g.ldo = (long)ldo;
  g.n = (int)n; g.m = (int)m; g.d = d;
  g.kernel = kernel; g.p = p;
  g.has_loc = has_loc; g.has_scale = has_scale; g.loc = loc; g.scale = scale;
  // _bessel.py:106-109: for k in reversed(range(p)): c = (p - k) / ((2p - k)(k + 1)); poly = 1 + poly*c*2*x
  for (int i = 0; i < MAX_P; ++i) g.coef[i] = T(0);
  int pos = 0;
  for (int k = p - 1; k >= 0; --k) g.coef[pos++] = (T)((double)(p - k) / ((double)(2 * p - k) * (double)(k + 1)));
  g.twop1 = (T)(2 * p + 1);
  g.par[0] = g.par[1] = g.par[2] = T(0);
  g.flag = 0;
  if (kernel == LGP_KERNEL_GAMMAEXP) {            // hpar = {gamma, gamma / 2}
    LGP_REQUIRE(npar == 2 && hpar != nullptr, "GammaExp takes {gamma, gamma/2}");
    LGP_REQUIRE(hpar[0] > T(0) && hpar[0] <= T(2), "gamma must be in (0, 2]");                 // _basic.py:109-110
    g.par[0] = hpar[0]; g.par[1] = hpar[1];
    g.flag = hpar[0] == T(2);
  } else if (kernel == LGP_KERNEL_CAUCHY) {       // hpar = {alpha, beta, -beta / alpha}
    LGP_REQUIRE(npar == 3 && hpar != nullptr, "Cauchy takes {alpha, beta, -beta/alpha}");
    LGP_REQUIRE(hpar[0] > T(0) && hpar[0] <= T(2) && hpar[1] > T(0), "alpha must be in (0, 2], beta positive");   // _basic.py:336-338
    g.par[0] = hpar[0] / T(2); g.par[1] = hpar[1]; g.par[2] = hpar[2];
    g.flag = hpar[0] == T(2);
  } else {
    LGP_REQUIRE(npar == 0, "this kernel takes no real parameters");
  }
  g.ops.n = nops;
  for (int i = 0; i < MAX_OPS; ++i) { g.ops.code[i] = 0; g.ops.val[i] = T(0); }
  for (int i = 0; i < nops; ++i) {
    LGP_REQUIRE(hops_code[i] == LGP_OP_MUL || hops_code[i] == LGP_OP_ADD, "unknown output operation");
    g.ops.code[i] = hops_code[i];
    g.ops.val[i] = hops_val[i];
  }
  if (pairs) {
    hipLaunchKernelGGL((gram_iso_pairs_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, g);
    LGP_LAUNCH_CHECK();
    return LGP_OK;
  }
  constexpr int W = VecOf<T>::W;
  constexpr int GT_COLS = GT_THREADS * W;
  const size_t ntr = (n + GT_ROWS - 1) / GT_ROWS, ntc = (m + GT_COLS - 1) / GT_COLS;
  LGP_REQUIRE(ntr * ntc < (1ull << 31), "gram grid too large");
  const size_t shmem = (size_t)d * (GT_ROWS + GT_COLS) * sizeof(T);
  dim3 gd((unsigned)(ntr * ntc)), bd(GT_THREADS);
  switch (d) {
    case 1: hipLaunchKernelGGL((gram_iso_kernel<T, 1>), gd, bd, shmem, ctx->stream, g); break;
    case 2: hipLaunchKernelGGL((gram_iso_kernel<T, 2>), gd, bd, shmem, ctx->stream, g); break;
    case 3: hipLaunchKernelGGL((gram_iso_kernel<T, 3>), gd, bd, shmem, ctx->stream, g); break;
    default: hipLaunchKernelGGL((gram_iso_kernel<T, 0>), gd, bd, shmem, ctx->stream, g); break;
  }
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

// ------------------------------------------------------------------------------------------------
// BART

constexpr int BART_MAX_P = 64;
constexpr int BART_MAX_ROWS = 24;   // repeat*size <= 24 probabilities
constexpr int BT_ROWS = 16, BT_COLS = 256;

template <typename T>
struct BartArgs {
  const int32_t *ix;
  const int32_t *iy;
  T *out;
  const T *gamma_mat;   // optional per-pair gamma (n x m, ld ldo), may alias out
  const T *psi;         // psi[k] = digamma(k) for k = 1..kmax (psi[0] unused)
  long ldo;
  int n, m, p;
  int size, repeat;
  T gamma;
  int nsplits[BART_MAX_P];
  T w[BART_MAX_P];
  T probs[BART_MAX_ROWS];
  OutOps<T> ops;
};

// correlation of one pair; X(a), Y(a), N(a) give the (already weight-masked) indices and split totals of axis a
template <typename T, typename FX, typename FY, typename FN>
__device__ __forceinline__ T bart_pair(const BartArgs<T> &g, FX X, FY Y, FN N, T gam) {
  const int p = g.p;
  const T one = T(1);
  if (p == 0) return one;                                 // no covariates (:651-652)
  bool anyn0 = false;
  for (int a = 0; a < p; ++a) anyn0 |= (X(a) != Y(a));    // reference: 64-bit hash inequality (:674-681)
  if (g.size == 1) {                                      // :685-691
    for (int k = 0; k < g.repeat; ++k) gam = anyn0 ? one - (one - gam) * g.probs[k] : one;
    return gam;
  }
  T Wn = T(0);                                            // :694
  for (int a = 0; a < p; ++a)
    if (N(a)) Wn += g.w[a];
  if (g.size == 2) {                                      // :697-708
    T sum_term = T(0);
    for (int a = 0; a < p; ++a) {
      const int na = N(a), x = X(a), y = Y(a);
      const int n0 = x > y ? x - y : y - x;
      const T wn = na ? g.w[a] / T(na) : T(0);
      sum_term += wn * T(n0);
    }
    for (int k = 0; k < g.repeat; ++k) {
      const T P0 = g.probs[2 * k], P1 = g.probs[2 * k + 1];
      const T Q = one - P1 + gam * P1;
      const T result = one - P0 + Q * (P0 - P0 / Wn * sum_term);
      gam = anyn0 ? result : one;
    }
    return gam;
  }
  // size == 3, :711-757
  const T inv_Wn = one / Wn;
  T S = T(0);
  for (int a = 0; a < p; ++a) {
    const int na = N(a), x = X(a), y = Y(a);
    const int n0 = x > y ? x - y : y - x;
    const T wn = na ? g.w[a] / T(na) : T(0);
    S += wn * T(na - n0);
  }
  T sumi = T(0);
  for (int a = 0; a < p; ++a) {
    const T w = g.w[a];
    const int na = N(a), x = X(a), y = Y(a);
    const bool xlty = x < y;
    const int mn = xlty ? x : y, mx = xlty ? y : x;
    const int n0 = mx - mn;
    const int nminus0 = mx, nplus0 = na - mn, nout = na - n0;
    const T inv_Wnmod = one / (Wn - (na ? w : T(0)));
    const T inv_Wnminus = nplus0 ? inv_Wn : inv_Wnmod;
    const T inv_Wnplus = nminus0 ? inv_Wn : inv_Wnmod;
    const T wn = na ? w / T(na) : T(0);
    const T t = wn * T(n0);
    const T terms1 = (S + t) * (inv_Wnminus + inv_Wnplus + inv_Wn * T(nout - 2));
    T terms2 = nplus0 ? w * inv_Wn * T(n0) / T(nplus0) : w * inv_Wnmod;
    terms2 += nminus0 ? w * inv_Wn * T(n0) / T(nminus0) : w * inv_Wnmod;
    const T psin = g.psi[na ? na : 1];
    const T psiminus = g.psi[1 + mx];          // where(xlty, psi(1+iy), psi(1+ix)) = psi(1 + max)
    const T psiplus = g.psi[1 + na - mn];      // where(xlty, psi(1+n-ix), psi(1+n-iy)) = psi(1 + n - min)
    const T terms3 = w * inv_Wn * T(n0) * (T(2) * psin - psiminus - psiplus);
    sumi += wn * (terms1 - terms2 - terms3);
  }
  for (int k = 0; k < g.repeat; ++k) {
    const T P0 = g.probs[3 * k], P1 = g.probs[3 * k + 1], P2 = g.probs[3 * k + 2];
    const T Q = one + P2 * (gam - one);
    const T sump = S + P1 * (Q * sumi - S);
    const T result = one + P0 * (inv_Wn * sump - one);
    gam = anyn0 ? result : one;
  }
  return gam;
}

template <typename T>
__global__ void __launch_bounds__(BT_COLS) gram_bart_kernel(const BartArgs<T> g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int32_t *sx = reinterpret_cast<int32_t *>(smem_raw);      // [BT_ROWS][p]
  int32_t *sy = sx + BT_ROWS * g.p;                          // [p][BT_COLS]  (transposed: conflict-free)
  const int p = g.p;
  const int ntc = (g.m + BT_COLS - 1) / BT_COLS;
  const int tr = blockIdx.x / ntc, tc = blockIdx.x % ntc;
  const int row0 = tr * BT_ROWS, col0 = tc * BT_COLS;
  const int tid = threadIdx.x;
  // zero-weight axes are ignored: n, ix, iy -> 0 (_bart.py:670-672)
  for (int idx = tid; idx < BT_ROWS * p; idx += BT_COLS) {
    const int r = idx / p, a = idx % p;
    int v = 0;
    if (row0 + r < g.n && g.w[a] != T(0)) v = g.ix[(long)(row0 + r) * p + a];
    sx[idx] = v;
  }
  for (int idx = tid; idx < BT_COLS * p; idx += BT_COLS) {
    const int c = idx / p, a = idx % p;
    int v = 0;
    if (col0 + c < g.m && g.w[a] != T(0)) v = g.iy[(long)(col0 + c) * p + a];
    sy[a * BT_COLS + c] = v;
  }
  __syncthreads();
  const int col = col0 + tid;
  if (col >= g.m) return;
  const int rmax = (g.n - row0) < BT_ROWS ? (g.n - row0) : BT_ROWS;
  for (int r = 0; r < rmax; ++r) {
    const T gam0 = g.gamma_mat ? g.gamma_mat[(long)(row0 + r) * g.ldo + col] : g.gamma;
    const T gam = bart_pair<T>(
        g, [&](int a) { return sx[r * p + a]; }, [&](int a) { return sy[a * BT_COLS + tid]; },
        [&](int a) { return (g.w[a] != T(0)) ? g.nsplits[a] : 0; }, gam0);
    g.out[(long)(row0 + r) * g.ldo + col] = g.ops.apply(gam);
  }
}

// elementwise form: pair i uses ix[i], iy[i] and, if npair != NULL, its own split totals npair[i]
// (the fully vectorised BART.correlation(..., altinput=True) of _bart.py:301-455)
template <typename T>
__global__ void __launch_bounds__(256) bart_pairs_kernel(const BartArgs<T> g, const int32_t *npair) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= g.n) return;
  const int p = g.p;
  const int32_t *x = g.ix + i * p, *y = g.iy + i * p, *nn = npair ? npair + i * p : nullptr;
  const T gam0 = g.gamma_mat ? g.gamma_mat[i] : g.gamma;
  const T gam = bart_pair<T>(
      g, [&](int a) { return (g.w[a] != T(0)) ? x[a] : 0; }, [&](int a) { return (g.w[a] != T(0)) ? y[a] : 0; },
      [&](int a) { return (g.w[a] != T(0)) ? (nn ? nn[a] : g.nsplits[a]) : 0; }, gam0);
  g.out[i] = g.ops.apply(gam);
}

template <typename T>
int gram_bart_launch(lgp_ctx *ctx, int p, const int32_t *dix, size_t n, const int32_t *diy, size_t m, const int32_t *hnsplits,
                     const T *hweights, int size, int repeat, const T *hprobs, T gamma, const T *dgamma_mat, const T *dpsi,
                     size_t npsi, int nops, const int *hops_code, const T *hops_val, T *dout, size_t ldo, bool pairs = false,
                     const int32_t *dnpair = nullptr) {
  LGP_REQUIRE(ctx != nullptr, "null context");
  if (n == 0 || m == 0) return LGP_OK;
  if (pairs) { LGP_REQUIRE(n == m, "pairs mode needs as many x as y points"); ldo = m; }
  LGP_REQUIRE(p >= 0 && p <= BART_MAX_P, "number of covariates must be in [0,64]");
  LGP_REQUIRE(size >= 1 && size <= 3, "bracket size must be 1, 2 or 3 (depth <= 2 per recursion slice)");
  LGP_REQUIRE(repeat >= 1 && size * repeat <= BART_MAX_ROWS, "too many probabilities");
  LGP_REQUIRE(dout != nullptr && (p == 0 || (dix != nullptr && diy != nullptr)), "null pointer");
  LGP_REQUIRE(nops >= 0 && nops <= MAX_OPS, "at most 8 output operations");
  LGP_REQUIRE(n < (1u << 30) && m < (1u << 30) && ldo >= m, "bad dimensions");
  DeviceGuard dg(ctx->device);
  BartArgs<T> g;
  memset(&g, 0, sizeof(g));
  g.ix = dix; g.iy = diy; g.out = dout; g.gamma_mat = dgamma_mat; g.psi = dpsi;
  g.ldo = (long)ldo; g.n = (int)n; g.m = (int)m; g.p = p; g.size = size; g.repeat = repeat; g.gamma = gamma;
  int maxn = 0;
  for (int a = 0; a < p; ++a) {
    LGP_REQUIRE(hweights[a] >= T(0), "weights must be nonnegative");
    g.w[a] = hweights[a];
    if (hnsplits) {
      LGP_REQUIRE(hnsplits[a] >= 0, "negative split count");
      g.nsplits[a] = hnsplits[a];
      if (hnsplits[a] > maxn) maxn = hnsplits[a];
    }
  }
  LGP_REQUIRE(hnsplits != nullptr || (pairs && dnpair != nullptr) || p == 0, "split totals missing");
  // with per-pair totals the caller guarantees npsi >= max total + 2
  if (size == 3) LGP_REQUIRE(dpsi != nullptr && npsi >= (size_t)maxn + 2, "digamma table too short");
  for (int i = 0; i < size * repeat; ++i) g.probs[i] = hprobs[i];
  g.ops.n = nops;
  for (int i = 0; i < nops; ++i) {
    LGP_REQUIRE(hops_code[i] == LGP_OP_MUL || hops_code[i] == LGP_OP_ADD, "unknown output operation");
    g.ops.code[i] = hops_code[i];
    g.ops.val[i] = hops_val[i];
  }
  if (pairs) {
    hipLaunchKernelGGL((bart_pairs_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, g, dnpair);
    LGP_LAUNCH_CHECK();
    return LGP_OK;
  }
  const size_t ntr = (n + BT_ROWS - 1) / BT_ROWS, ntc = (m + BT_COLS - 1) / BT_COLS;
  LGP_REQUIRE(ntr * ntc < (1ull << 31), "gram grid too large");
  const size_t shmem = (size_t)(p > 0 ? p : 1) * (BT_ROWS + BT_COLS) * sizeof(int32_t);
  hipLaunchKernelGGL((gram_bart_kernel<T>), dim3((unsigned)(ntr * ntc)), dim3(BT_COLS), shmem, ctx->stream, g);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

}  // namespace

extern "C" int lgp_gram_isotropic_f64(lgp_ctx *ctx, int kernel, int p, int npar, const double *hpar, int d, const double *dx, size_t ldx, size_t n, const double *dy,
                                      size_t ldy, size_t m, int has_loc, double loc, int has_scale, double scale, int nops,
                                      const int *hops_code, const double *hops_val, int symmetric, double *dout, size_t ldo) {
  (void)symmetric;   // (u-v)^2 == (v-u)^2 bit for bit: the full evaluation is already exactly symmetric
  return gram_iso_launch<double>(ctx, kernel, p, npar, hpar, d, dx, ldx, n, dy, ldy, m, has_loc, loc, has_scale, scale, nops, hops_code, hops_val,
                                 dout, ldo);
}
extern "C" int lgp_gram_isotropic_f32(lgp_ctx *ctx, int kernel, int p, int npar, const float *hpar, int d, const float *dx, size_t ldx, size_t n, const float *dy,
                                      size_t ldy, size_t m, int has_loc, float loc, int has_scale, float scale, int nops,
                                      const int *hops_code, const float *hops_val, int symmetric, float *dout, size_t ldo) {
  (void)symmetric;
  return gram_iso_launch<float>(ctx, kernel, p, npar, hpar, d, dx, ldx, n, dy, ldy, m, has_loc, loc, has_scale, scale, nops, hops_code, hops_val,
                                dout, ldo);
}
extern "C" int lgp_gram_bart_f32(lgp_ctx *ctx, int p, const int32_t *dix, size_t n, const int32_t *diy, size_t m,
                                 const int32_t *hnsplits, const float *hweights, int size, int repeat, const float *hprobs, float gamma,
                                 const float *dgamma_mat, const float *dpsi, size_t npsi, int nops, const int *hops_code,
                                 const float *hops_val, float *dout, size_t ldo) {
  return gram_bart_launch<float>(ctx, p, dix, n, diy, m, hnsplits, hweights, size, repeat, hprobs, gamma, dgamma_mat, dpsi, npsi, nops,
                                 hops_code, hops_val, dout, ldo);
}
extern "C" int lgp_gram_bart_f64(lgp_ctx *ctx, int p, const int32_t *dix, size_t n, const int32_t *diy, size_t m,
                                 const int32_t *hnsplits, const double *hweights, int size, int repeat, const double *hprobs,
                                 double gamma, const double *dgamma_mat, const double *dpsi, size_t npsi, int nops,
                                 const int *hops_code, const double *hops_val, double *dout, size_t ldo) {
  return gram_bart_launch<double>(ctx, p, dix, n, diy, m, hnsplits, hweights, size, repeat, hprobs, gamma, dgamma_mat, dpsi, npsi, nops,
                                  hops_code, hops_val, dout, ldo);
}

// ---- elementwise ("pairs") forms: out[i] = k(x_i, y_i) ------------------------------------------------
extern "C" int lgp_pairs_isotropic_f64(lgp_ctx *ctx, int kernel, int p, int npar, const double *hpar, int d, const double *dx, size_t ldx, const double *dy, size_t ldy,
                                       size_t npairs, int has_loc, double loc, int has_scale, double scale, int nops,
                                       const int *hops_code, const double *hops_val, double *dout) {
  return gram_iso_launch<double>(ctx, kernel, p, npar, hpar, d, dx, ldx, npairs, dy, ldy, npairs, has_loc, loc, has_scale, scale, nops, hops_code,
                                 hops_val, dout, npairs, true);
}
extern "C" int lgp_pairs_isotropic_f32(lgp_ctx *ctx, int kernel, int p, int npar, const float *hpar, int d, const float *dx, size_t ldx, const float *dy, size_t ldy,
                                       size_t npairs, int has_loc, float loc, int has_scale, float scale, int nops,
                                       const int *hops_code, const float *hops_val, float *dout) {
  return gram_iso_launch<float>(ctx, kernel, p, npar, hpar, d, dx, ldx, npairs, dy, ldy, npairs, has_loc, loc, has_scale, scale, nops, hops_code,
                                hops_val, dout, npairs, true);
}
extern "C" int lgp_pairs_bart_f32(lgp_ctx *ctx, int p, const int32_t *dix, const int32_t *diy, size_t npairs, const int32_t *hnsplits,
                                  const int32_t *dnpair, const float *hweights, int size, int repeat, const float *hprobs, float gamma,
                                  const float *dgamma_vec, const float *dpsi, size_t npsi, int nops, const int *hops_code,
                                  const float *hops_val, float *dout) {
  return gram_bart_launch<float>(ctx, p, dix, npairs, diy, npairs, hnsplits, hweights, size, repeat, hprobs, gamma, dgamma_vec, dpsi, npsi,
                                 nops, hops_code, hops_val, dout, npairs, true, dnpair);
}
extern "C" int lgp_pairs_bart_f64(lgp_ctx *ctx, int p, const int32_t *dix, const int32_t *diy, size_t npairs, const int32_t *hnsplits,
                                  const int32_t *dnpair, const double *hweights, int size, int repeat, const double *hprobs, double gamma,
                                  const double *dgamma_vec, const double *dpsi, size_t npsi, int nops, const int *hops_code,
                                  const double *hops_val, double *dout) {
  return gram_bart_launch<double>(ctx, p, dix, npairs, diy, npairs, hnsplits, hweights, size, repeat, hprobs, gamma, dgamma_vec, dpsi,
                                  npsi, nops, hops_code, hops_val, dout, npairs, true, dnpair);
}

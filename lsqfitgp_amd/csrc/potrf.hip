// potrf.hip -- regularised blocked Cholesky for gfx950: the decomposition step of the hot path,
// src/lsqfitgp/_linalg/_decomp.py:380-393 (Chol.__init__), and the blocked triangular solves built on it
// (jlinalg.solve_triangular call sites, _decomp.py:402-439,467).
//
// Structure (row-major, lower):
//   prep      : s = diag_scale_pow2(K) (:356-361); Gershgorin bound of K/s/s' (:349-354); eps (:245-255);
//               L <- tril(K/s/s') + eps*I with the strict upper triangle zeroed      [HBM-bound passes]
//   potrf     : right-looking, outer panel LGP_NB = 512, inner diagonal blocks LGP_DB = 128:
//                 potf2       : one workgroup factors a 128x128 diagonal block held in REGISTERS
//                               (8x8 cyclic tile per thread);
//                 panel trsm  : X = B * L_jj^-T by substitution, one thread per row, L_jj through scalar loads;
//                 panel update: gemm NT, K = 128;
//                 trailing    : syrk (gemm NT, lower tiles only), K = 512 -- the N^3/3 flops, MFMA-bound;
//               the next panel is factored on a second stream while the rest of the trailing update
//               runs (look-ahead).
//   finish    : L <- L * s[:,None], diagonal-block workspace re-extracted, eps_out = eps * min(s^2).
// A non-positive or non-finite pivot is recorded in ctx->d_info (first one wins) and the factor goes
// NaN from there on, like LAPACK-in-jax; the host turns info != 0 into the reference's LinAlgError.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Diagonal-block workspace ("dblk"): for every 128x128 diagonal block of L, LGP_DBLK elements:
//   [0, 128*128)          D   = the block, row-major, zeros above the diagonal, identity padding past n
//   [128*128, 2*128*128)  Dt  = its transpose, row-major
//   [2*128*128, +128)     rd  = 1 / diag
// The triangular solves read D / Dt rows with wave-uniform (scalar) loads.

// potf2 of one diagonal block (nb <= 128) by one 256-thread workgroup, matrix held in REGISTERS.
// Thread (ti, tj), tid = tj*16 + ti, owns A[ti+16a][tj+16b], a,b in 0..7 (cyclic, so the shrinking
// trailing matrix stays balanced).  Step j: the 16 owners of column j (one 16-lane row of one wave)
// read the pivot with v_readlane, scale their column by an accurate reciprocal square root (fp64
// latency on gfx950 is ~32 cycles per dependent op, so the 128-pivot chain is what bounds this kernel:
// rsqrt + multiply is ~9 dependent ops, sqrt + divide ~20) and publish l_i in LDS; ONE barrier;
// everybody applies the rank-1 update A_ic -= l_i l_c (i, c > j) branch-free.  The 16-column group
// index JA is a template parameter so that every register index is static.  The LDS column is
// double-buffered on j&1 so that the next step's publish does not wait for stragglers.
template <typename T>
__device__ __forceinline__ T rsqrt_accurate(T d) {
  if (sizeof(T) == 8) {
    double y = __builtin_amdgcn_rsq((double)d);
    // two Newton steps: y <- y * (1.5 - 0.5 d y^2)
    double h = 0.5 * (double)d;
    y = y * (1.5 - h * y * y);
    y = y * (1.5 - h * y * y);
    return (T)y;
  } else {
    float y = __builtin_amdgcn_rsqf((float)d);
    float h = 0.5f * (float)d;
    y = y * (1.5f - h * y * y);
    return (T)y;
  }
}

template <typename T>
__device__ __forceinline__ T readlane_t(T v, int lane) {
  if (sizeof(T) == 8) {
    long long bits = __builtin_bit_cast(long long, (double)v);
    int lo = __builtin_amdgcn_readlane((int)bits, lane);
    int hi = __builtin_amdgcn_readlane((int)(bits >> 32), lane);
    long long r = ((long long)hi << 32) | (unsigned int)lo;
    return (T)__builtin_bit_cast(double, r);
  } else {
    int bits = __builtin_bit_cast(int, (float)v);
    return (T)__builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, lane));
  }
}

template <typename T, int JA>
__device__ __forceinline__ void potf2_group(T (&a)[8][8], T (*scol)[LGP_DB], int ti, int tj, int nb, int *info, long row0) {
#pragma unroll 1
  for (int jm = 0; jm < 16; ++jm) {
    const int j = JA * 16 + jm, buf = jm & 1;
    if (tj == jm) {   // column owners: lanes (jm&3)*16 .. +15 of wave jm>>2; the diagonal owner has ti == jm
      const T d = readlane_t<T>(a[JA][JA], (jm & 3) * 16 + jm);
      const bool bad = !(d > T(0)) || (d + d == d);   // <= 0, NaN, or +inf
      if (bad && ti == jm && j < nb) atomicCAS(info, 0, (int)(row0 + j + 1));
      const T rinv = bad ? T(NAN) : rsqrt_accurate<T>(d);
#pragma unroll
      for (int p = JA; p < 8; ++p) {
        const int i = ti + 16 * p;
        T l = a[p][JA] * rinv;
        if (i == j) l = d * rinv;
        if (i >= j) {
          scol[buf][i] = l;
          a[p][JA] = l;
        }
      }
    }
    __syncthreads();
    T lc[8], lr[8];
#pragma unroll
    for (int q = JA; q < 8; ++q) {
      const int c = tj + 16 * q, i = ti + 16 * q;
      lc[q] = (c > j) ? scol[buf][c] : T(0);
      lr[q] = (i > j) ? scol[buf][i] : T(0);
    }
#pragma unroll
    for (int p = JA; p < 8; ++p)
#pragma unroll
      for (int q = JA; q < 8; ++q) a[p][q] -= lr[p] * lc[q];   // zero factors where i <= j or c <= j
  }
}

template <typename T>
__global__ void __launch_bounds__(256, 1) potf2_kernel(T *A, long lda, int nb, T *dblk, int *info, long row0) {
  // LDS: only the 2 KB double-buffered column, so that this single workgroup can always be placed next
  // to the 73 KB workgroups of the trailing update running on the other stream (look-ahead).
  constexpr int D = LGP_DB;        // 128
  __shared__ T scol[2][D];

  const int tid = threadIdx.x;
  const int ti = tid & 15, tj = tid >> 4;

  // lower triangle straight into registers (identity padding beyond nb); the block was just written by
  // the previous update, so these 64 strided loads per thread come out of L2
  T a[8][8];
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = ti + 16 * p, c = tj + 16 * q;
      T v = (i == c) ? T(1) : T(0);
      if (i < nb && c <= i) v = A[(long)i * lda + c];
      a[p][q] = v;
    }

  potf2_group<T, 0>(a, scol, ti, tj, nb, info, row0);
  potf2_group<T, 1>(a, scol, ti, tj, nb, info, row0);
  potf2_group<T, 2>(a, scol, ti, tj, nb, info, row0);
  potf2_group<T, 3>(a, scol, ti, tj, nb, info, row0);
  potf2_group<T, 4>(a, scol, ti, tj, nb, info, row0);
  potf2_group<T, 5>(a, scol, ti, tj, nb, info, row0);
  potf2_group<T, 6>(a, scol, ti, tj, nb, info, row0);
  potf2_group<T, 7>(a, scol, ti, tj, nb, info, row0);

  // write-back: L block (lower part only) and the workspace copies D, D^T, 1/diag
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = ti + 16 * p, c = tj + 16 * q;
      const T v = (c <= i) ? a[p][q] : T(0);
      if (i < nb && c <= i) A[(long)i * lda + c] = v;
      dblk[i * D + c] = v;
      dblk[D * D + c * D + i] = v;
      if (i == c) dblk[2 * D * D + i] = T(1) / v;
    }
}

// Re-extract the workspace copies of every diagonal block from the finished factor (after row scaling)
template <typename T>
__global__ void __launch_bounds__(256) extract_dblk_kernel(const T *L, long ldl, int n, T *dblk) {
  constexpr int D = LGP_DB;
  const int b = blockIdx.x;
  T *w = dblk + (long)b * LGP_DBLK;
  const long j0 = (long)b * D;
  for (int idx = threadIdx.x; idx < D * D; idx += 256) {
    const int i = idx / D, c = idx % D;
    // lower element (i, c), c <= i
    T v = (i == c) ? T(1) : T(0);
    if (c <= i && j0 + i < n) v = L[(j0 + i) * ldl + j0 + c];
    if (c > i) v = T(0);
    w[idx] = v;
    // transposed image: Dt[i][c] = D[c][i], nonzero for i <= c
    T u = (i == c) ? T(1) : T(0);
    if (i <= c && j0 + c < n) u = L[(j0 + c) * ldl + j0 + i];
    if (i > c) u = T(0);
    w[D * D + idx] = u;
  }
  if (threadIdx.x < D) {
    const long g = j0 + threadIdx.x;
    w[2 * D * D + threadIdx.x] = (g < n) ? T(1) / L[g * ldl + g] : T(1);
  }
}

// ------------------------------------------------------------------------------------------------
// Triangular solves with one diagonal block by SUBSTITUTION (backward stable, like LAPACK trsm/trsv; an
// explicit inverse of the block loses cond(L_jj) * eps, which is fatal for the near-singular ExpQuad
// Gram matrices the reference's jitter rule is tuned for).  One thread solves one right-hand side of
// 128 unknowns held in registers; the matrix entries are wave-uniform and come through scalar loads.
// Right-looking (axpy) form in 8 steps of 16 unknowns: the current 16 always sit in x[0..15]
// (x[112..127] for the transposed solve), are solved against the 16x16 diagonal sub-block, applied to the
// remaining unknowns (independent FMAs: the DP pipe stays full with one wave per SIMD), handed to
// `emit`, and the register array is rotated by 16 -- so the step loop is a RUNTIME loop whose body is
// ~2000 FMAs instead of an 8128-FMA straight line.
constexpr int SUBST_NT = 128;   // threads (right-hand sides) per workgroup: 2 waves share one LDS image of the block
constexpr int SUBST_HALF = LGP_DB / 2;

// stage 64 rows (row0 .. row0+63) of the 128x128 image `src` into LDS (64 KB for fp64: the kernel must fit
// beside a 73 KB trailing-update workgroup so that the look-ahead stream is never starved of a CU)
template <typename T>
__device__ __forceinline__ void subst_stage(const T *__restrict__ src, int row0, T *sM) {
  constexpr int D = LGP_DB;
  constexpr int VEC = 16 / sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
  const vec_t *s = reinterpret_cast<const vec_t *>(src + (long)row0 * D);
  vec_t *d = reinterpret_cast<vec_t *>(sM);
  for (int idx = threadIdx.x; idx < SUBST_HALF * D / VEC; idx += SUBST_NT) d[idx] = s[idx];
}

// 16 consecutive matrix entries from LDS into registers (ds_read_b128: all lanes read the same address)
template <typename T>
__device__ __forceinline__ void load16(const T *p, T (&m)[16]) {
  constexpr int VEC = 16 / sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
#pragma unroll
  for (int i = 0; i < 16 / VEC; ++i) {
    const vec_t v = reinterpret_cast<const vec_t *>(p)[i];
#pragma unroll
    for (int e = 0; e < VEC; ++e) m[i * VEC + e] = v[e];
  }
}

// xs[XO .. XO+15] -= sum_t M[t][0..15] * piv[PO + t], t = 0..15 (REV: 15..0), M rows `stride` apart in
// LDS.  Software-pipelined by hand: the rows of steps t+1 and t+2 are already in flight while step t's 16
// independent FMAs issue (an LDS read takes ~130 cycles, 16 fp64 FMAs 64: two rows ahead covers it).
template <typename T, bool REV, int XO, int PO, int NX, int NP>
__device__ __forceinline__ void rank16_update(const T *M, int stride, T (&xs)[NX], const T (&piv)[NP]) {
  T m[3][16];
  load16<T>(M + (REV ? 15 : 0) * stride, m[0]);
  load16<T>(M + (REV ? 14 : 1) * stride, m[1]);
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    if (s + 2 < 16) load16<T>(M + (REV ? 15 - (s + 2) : s + 2) * stride, m[(s + 2) % 3]);
    const int t = REV ? 15 - s : s;
#pragma unroll
    for (int e = 0; e < 16; ++e) xs[XO + e] -= m[s % 3][e] * piv[PO + t];
  }
}

constexpr int SUBST_W = 64;   // unknowns held in registers at a time

// Solve the 64 unknowns of one register window against the 64x64 diagonal sub-block whose rows are the
// staged LDS rows 0..63 and whose columns start at `coff`:  forward (L x = b, image = Dt) or backward
// (L^T x = b, image = D), 16 unknowns at a time: substitution inside the 16x16 block, rank-16 updates of
// the window's other groups.
template <typename T, bool TRANS>
__device__ __forceinline__ void solve_window(const T *sM, int coff, const T *rdw, T (&xw)[SUBST_W]) {
  constexpr int D = LGP_DB, S = 16;
  const T *M0 = sM + coff;
  if (!TRANS) {
#define LGP_FWD_GROUP(G)                                                                       \
  {                                                                                            \
    const T *Mb = M0 + (G * S) * D + G * S;                                                    \
    _Pragma("unroll") for (int t = 0; t < S; ++t) {                                            \
      const T xt = xw[G * S + t] * rdw[G * S + t];                                             \
      xw[G * S + t] = xt;                                                                      \
      _Pragma("unroll") for (int c = t + 1; c < S; ++c) xw[G * S + c] -= Mb[t * D + c] * xt;   \
    }                                                                                          \
  }
    LGP_FWD_GROUP(0)
    rank16_update<T, false, 1 * S, 0 * S>(M0 + (0 * S) * D + 1 * S, D, xw, xw);
    rank16_update<T, false, 2 * S, 0 * S>(M0 + (0 * S) * D + 2 * S, D, xw, xw);
    rank16_update<T, false, 3 * S, 0 * S>(M0 + (0 * S) * D + 3 * S, D, xw, xw);
    LGP_FWD_GROUP(1)
    rank16_update<T, false, 2 * S, 1 * S>(M0 + (1 * S) * D + 2 * S, D, xw, xw);
    rank16_update<T, false, 3 * S, 1 * S>(M0 + (1 * S) * D + 3 * S, D, xw, xw);
    LGP_FWD_GROUP(2)
    rank16_update<T, false, 3 * S, 2 * S>(M0 + (2 * S) * D + 3 * S, D, xw, xw);
    LGP_FWD_GROUP(3)
#undef LGP_FWD_GROUP
  } else {
#define LGP_BWD_GROUP(G)                                                                       \
  {                                                                                            \
    const T *Mr = M0 + (G * S) * D + G * S;                                                    \
    _Pragma("unroll") for (int t = S - 1; t >= 0; --t) {                                       \
      const T xt = xw[G * S + t] * rdw[G * S + t];                                             \
      xw[G * S + t] = xt;                                                                      \
      _Pragma("unroll") for (int c = 0; c < t; ++c) xw[G * S + c] -= Mr[t * D + c] * xt;       \
    }                                                                                          \
  }
    LGP_BWD_GROUP(3)
    rank16_update<T, true, 2 * S, 3 * S>(M0 + (3 * S) * D + 2 * S, D, xw, xw);
    rank16_update<T, true, 1 * S, 3 * S>(M0 + (3 * S) * D + 1 * S, D, xw, xw);
    rank16_update<T, true, 0 * S, 3 * S>(M0 + (3 * S) * D + 0 * S, D, xw, xw);
    LGP_BWD_GROUP(2)
    rank16_update<T, true, 1 * S, 2 * S>(M0 + (2 * S) * D + 1 * S, D, xw, xw);
    rank16_update<T, true, 0 * S, 2 * S>(M0 + (2 * S) * D + 0 * S, D, xw, xw);
    LGP_BWD_GROUP(1)
    rank16_update<T, true, 0 * S, 1 * S>(M0 + (1 * S) * D + 0 * S, D, xw, xw);
    LGP_BWD_GROUP(0)
#undef LGP_BWD_GROUP
  }
}

// One right-hand side of 128 unknowns per thread, processed as two register windows of 64 (the kernel
// must stay below ~270 registers so that its waves can be placed on a SIMD next to a wave of the trailing
// update: otherwise the look-ahead stream starves).  ld(i) / st(i, v) access unknown i of this thread's
// right-hand side in memory; the already solved window is re-read from there (L2) 16 values at a time.
template <typename T, bool TRANS, typename Ld, typename St>
__device__ __forceinline__ void subst128(const T *__restrict__ w, T *sM, Ld ld, St st) {
  constexpr int D = LGP_DB, S = 16, W = SUBST_W;
  T *rd = sM + SUBST_HALF * D;
  for (int idx = threadIdx.x; idx < D; idx += SUBST_NT) rd[idx] = w[2 * D * D + idx];
  const T *__restrict__ img = TRANS ? w : w + D * D;   // D for the transposed solve, Dt for the forward one
  subst_stage<T>(img, TRANS ? SUBST_HALF : 0, sM);
  __syncthreads();
  T xw[W];
#pragma unroll 1
  for (int win = 0; win < 2; ++win) {
    const int wbase = (TRANS ? 1 - win : win) * W;       // first unknown of this window
    const int obase = W - wbase;                         // first unknown of the other (already solved) window
#pragma unroll
    for (int i = 0; i < W; ++i) xw[i] = ld(wbase + i);
    if (win == 1) {
      // contributions of the solved window: staged rows are the solved unknowns, columns this window's
#pragma unroll 1
      for (int gg = 0; gg < 4; ++gg) {
        const int g = TRANS ? 3 - gg : gg;
        T piv[S];
#pragma unroll
        for (int t = 0; t < S; ++t) piv[t] = ld(obase + g * S + t);
        const T *Mg = sM + (g * S) * D + wbase;
        rank16_update<T, TRANS, 0 * S, 0>(Mg + 0 * S, D, xw, piv);
        rank16_update<T, TRANS, 1 * S, 0>(Mg + 1 * S, D, xw, piv);
        rank16_update<T, TRANS, 2 * S, 0>(Mg + 2 * S, D, xw, piv);
        rank16_update<T, TRANS, 3 * S, 0>(Mg + 3 * S, D, xw, piv);
      }
      __syncthreads();
      subst_stage<T>(img, wbase, sM);
      __syncthreads();
    }
    solve_window<T, TRANS>(sM, wbase, rd + wbase, xw);
#pragma unroll
    for (int i = 0; i < W; ++i) st(wbase + i, xw[i]);
  }
}

// left side: B (jb x nrhs, ld ldb) <- L_jj^-1 B or L_jj^-T B; thread per column (coalesced across lanes).
// FULL (jb == 128): branch-free body.  Lanes past the last column recompute that column and store the same
// values to the same addresses, so no lane guard is needed.
template <typename T, bool TRANS, bool FULL>
__global__ void __launch_bounds__(SUBST_NT) __attribute__((amdgpu_waves_per_eu(1, 1), amdgpu_num_vgpr(136)))
subst_left_kernel(const T *__restrict__ w, int jb, T *B, long ldb, int nrhs) {
  __shared__ __attribute__((aligned(16))) T sM[SUBST_HALF * LGP_DB + LGP_DB];
  int c = blockIdx.x * SUBST_NT + threadIdx.x;
  if (c >= nrhs) c = nrhs - 1;
  T *col = B + c;
  if (FULL) {
    subst128<T, TRANS>(
        w, sM, [&](int i) { return col[(long)i * ldb]; }, [&](int i, T v) { col[(long)i * ldb] = v; });
  } else {
    subst128<T, TRANS>(
        w, sM, [&](int i) { return (i < jb) ? col[(long)i * ldb] : T(0); },
        [&](int i, T v) {
          if (i < jb) col[(long)i * ldb] = v;
        });
  }
}

// right side: X (rows x jb, ld ldx) <- X L_jj^-T, i.e. every row solves L_jj x = b; thread per row
// (each lane streams its own contiguous row: 16-byte loads, whole 128-byte lines consumed through L1)
template <typename T, bool FULL>
__global__ void __launch_bounds__(SUBST_NT) __attribute__((amdgpu_waves_per_eu(1, 1), amdgpu_num_vgpr(136)))
subst_right_kernel(const T *__restrict__ w, int jb, T *X, long ldx, long rows) {
  constexpr int D = LGP_DB;
  __shared__ __attribute__((aligned(16))) T sM[SUBST_HALF * D + D];
  long r = (long)blockIdx.x * SUBST_NT + threadIdx.x;
  if (r >= rows) r = rows - 1;
  T *row = X + r * ldx;
  if (FULL) {
    subst128<T, false>(
        w, sM, [&](int i) { return row[i]; }, [&](int i, T v) { row[i] = v; });
  } else {
    subst128<T, false>(
        w, sM, [&](int i) { return (i < jb) ? row[i] : T(0); },
        [&](int i, T v) {
          if (i < jb) row[i] = v;
        });
  }
}

// ------------------------------------------------------------------------------------------------
// prep kernels

template <typename T>
__global__ void diag_scale_kernel(const T *K, long ldk, int n, T *s, T *stats) {
  // s_i = 2^rint(log2(K_ii)/2) where K_ii != 0 else 1   (_decomp.py:356-361)
  // stats[0] = min s^2 (atomic), stats[1] = count of s != 1 (as T)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const T d = K[(long)i * ldk + i];
  T si = T(1);
  if (d != T(0)) si = exp2(rint(T(0.5) * log2(d)));
  s[i] = si;
  if (si != T(1)) {
    atomicAdd(reinterpret_cast<unsigned int *>(stats) + 8, 1u);   // stats[4] as uint counter (T >= 4 bytes)
  }
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

template <typename T>
__device__ __forceinline__ T block_sum(T v, T *sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sh[w] = v;
  __syncthreads();
  T r = T(0);
  if (threadIdx.x < (blockDim.x >> 6)) r = sh[threadIdx.x];
  if (w == 0) r = wave_sum(r);
  __syncthreads();
  return r;   // valid on thread 0
}

// Gershgorin row sums of |K_ij / s_j / s_i|, one workgroup per row; per-row result to rows[i]
template <typename T>
__global__ void __launch_bounds__(256) rowabs_kernel(const T *K, long ldk, int n, const T *s, int unit_scale, T *rows) {
  __shared__ T sh[4];
  const int i = blockIdx.x;
  const T *row = K + (long)i * ldk;
  const T si = s[i];
  T acc = T(0);
  for (int j = threadIdx.x; j < n; j += 256) {
    T v = row[j];
    if (!unit_scale) v = v / s[j] / si;
    acc += fabs(v);
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) rows[i] = acc;
}

// single workgroup: maxeig = max_i rows[i] (NaN-propagating like jnp.max), min s^2, eps
template <typename T>
__global__ void __launch_bounds__(256) eps_kernel(const T *rows, const T *s, int n, T epsrel, T epsabs, int flags, T *out) {
  __shared__ T shmax[256], shmin[256];
  __shared__ int shnan[256];
  T mx = -INFINITY, mn = INFINITY;
  int nan = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const T v = rows[i];
    if (v != v) nan = 1;
    mx = v > mx ? v : mx;
    const T s2 = s[i] * s[i];
    if (s2 != s2) nan = 1;
    mn = s2 < mn ? s2 : mn;
  }
  shmax[threadIdx.x] = mx; shmin[threadIdx.x] = mn; shnan[threadIdx.x] = nan;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      shmax[threadIdx.x] = shmax[threadIdx.x + st] > shmax[threadIdx.x] ? shmax[threadIdx.x + st] : shmax[threadIdx.x];
      shmin[threadIdx.x] = shmin[threadIdx.x + st] < shmin[threadIdx.x] ? shmin[threadIdx.x + st] : shmin[threadIdx.x];
      shnan[threadIdx.x] |= shnan[threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const T meps = sizeof(T) == 8 ? T(2.220446049250313e-16) : T(1.1920928955078125e-07);
    T er = epsrel, ea = epsabs;
    if (flags & LGP_EPS_AUTO_REL) er = T(n) * meps;      // _decomp.py:248-249
    if (flags & LGP_EPS_AUTO_ABS) ea = meps;             // :250-251
    T maxeig = shnan[0] ? T(NAN) : shmax[0];
    const T eps = er * maxeig + ea;                      // :254
    out[0] = eps;
    out[1] = eps * (shnan[0] ? T(NAN) : shmin[0]);       // :393
    out[2] = maxeig;
  }
}

// L_ij = K_ij / s_j / s_i (+ eps on the diagonal) for j <= i, 0 above the diagonal
template <typename T>
__global__ void __launch_bounds__(256) prep_lower_kernel(const T *K, long ldk, int n, int nbx, const T *s, int unit_scale,
                                                         const T *epsp, T *L, long ldl) {
  const int i = blockIdx.x / nbx;
  const int j = (blockIdx.x % nbx) * 256 + threadIdx.x;
  if (j >= n) return;
  T v = T(0);
  if (j <= i) {
    v = K[(long)i * ldk + j];
    if (!unit_scale) v = v / s[j] / s[i];
    if (j == i) v += epsp[0];
  }
  L[(long)i * ldl + j] = v;
}

template <typename T>
__global__ void __launch_bounds__(256) scale_rows_kernel(T *L, long ldl, int n, int nbx, const T *s) {
  const int i = blockIdx.x / nbx;
  const int j = (blockIdx.x % nbx) * 256 + threadIdx.x;
  if (j > i || j >= n) return;
  L[(long)i * ldl + j] *= s[i];
}

}  // namespace

// ================================================================================================
// host drivers

template <typename T>
int lgp_potf2_launch(lgp_ctx *ctx, hipStream_t stream, T *A, size_t lda, int nb, T *dblk, size_t row0) {
  LGP_REQUIRE(nb >= 1 && nb <= LGP_DB, "diagonal block size must be in [1,128]");
  hipLaunchKernelGGL((potf2_kernel<T>), dim3(1), dim3(256), 0, stream, A, (long)lda, nb, dblk, ctx->d_info, (long)row0);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

// Factor a block-column panel: P is rows x nb (ld ldp), top nb x nb block is the diagonal block.
// dblk receives ceil(nb/128) diagonal-block workspaces.
template <typename T>
int lgp_panel_factor_impl(lgp_ctx *ctx, hipStream_t stream, size_t rows, size_t nb, T *P, size_t ldp, T *dblk, size_t row0) {
  LGP_REQUIRE(rows >= nb, "panel must have at least nb rows");
  for (size_t j0 = 0; j0 < nb; j0 += LGP_DB) {
    const size_t jb = (nb - j0) < (size_t)LGP_DB ? (nb - j0) : (size_t)LGP_DB;
    T *Dg = P + j0 * ldp + j0;
    T *w = dblk + (j0 / LGP_DB) * (size_t)LGP_DBLK;
    LGP_CHECK(lgp_potf2_launch<T>(ctx, stream, Dg, ldp, (int)jb, w, row0 + j0));
    const size_t below = rows - (j0 + jb);
    if (below > 0) {
      T *X = P + (j0 + jb) * ldp + j0;   // below x jb, solved in place: X <- X * L_jj^-T
      const dim3 gr((unsigned)((below + SUBST_NT - 1) / SUBST_NT)), br(SUBST_NT);
      if (jb == (size_t)LGP_DB) hipLaunchKernelGGL((subst_right_kernel<T, true>), gr, br, 0, stream, w, (int)jb, X, (long)ldp, (long)below);
      else hipLaunchKernelGGL((subst_right_kernel<T, false>), gr, br, 0, stream, w, (int)jb, X, (long)ldp, (long)below);
      LGP_LAUNCH_CHECK();
      const size_t wc = nb - (j0 + jb);
      if (wc > 0) {
        // remaining panel columns: C (below x wc, C(0,0) on the diagonal) -= X * X[0:wc]^T, lower part
        T *C = P + (j0 + jb) * ldp + (j0 + jb);
        LGP_CHECK(lgp_gemm_launch<T>(ctx, stream, false, true, below, wc, jb, T(-1), X, ldp, X, ldp, T(1), C, ldp, true));
      }
    }
  }
  return LGP_OK;
}

template <typename T>
int lgp_potrf_impl(lgp_ctx *ctx, size_t n, T *L, size_t ldl, T *dblk) {
  LGP_REQUIRE(ldl >= n, "ldl < n");
  hipStream_t s0 = ctx->stream, s1 = ctx->stream2;
  const size_t NB = LGP_NB;
  // look-ahead: panel k+1 is updated and factored on s1 while s0 finishes the trailing update of step k
  for (size_t k0 = 0; k0 < n; k0 += NB) {
    const size_t kb = (n - k0) < NB ? (n - k0) : NB;
    T *P = L + k0 * ldl + k0;
    T *Yk = dblk + (k0 / LGP_DB) * (size_t)LGP_DBLK;
    if (k0 == 0) {
      LGP_CHECK(lgp_panel_factor_impl<T>(ctx, s0, n - k0, kb, P, ldl, Yk, k0));
    }
    // (panel k0 is complete on s0 at this point: either factored above or joined from s1 below)
    const size_t t0 = k0 + kb;
    if (t0 >= n) break;
    const size_t rows_t = n - t0;
    const T *Pt = L + t0 * ldl + k0;            // rows_t x kb: the panel rows below the diagonal block
    const size_t nb1 = rows_t < NB ? rows_t : NB;  // width of the next panel
    // s1: wait for panel k0, update the next panel's columns, factor it
    LGP_HIP(hipEventRecord(ctx->ev_a, s0));
    LGP_HIP(hipStreamWaitEvent(s1, ctx->ev_a, 0));
    T *C1 = L + t0 * ldl + t0;                  // rows_t x nb1 block column (lower part)
    LGP_CHECK(lgp_gemm_launch<T>(ctx, s1, false, true, rows_t, nb1, kb, T(-1), Pt, ldl, Pt, ldl, T(1), C1, ldl, true));
    LGP_CHECK(lgp_panel_factor_impl<T>(ctx, s1, rows_t, nb1, C1, ldl, dblk + (t0 / LGP_DB) * (size_t)LGP_DBLK, t0));
    LGP_HIP(hipEventRecord(ctx->ev_b, s1));
    // s0: rest of the trailing update (columns beyond the next panel), issued as block-column chunks of
    // roughly equal area.  A single launch would keep every CU's registers and LDS occupied until its
    // grid drains, and the (few, large-footprint) panel workgroups of s1 would only be placed then; at
    // each chunk boundary the high-priority stream gets its next kernel in, so the panel chain advances
    // one kernel per chunk while the chunks keep the other CUs busy.
    if (rows_t > nb1) {
      const size_t rows_r = rows_t - nb1;
      const size_t base = t0 + nb1;
      size_t nch = rows_r / 1024;
      if (nch > LGP_TRAIL_CHUNKS) nch = LGP_TRAIL_CHUNKS;
      if (nch < 1) nch = 1;
      const double total = 0.5 * (double)rows_r * (double)rows_r;
      size_t c0 = 0;
      for (size_t ch = 0; ch < nch && c0 < rows_r; ++ch) {
        size_t c1 = rows_r;
        if (ch + 1 < nch) {
          // smallest multiple of 128 such that area(c0, c1) >= total / nch
          const double want = total / (double)nch;
          c1 = c0;
          double area = 0;
          while (c1 < rows_r && area < want) {
            c1 += 128;
            if (c1 > rows_r) c1 = rows_r;
            area = (double)(c1 - c0) * ((double)rows_r - 0.5 * (double)(c0 + c1));
          }
        }
        const T *Pa = L + (base + c0) * ldl + k0;          // rows below and including the chunk's diagonal
        T *Cc = L + (base + c0) * ldl + (base + c0);
        // algorithmic flops of this launch: 2 k per produced entry of the lower trapezoid
        const double entries = (double)(c1 - c0) * ((double)(rows_r - c0) - 0.5 * (double)(c1 - c0 - 1));
        LGP_CHECK(lgp_prof_start(ctx, s0, 2.0 * (double)kb * entries));
        LGP_CHECK(lgp_gemm_launch<T>(ctx, s0, false, true, rows_r - c0, c1 - c0, kb, T(-1), Pa, ldl, Pa, ldl, T(1), Cc, ldl, true));
        LGP_CHECK(lgp_prof_stop(ctx, s0));
        c0 = c1;
      }
    }
    LGP_HIP(hipStreamWaitEvent(s0, ctx->ev_b, 0));
  }
  return LGP_OK;
}

template <typename T>
int lgp_trsm_impl(lgp_ctx *ctx, hipStream_t stream, size_t n, const T *L, size_t ldl, const T *dblk, int trans, T *B, size_t ldb, size_t nrhs) {
  if (n == 0 || nrhs == 0) return LGP_OK;
  LGP_REQUIRE(ldl >= n && ldb >= nrhs, "leading dimension too small");
  LGP_REQUIRE(nrhs < (1u << 30), "too many right-hand sides");
  const size_t nblk = (n + LGP_DB - 1) / LGP_DB;
  const dim3 gs((unsigned)((nrhs + SUBST_NT - 1) / SUBST_NT)), bs(SUBST_NT);
  if (!trans) {
    // forward: B_j <- L_jj^-1 B_j (substitution), then B_below -= L[below, j] B_j (MFMA gemm)
    for (size_t b = 0; b < nblk; ++b) {
      const size_t j0 = b * LGP_DB, jb = (n - j0) < (size_t)LGP_DB ? (n - j0) : (size_t)LGP_DB;
      T *Bj = B + j0 * ldb;
      if (jb == (size_t)LGP_DB) hipLaunchKernelGGL((subst_left_kernel<T, false, true>), gs, bs, 0, stream, dblk + b * (size_t)LGP_DBLK, (int)jb, Bj, (long)ldb, (int)nrhs);
      else hipLaunchKernelGGL((subst_left_kernel<T, false, false>), gs, bs, 0, stream, dblk + b * (size_t)LGP_DBLK, (int)jb, Bj, (long)ldb, (int)nrhs);
      LGP_LAUNCH_CHECK();
      const size_t below = n - (j0 + jb);
      if (below > 0)
        LGP_CHECK(lgp_gemm_launch<T>(ctx, stream, false, false, below, nrhs, jb, T(-1), L + (j0 + jb) * ldl + j0, ldl, Bj, ldb, T(1),
                                     B + (j0 + jb) * ldb, ldb, false));
    }
  } else {
    // backward: B_j <- L_jj^-T B_j, then B_above -= L[j, above]^T B_j
    for (size_t bb = nblk; bb-- > 0;) {
      const size_t j0 = bb * LGP_DB, jb = (n - j0) < (size_t)LGP_DB ? (n - j0) : (size_t)LGP_DB;
      T *Bj = B + j0 * ldb;
      if (jb == (size_t)LGP_DB) hipLaunchKernelGGL((subst_left_kernel<T, true, true>), gs, bs, 0, stream, dblk + bb * (size_t)LGP_DBLK, (int)jb, Bj, (long)ldb, (int)nrhs);
      else hipLaunchKernelGGL((subst_left_kernel<T, true, false>), gs, bs, 0, stream, dblk + bb * (size_t)LGP_DBLK, (int)jb, Bj, (long)ldb, (int)nrhs);
      LGP_LAUNCH_CHECK();
      if (j0 > 0)
        LGP_CHECK(lgp_gemm_launch<T>(ctx, stream, true, false, j0, nrhs, jb, T(-1), L + j0 * ldl, ldl, Bj, ldb, T(1), B, ldb, false));
    }
  }
  return LGP_OK;
}

template <typename T>
static int chol_factor_impl(lgp_ctx *ctx, size_t n, const T *K, size_t ldk, T epsrel, T epsabs, int flags, T *L, size_t ldl,
                            T *dblk, T *heps_out) {
  LGP_REQUIRE(ctx != nullptr, "null context");
  LGP_REQUIRE(n >= 1 && n < (1u << 30), "matrix size out of range");
  LGP_REQUIRE(K != nullptr && L != nullptr && dblk != nullptr, "null pointer");
  LGP_REQUIRE(ldk >= n && ldl >= n, "leading dimension too small");
  DeviceGuard dg(ctx->device);
  hipStream_t st = ctx->stream;
  // workspace: s (n), rows (n), stats (16)
  T *ws = nullptr;
  LGP_HIP(hipMallocAsync(reinterpret_cast<void **>(&ws), (2 * n + 16) * sizeof(T), st));
  T *s = ws, *rows = ws + n, *stats = ws + 2 * n;
  LGP_HIP(hipMemsetAsync(stats, 0, 16 * sizeof(T), st));
  LGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), st));
  const int ni = (int)n;
  hipLaunchKernelGGL((diag_scale_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, K, (long)ldk, ni, s, stats);
  LGP_LAUNCH_CHECK();
  // is the scaling trivial?  (one small sync; lets the big passes skip two divisions per element)
  unsigned int nonunit = 0;
  LGP_HIP(hipMemcpyAsync(&nonunit, reinterpret_cast<unsigned int *>(stats) + 8, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
  LGP_HIP(hipStreamSynchronize(st));
  const int unit = nonunit == 0 ? 1 : 0;
  hipLaunchKernelGGL((rowabs_kernel<T>), dim3((unsigned)n), dim3(256), 0, st, K, (long)ldk, ni, s, unit, rows);
  LGP_LAUNCH_CHECK();
  hipLaunchKernelGGL((eps_kernel<T>), dim3(1), dim3(256), 0, st, rows, s, ni, epsrel, epsabs, flags, stats);
  LGP_LAUNCH_CHECK();
  const int nbx = (int)((n + 255) / 256);
  hipLaunchKernelGGL((prep_lower_kernel<T>), dim3((unsigned)((size_t)nbx * n)), dim3(256), 0, st, K, (long)ldk, ni, nbx, s, unit,
                     stats, L, (long)ldl);
  LGP_LAUNCH_CHECK();
  LGP_CHECK(lgp_potrf_impl<T>(ctx, n, L, ldl, dblk));
  if (!unit) {
    hipLaunchKernelGGL((scale_rows_kernel<T>), dim3((unsigned)((size_t)nbx * n)), dim3(256), 0, st, L, (long)ldl, ni, nbx, s);
    LGP_LAUNCH_CHECK();
    hipLaunchKernelGGL((extract_dblk_kernel<T>), dim3((unsigned)((n + LGP_DB - 1) / LGP_DB)), dim3(256), 0, st, L, (long)ldl, ni, dblk);
    LGP_LAUNCH_CHECK();
  }
  T hstats[3];
  int info = 0;
  LGP_HIP(hipMemcpyAsync(hstats, stats, 3 * sizeof(T), hipMemcpyDeviceToHost, st));
  LGP_HIP(hipMemcpyAsync(&info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, st));
  LGP_HIP(hipFreeAsync(ws, st));
  LGP_HIP(hipStreamSynchronize(st));
  if (heps_out) *heps_out = hstats[1];
  return info;
}

extern "C" size_t lgp_chol_dblk_size(size_t n) { return ((n + LGP_DB - 1) / LGP_DB) * (size_t)LGP_DBLK; }

extern "C" int lgp_chol_factor_f64(lgp_ctx *ctx, size_t n, const double *dK, size_t ldk, double epsrel, double epsabs, int eps_flags,
                                   double *dL, size_t ldl, double *ddblk, double *heps_out) {
  return chol_factor_impl<double>(ctx, n, dK, ldk, epsrel, epsabs, eps_flags, dL, ldl, ddblk, heps_out);
}
extern "C" int lgp_chol_factor_f32(lgp_ctx *ctx, size_t n, const float *dK, size_t ldk, float epsrel, float epsabs, int eps_flags,
                                   float *dL, size_t ldl, float *ddblk, float *heps_out) {
  return chol_factor_impl<float>(ctx, n, dK, ldk, epsrel, epsabs, eps_flags, dL, ldl, ddblk, heps_out);
}

template <typename T>
static int potrf_entry(lgp_ctx *ctx, size_t n, T *L, size_t ldl, T *dblk) {
  LGP_REQUIRE(ctx != nullptr && L != nullptr && dblk != nullptr, "null pointer");
  LGP_REQUIRE(n >= 1 && n < (1u << 30), "matrix size out of range");
  DeviceGuard dg(ctx->device);
  LGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), ctx->stream));
  LGP_CHECK(lgp_potrf_impl<T>(ctx, n, L, ldl, dblk));
  int info = 0;
  LGP_HIP(hipMemcpyAsync(&info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  LGP_HIP(hipStreamSynchronize(ctx->stream));
  return info;
}
extern "C" int lgp_potrf_f64(lgp_ctx *ctx, size_t n, double *dL, size_t ldl, double *ddblk) { return potrf_entry<double>(ctx, n, dL, ldl, ddblk); }
extern "C" int lgp_potrf_f32(lgp_ctx *ctx, size_t n, float *dL, size_t ldl, float *ddblk) { return potrf_entry<float>(ctx, n, dL, ldl, ddblk); }

extern "C" int lgp_trsm_f64(lgp_ctx *ctx, size_t n, const double *dL, size_t ldl, const double *ddblk, int trans, double *dB, size_t ldb,
                            size_t nrhs) {
  LGP_REQUIRE(ctx != nullptr && dL != nullptr && ddblk != nullptr && (dB != nullptr || nrhs == 0), "null pointer");
  DeviceGuard dg(ctx->device);
  return lgp_trsm_impl<double>(ctx, ctx->stream, n, dL, ldl, ddblk, trans, dB, ldb, nrhs);
}
extern "C" int lgp_trsm_f32(lgp_ctx *ctx, size_t n, const float *dL, size_t ldl, const float *ddblk, int trans, float *dB, size_t ldb,
                            size_t nrhs) {
  LGP_REQUIRE(ctx != nullptr && dL != nullptr && ddblk != nullptr && (dB != nullptr || nrhs == 0), "null pointer");
  DeviceGuard dg(ctx->device);
  return lgp_trsm_impl<float>(ctx, ctx->stream, n, dL, ldl, ddblk, trans, dB, ldb, nrhs);
}

extern "C" int lgp_panel_factor_f64(lgp_ctx *ctx, size_t rows, size_t nb, double *dP, size_t ldp, double *ddblk, size_t row0) {
  LGP_REQUIRE(ctx != nullptr && dP != nullptr && ddblk != nullptr, "null pointer");
  DeviceGuard dg(ctx->device);
  return lgp_panel_factor_impl<double>(ctx, ctx->stream, rows, nb, dP, ldp, ddblk, row0);
}

extern "C" int lgp_get_info(lgp_ctx *ctx, int *hinfo) {
  LGP_REQUIRE(ctx != nullptr && hinfo != nullptr, "null pointer");
  DeviceGuard dg(ctx->device);
  LGP_HIP(hipMemcpyAsync(hinfo, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  LGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), ctx->stream));
  LGP_HIP(hipStreamSynchronize(ctx->stream));
  return LGP_OK;
}

template int lgp_panel_factor_impl<double>(lgp_ctx *, hipStream_t, size_t, size_t, double *, size_t, double *, size_t);
template int lgp_panel_factor_impl<float>(lgp_ctx *, hipStream_t, size_t, size_t, float *, size_t, float *, size_t);
template int lgp_potrf_impl<double>(lgp_ctx *, size_t, double *, size_t, double *);
template int lgp_potrf_impl<float>(lgp_ctx *, size_t, float *, size_t, float *);
template int lgp_trsm_impl<double>(lgp_ctx *, hipStream_t, size_t, const double *, size_t, const double *, int, double *, size_t, size_t);
template int lgp_trsm_impl<float>(lgp_ctx *, hipStream_t, size_t, const float *, size_t, const float *, int, float *, size_t, size_t);

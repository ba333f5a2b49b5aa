// potrf.hip -- regularised blocked Cholesky for gfx950: the decomposition step of the hot path,
// src/lsqfitgp/_linalg/_decomp.py:380-393 (Chol.__init__), and the blocked triangular solves built on it
// (jlinalg.solve_triangular call sites, _decomp.py:402-439,467).
//
// Structure (row-major, lower):
//   prep      : s = diag_scale_pow2(K) (:356-361); Gershgorin bound of K/s/s' (:349-354); eps (:245-255);
//               L <- tril(K/s/s') + eps*I with the strict upper triangle zeroed      [HBM-bound passes]
//   potrf     : right-looking, outer panel LGP_NB = 1024, inner diagonal blocks LGP_DB = 128:
//                 potf2       : one workgroup factors a 128x128 diagonal block held in REGISTERS
//                               (8x8 cyclic tile per thread);
//                 panel trsm  : X = B * L_jj^-T by substitution on the matrix cores (16 rows per wave as 16x16
//                               accumulator tiles, operands from a fragment image packed per diagonal block);
//                 panel update: gemm NT, K = 128;
//                 trailing    : syrk (gemm NT, lower tiles only), K = 1024 -- the N^3/3 flops, MFMA-bound;
//               the next panel is factored on a second stream while the rest of the trailing update
//               runs (look-ahead); optional extra rows (right-hand sides) ride through all of it.
//   finish    : L <- L * s[:,None], diagonal-block workspace re-extracted, eps_out = eps * min(s^2).
// A non-positive or non-finite pivot is recorded in ctx->d_info (first one wins) and the factor goes
// NaN from there on, like LAPACK-in-jax; the host turns info != 0 into the reference's LinAlgError.
#include "chain.h"
#include <cstdlib>

int lgp_info_result(lgp_ctx *ctx, int info) {
  if (info != LGP_INFO_STALLED) return info;
  // a stall means that this process does not run the context's streams side by side the way the self-test saw them
  // (a stream added since then shares a hardware queue with the chain kernel's): later calls use the launch-per-operation driver
  ctx->chain_mode = 1;
  lgp_set_error("the panel task graph stalled (a worker waited longer than %.1f s for its inputs); the factor is incomplete; "
                "this context now uses the launch-per-operation driver", (double)ctx->chain_timeout / 1e8);
  return LGP_EHIP;
}

namespace {

// restores ctx->gemm_thin_lds (the thinning of the factorisation's updates) on every way out of a driver
// one wave that leaves after `ticks` of the 100 MHz clock (LGP_LOOKAHEAD_DELAY)
__global__ void lookahead_delay_kernel(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

struct ThinGuard {
  lgp_ctx *c;
  explicit ThinGuard(lgp_ctx *ctx) : c(ctx) { c->gemm_thin_lds = 0; c->gemm_prio = 0; }
  ~ThinGuard() { c->gemm_thin_lds = 0; c->gemm_prio = 0; }
};

// ------------------------------------------------------------------------------------------------
// Diagonal-block workspace ("dblk"): for every 128x128 diagonal block of L, LGP_DBLK elements:
//   [0, 128*128)          D   = the strictly lower part of the block, row-major (zero diagonal and above)
//   [128*128, 2*128*128)  Dt  = its transpose, row-major
//   [2*128*128, +128)     rd  = 1 / diag (1 past n)
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
    // only the register tiles that hold lower-triangle elements (p >= q; i = ti + 16 p, c = tj + 16 q): the rest of the
    // matrix is never read, and the update is 45 % of the vector work of a pivot step
#pragma unroll
    for (int p = JA; p < 8; ++p)
#pragma unroll
      for (int q = JA; q <= p; ++q) a[p][q] -= lr[p] * lc[q];   // zero factors where i <= j or c <= j
  }
}

template <typename T>
__global__ void __launch_bounds__(256, 1) potf2_kernel(T *A, long lda, int nb, T *dblk, int *info, long row0) {
  // LDS: only the 2 KB double-buffered column, so that this single workgroup can always be placed next
  // to the 73 KB workgroups of the trailing update running on the other stream (look-ahead).
  constexpr int D = LGP_DB;        // 128
  __shared__ T scol[2][D];
  // This workgroup is the head of the factorisation's dependent chain and shares its CU with two long-running
  // workgroups of the trailing update; VALU issue is arbitrated by priority, then age, so without this the
  // (younger) chain waves only get the slots the MFMA streams leave over (3x slower in situ than alone).
  __builtin_amdgcn_s_setprio(LGP_CHAIN_PRIO);

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
      const T off = (i == c) ? T(0) : v;     // the workspace images carry a ZERO diagonal, 1/diag is kept apart
      dblk[i * D + c] = off;
      dblk[D * D + c * D + i] = off;
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
    // strictly lower element (i, c), c < i (zero diagonal, see potf2_kernel)
    T v = T(0);
    if (c < i && j0 + i < n) v = L[(j0 + i) * ldl + j0 + c];
    w[idx] = v;
    // transposed image: Dt[i][c] = D[c][i], nonzero for i < c
    T u = T(0);
    if (i < c && j0 + c < n) u = L[(j0 + c) * ldl + j0 + i];
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
// Gram matrices the reference's jitter rule is tuned for).
//
// One WAVE solves R = 4 right-hand sides at once; lane l holds unknowns l and 64 + l of each.  Step t
// (forward: t = 0..127, L x = b, image Dt whose row t is column t of L; transposed: t = 127..0, image D):
//     y   = x_half * rd_half          (all lanes; lane t&63 now holds the final x_t)
//     x_t = readlane(y, t & 63)       (wave-uniform, goes to the FMA as a scalar operand)
//     x  -= M[t][lane] * x_t          (the images have a ZERO diagonal and zeros on the finished side, so
//                                      finished unknowns keep their unscaled value and no mask is needed)
// and at the end x * rd is the solution in every lane.  The matrix rows stream straight from L2 (coalesced
// 512-byte rows, software-prefetched 8 rows ahead); no LDS and ~50 registers, so these waves slot in beside
// the two trailing-update workgroups resident on every CU and the look-ahead chain never waits for a CU.
// With 4 independent chains per wave the 32-cycle fp64 latency is covered at any occupancy.
constexpr int SUBST_R = 4;        // right-hand sides per wave
constexpr int SUBST_PF = 8;       // matrix rows in flight

template <typename T>
__device__ __forceinline__ T readlane_any(T v, int lane) { return readlane_t<T>(v, lane); }

template <typename T, bool TRANS, typename Ld, typename St>
__device__ __forceinline__ void subst_wave(const T *__restrict__ w, Ld ld, St st) {
  constexpr int D = LGP_DB, R = SUBST_R, PF = SUBST_PF;
  const int lane = threadIdx.x & 63;
  const T *__restrict__ M = TRANS ? w : w + D * D;
  const T rd0 = w[2 * D * D + lane], rd1 = w[2 * D * D + 64 + lane];
  T x0[R], x1[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    x0[r] = ld(r, lane);
    x1[r] = ld(r, 64 + lane);
  }
  // prefetch ring: row of step s lives in slot s % PF
  T m0[PF], m1[PF];
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    const int t = TRANS ? D - 1 - s : s;
    m0[s] = M[t * D + lane];
    m1[s] = M[t * D + 64 + lane];
  }
#pragma unroll 1
  for (int s0 = 0; s0 < D; s0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int s = s0 + u;
      const int t = TRANS ? D - 1 - s : s;
      const T a0 = m0[u], a1 = m1[u];
      // refill this slot with the row of step s + PF (clamped: the last refills are harmless repeats)
      {
        const int sn = (s + PF) < D ? (s + PF) : (D - 1);
        const int tn = TRANS ? D - 1 - sn : sn;
        m0[u] = M[tn * D + lane];
        m1[u] = M[tn * D + 64 + lane];
      }
      const bool hi = t >= 64;          // wave-uniform
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const T yv = hi ? x1[r] * rd1 : x0[r] * rd0;
        const T xt = readlane_any<T>(yv, t & 63);
        // the half on the finished side only sees zeros of the image: skip it (wave-uniform)
        if (TRANS || !hi) x0[r] -= a0 * xt;
        if (!TRANS || hi) x1[r] -= a1 * xt;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    st(r, lane, x0[r] * rd0);
    st(r, 64 + lane, x1[r] * rd1);
  }
}

constexpr int SUBST_NT = 256;     // 4 waves = 16 right-hand sides per workgroup

// left side: B (jb x nrhs, ld ldb) <- L_jj^-1 B or L_jj^-T B; right-hand sides are the columns of B
template <typename T, bool TRANS>
__global__ void __launch_bounds__(SUBST_NT) subst_left_kernel(const T *__restrict__ w, int jb, T *B, long ldb, int nrhs) {
  const int c0 = (blockIdx.x * (SUBST_NT / 64) + (threadIdx.x >> 6)) * SUBST_R;
  if (c0 >= nrhs) return;     // wave-uniform
  __builtin_amdgcn_s_setprio(LGP_CHAIN_PRIO);   // latency chain next to MFMA streams, see potf2_kernel
  subst_wave<T, TRANS>(
      w,
      [&](int r, int i) {
        const int c = c0 + r;
        return (i < jb && c < nrhs) ? B[(long)i * ldb + c] : T(0);
      },
      [&](int r, int i, T v) {
        const int c = c0 + r;
        if (i < jb && c < nrhs) B[(long)i * ldb + c] = v;
      });
}

// ------------------------------------------------------------------------------------------------
// Right-side substitution on the matrix cores: X (rows x 128) <- X L_jj^-T for the panel rows below a diagonal
// block.  While the trailing update owns the chip, every vector-ALU instruction of a co-resident wave waits for a
// 64-cycle fp64 MFMA slot (tools/coexec_probe.hip: the plain substitution above, ~3000 VALU instructions per
// wave, goes from 61 us alone to 300-430 us), so this kernel is organised to need as few VALU instructions as
// possible: ~300 per wave for FOUR TIMES the rows.
//
// One wave owns 16 rows of X, i.e. 16 columns of Y = X^T (128 x 16), kept in registers as eight 16x16 MFMA
// accumulator tiles: lane (n = lane & 15, g = lane >> 4), register r of tile c holds unknown 16c + 4r + g of row n.
// With the rows of L pre-scaled by 1/diag (M = diag^-1 L, unit diagonal) the forward substitution
//     y_c = x~_c - sum_{i<c} M_ci y_i            (16x16 tiles; x~ = x / diag)
// is  (a) 4 MFMAs per tile pair (c, i): the accumulator registers of tile i ARE the B operand of k-step s = r
//         (unknowns 4s + g), the A operand is a 512-byte fragment of -M in lane order, read straight from L2;
//     (b) inside a diagonal tile, register r (unknowns 4r .. 4r+3, one per lane group) is a 4x4 unit-triangular
//         solve: 3 x (ds_bpermute broadcast of the finished group + 1 FMA), followed by ONE MFMA that applies the
//         four finished unknowns to the later registers (A operand masked to those rows).
// No LDS memory and no address arithmetic: every operand is `fragment base + lane * sizeof(T)` with a scalar
// fragment offset (raw buffer loads).  pack_frag_kernel lays the fragments out once per diagonal block.
//
// Fragment ids (64 elements each, element = lane):
//   off(c,i,s)  = (c(c-1)/2 + i) * 4 + s     -M[16c + u(n)][16i + 4s + g]                         112 fragments
//   diag(c,r)   = 112 + 3c + r  (r < 3)      -M[16c + u(n)][16c + 4r + g]  if u(n) >= 4(r+1) else 0   24
//   coef(c,r,a) = 136 + (4c + r) 3 + a       -M[16c + 4r + g][16c + 4r + a] if g > a else 0            96
//   rdv(c,r)    = 232 + 4c + r               1 / L[16c + 4r + g][16c + 4r + g]                         32
// u(n) is the unknown-in-tile that MFMA output row n carries: n for fp64 (row = g + 4r), 4(n & 3) + (n >> 2) for
// fp32 (row = 4g + r): with it both types share the (g, r) -> unknown 4r + g register map.
template <typename T>
__device__ __forceinline__ int frag_urow(int n) { return sizeof(T) == 8 ? n : 4 * (n & 3) + (n >> 2); }

// one wave per fragment (the fragment id is wave-uniform, so its decoding is scalar work); grid = LGP_NFRAG / 4
template <typename T>
__global__ void __launch_bounds__(256) pack_frag_kernel(T *dblk) {
  constexpr int D = LGP_DB;
  T *w = dblk;
  const T *Dl = w;                 // strictly lower block, zero diagonal
  const T *rd = w + 2 * D * D;
  T *F = w + LGP_FRAG0;
  const int f = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (f >= LGP_NFRAG) return;
  const int l = threadIdx.x & 63, n = l & 15, g = l >> 4;
  int u, v;          // element (u, v) of -M = -diag^-1 L, or rd[u] for the last group
  bool live = true;
  if (f < 112) {
    const int pair = f >> 2, s = f & 3;
    int c = 1;
    while ((c + 1) * c / 2 <= pair) ++c;      // pair = c(c-1)/2 + i, i < c
    const int i = pair - c * (c - 1) / 2;
    u = 16 * c + frag_urow<T>(n);
    v = 16 * i + 4 * s + g;
  } else if (f < 136) {
    const int c = (f - 112) / 3, r = (f - 112) % 3, ur = frag_urow<T>(n);
    live = ur >= 4 * (r + 1);
    u = 16 * c + ur;
    v = 16 * c + 4 * r + g;
  } else if (f < 232) {
    const int q = (f - 136) / 3, a = (f - 136) % 3, c = q >> 2, r = q & 3;
    live = g > a;
    u = 16 * c + 4 * r + g;
    v = 16 * c + 4 * r + a;
  } else {
    const int c = (f - 232) >> 2, r = (f - 232) & 3;
    u = 16 * c + 4 * r + g;
    F[f * 64 + l] = rd[u];
    return;
  }
  F[f * 64 + l] = live ? -(Dl[u * D + v] * rd[u]) : T(0);
}

template <typename T>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) subst_right_mfma_kernel(const T *__restrict__ w, int jb, T *X, long ldx, long rows) {
  __builtin_amdgcn_s_setprio(LGP_CHAIN_PRIO);
  subst_right_wave<T>(w, jb, X, ldx, rows, ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16);
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
__global__ void __launch_bounds__(256) rowabs_kernel(const T *K, long ldk, int n, const T *s, const unsigned int *nonunit, T *rows) {
  __shared__ T sh[4];
  const bool unit_scale = *nonunit == 0;      // all s_i == 1 (device-side flag: no host round trip inside the factorisation)
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
__global__ void __launch_bounds__(256) prep_lower_kernel(const T *K, long ldk, int n, int nbx, const T *s, const unsigned int *nonunit,
                                                         const T *epsp, T *L, long ldl) {
  const bool unit_scale = *nonunit == 0;
  // one workgroup per 2048-column segment of a row (grid.x * 256 must stay below 2^32: N = 65536 rows)
  const int i = blockIdx.x / nbx;
  const int jbeg = (blockIdx.x % nbx) * 2048;
  for (int j = jbeg + threadIdx.x; j < jbeg + 2048 && j < n; j += 256) {
    T v = T(0);
    if (j <= i) {
      v = K[(long)i * ldk + j];
      if (!unit_scale) v = v / s[j] / s[i];
      if (j == i) v += epsp[0];
    }
    L[(long)i * ldl + j] = v;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) scale_rows_kernel(T *L, long ldl, int n, int nbx, const T *s, const unsigned int *nonunit) {
  if (*nonunit == 0) return;
  const int i = blockIdx.x / nbx;
  const int jbeg = (blockIdx.x % nbx) * 2048;
  for (int j = jbeg + threadIdx.x; j < jbeg + 2048 && j <= i && j < n; j += 256) L[(long)i * ldl + j] *= s[i];
}

// the zeros above the diagonal of a factor made with LGP_FACTOR_LOWER_ONLY (one workgroup per 2048-column segment of a row)
template <typename T>
__global__ void __launch_bounds__(256) zero_upper_kernel(T *L, long ldl, int n, int nbx) {
  const int i = blockIdx.x / nbx;
  const int jbeg = (blockIdx.x % nbx) * 2048;
  for (int j = jbeg + threadIdx.x; j < jbeg + 2048 && j < n; j += 256)
    if (j > i) L[(long)i * ldl + j] = T(0);
}

}  // namespace

// ================================================================================================
// host drivers

// Factors one diagonal block and leaves the fragment image of the block in its workspace (read by the right-side
// substitution of the panel).  fp64: the matrix-core kernel; fp32: the VALU kernel + pack_frag_kernel.  The images that
// the left-side substitutions of the solves read (D, D^T, 1/diag) are extracted from the finished factor afterwards
// (extract_dblk_kernel), off the dependent chain of the factorisation.
template <typename T>
int lgp_potf2_launch(lgp_ctx *ctx, hipStream_t stream, T *A, size_t lda, int nb, T *dblk, size_t row0) {
  LGP_REQUIRE(nb >= 1 && nb <= LGP_DB, "diagonal block size must be in [1,128]");
  if (ctx->tracer != nullptr) {
    long long sa, oa, sw, ow;
    lgp_tr_ptr(ctx, A, sa, oa); lgp_tr_ptr(ctx, dblk, sw, ow);
    if (!LGP_TRACED(ctx, LGP_TR_POTF2, stream, sa, oa, (long long)lda, nb, ow)) return LGP_OK;
  }
  // fp64: the four-wave matrix-core kernel of potf2m.hip (46 us alone against 60 us for the VALU kernel below; end to end
  // N = 4096 4.24 -> 3.69 ms, N = 8192 10.7 -> 9.7 ms, N = 16384 39.6 -> 37.4 ms, N = 32768 204 -> 200.7 ms).
  // ctx->potf2_mode (env LGP_POTF2): 0 / 2 = that kernel, 1 = VALU kernel, 3 = the one-wave matrix-core kernel, which needs a
  // whole SIMD's register file and therefore waits for a SIMD to drain beside a large trailing update (0.58 ms on average
  // in the N = 32768 profile: 204 -> 213 ms per step).
  const bool mfma = std::is_same<T, double>::value && ctx->potf2_mode != 1;
  if constexpr (std::is_same<T, double>::value) {
    if (mfma) return lgp_potf2_mfma_launch(ctx, stream, A, lda, nb, dblk, row0);
  }
  {
    hipLaunchKernelGGL((potf2_kernel<T>), dim3(1), dim3(256), 0, stream, A, (long)lda, nb, dblk, ctx->d_info, (long)row0);
    LGP_LAUNCH_CHECK();
    hipLaunchKernelGGL((pack_frag_kernel<T>), dim3((LGP_NFRAG + 3) / 4), dim3(256), 0, stream, dblk);
    LGP_LAUNCH_CHECK();
  }
  return LGP_OK;
}

template <typename T>
int lgp_extract_dblk(lgp_ctx *ctx, hipStream_t stream, const T *L, size_t ldl, size_t n, T *dblk) {
  hipLaunchKernelGGL((extract_dblk_kernel<T>), dim3((unsigned)((n + LGP_DB - 1) / LGP_DB)), dim3(256), 0, stream, L, (long)ldl, (int)n, dblk);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

// Factor a block-column panel: P is rows x nb (ld ldp), top nb x nb block is the diagonal block.
// dblk receives ceil(nb/128) diagonal-block workspaces.
//
// Inside the panel the dependent chain is  potf2(D_j) -> rows of the NEXT diagonal block solved -> D_j+1 updated -> potf2(D_j+1):
// only those 128 rows and that one 128 x 128 update stay on `stream`.  Everything else a block produces -- the substitution
// of all the other rows below it and the update of the rest of the panel with them -- goes to the context's bulk stream, which
// follows one block behind (it needs potf2(D_j) and the look-ahead rows of block j; the chain needs bulk(j - 1) before it
// solves the look-ahead rows of block j: by then bulk(j - 1) has had a whole potf2 to finish).  Per block the chain is
// potf2 + one latency-bound substitution of 128 rows + one small product instead of potf2 + substitution of all rows +
// update of the whole rest of the panel.
template <typename T>
int lgp_panel_factor_impl(lgp_ctx *ctx, hipStream_t stream, size_t rows, size_t nb, T *P, size_t ldp, T *dblk, size_t row0) {
  LGP_REQUIRE(rows >= nb, "panel must have at least nb rows");
  const size_t nblk = (nb + LGP_DB - 1) / LGP_DB;
  hipStream_t bulk = ctx->stream3;
  // measured: the split costs more in cross-stream event hops (three per block) than it takes off the chain -- N = 4096
  // 3.64 -> 4.25 ms, N = 8192 9.8 -> 10.7 ms, N = 16384 38.4 -> 40.6 ms -- so it is off unless LGP_PANEL_SPLIT=1
  static const bool want_split = getenv("LGP_PANEL_SPLIT") != nullptr;
  const bool split = want_split && bulk != nullptr && bulk != stream && nblk > 1;
  while (ctx->panel_events.size() < 3 * nblk + 1) {
    hipEvent_t e;
    LGP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->panel_events.push_back(e);
  }
  auto ev = [&](size_t blk, int which) { return ctx->panel_events[3 * blk + which]; };     // 0: potf2 done, 1: look-ahead rows solved, 2: bulk done
  if (split) {      // the bulk stream starts from the panel as `stream` has it now
    hipEvent_t e0 = ctx->panel_events[3 * nblk];
    LGP_HIP(hipEventRecord(e0, stream));
    LGP_HIP(hipStreamWaitEvent(bulk, e0, 0));
  }
  auto subst = [&](hipStream_t st, const T *w, size_t jb, T *X, size_t nrows) -> int {
    if (nrows == 0) return LGP_OK;
    hipLaunchKernelGGL((subst_right_mfma_kernel<T>), dim3((unsigned)((nrows + SUBSTM_ROWS - 1) / SUBSTM_ROWS)), dim3(256), 0, st, w, (int)jb, X,
                       (long)ldp, (long)nrows);
    LGP_LAUNCH_CHECK();
    return LGP_OK;
  };
  size_t bi = 0;
  hipEvent_t last_bulk = nullptr;
  for (size_t j0 = 0; j0 < nb; j0 += LGP_DB, ++bi) {
    const size_t jb = (nb - j0) < (size_t)LGP_DB ? (nb - j0) : (size_t)LGP_DB;
    T *Dg = P + j0 * ldp + j0;
    T *w = dblk + (j0 / LGP_DB) * (size_t)LGP_DBLK;
    LGP_CHECK(lgp_potf2_launch<T>(ctx, stream, Dg, ldp, (int)jb, w, row0 + j0));
    const size_t below = rows - (j0 + jb);
    if (below == 0) continue;
    T *X = P + (j0 + jb) * ldp + j0;           // below x jb, solved in place: X <- X * L_jj^-T
    T *C = P + (j0 + jb) * ldp + (j0 + jb);    // below x wc: the rest of the panel, C(0, 0) on the diagonal
    const size_t wc = nb - (j0 + jb);
    if (!split) {
      LGP_CHECK(subst(stream, w, jb, X, below));
      if (wc > 0) LGP_CHECK(lgp_gemm_launch<T>(ctx, stream, false, true, below, wc, jb, T(-1), X, ldp, X, ldp, T(1), C, ldp, true));
      continue;
    }
    const size_t jn = wc < (size_t)LGP_DB ? wc : (size_t)LGP_DB;      // rows of the next diagonal block (0: this is the last block)
    LGP_HIP(hipEventRecord(ev(bi, 0), stream));
    if (jn > 0) {
      // chain: the look-ahead rows and the next diagonal block
      if (bi > 0) LGP_HIP(hipStreamWaitEvent(stream, ev(bi - 1, 2), 0));      // they carry the updates of every earlier block
      LGP_CHECK(subst(stream, w, jb, X, jn));
      LGP_HIP(hipEventRecord(ev(bi, 1), stream));
      LGP_CHECK(lgp_gemm_launch<T>(ctx, stream, false, true, jn, jn, jb, T(-1), X, ldp, X, ldp, T(1), C, ldp, true));
    }
    // bulk: every other row below, then the rest of the panel
    LGP_HIP(hipStreamWaitEvent(bulk, ev(bi, 0), 0));
    LGP_CHECK(subst(bulk, w, jb, X + jn * ldp, below - jn));
    if (jn > 0 && below > jn) {
      LGP_HIP(hipStreamWaitEvent(bulk, ev(bi, 1), 0));
      T *Xr = X + jn * ldp, *Cr = C + jn * ldp;
      // rows [jn, below) x columns [0, jn): full rectangle;  x columns [jn, wc): lower part, diagonal at (jn, jn)
      LGP_CHECK(lgp_gemm_launch<T>(ctx, bulk, false, true, below - jn, jn, jb, T(-1), Xr, ldp, X, ldp, T(1), Cr, ldp, false));
      if (wc > jn) LGP_CHECK(lgp_gemm_launch<T>(ctx, bulk, false, true, below - jn, wc - jn, jb, T(-1), Xr, ldp, Xr, ldp, T(1), Cr + jn, ldp, true));
    }
    LGP_HIP(hipEventRecord(ev(bi, 2), bulk));
    last_bulk = ev(bi, 2);
  }
  if (last_bulk) LGP_HIP(hipStreamWaitEvent(stream, last_bulk, 0));      // the panel is complete on `stream` when the bulk stream is
  return LGP_OK;
}

template <typename T>
static int potrf_lookahead_v1(lgp_ctx *ctx, size_t n, T *L, size_t ldl, T *dblk, size_t nx) {
  LGP_REQUIRE(ldl >= n, "ldl < n");
  hipStream_t s0 = ctx->stream, s1 = ctx->stream2;
  const size_t NB = LGP_NB;
  ctx->potf2_small = n <= 8192;
  // Rows n .. n+nx-1 (optional) ride along: they are "rows below" for every panel, so the right-side substitutions and
  // the trailing updates turn B^T into (L^-1 B)^T -- the forward solve of the posterior done by the MFMA updates of
  // the factorisation instead of a separate latency chain of 2 n / 128 small kernels.
  // look-ahead: panel k+1 is updated and factored on s1 while s0 finishes the trailing update of step k
  for (size_t k0 = 0; k0 < n; k0 += NB) {
    const size_t kb = (n - k0) < NB ? (n - k0) : NB;
    T *P = L + k0 * ldl + k0;
    T *Yk = dblk + (k0 / LGP_DB) * (size_t)LGP_DBLK;
    if (k0 == 0) {
      LGP_CHECK(lgp_panel_factor_impl<T>(ctx, s0, n + nx - k0, kb, P, ldl, Yk, k0));
    }
    // (panel k0 is complete on s0 at this point: either factored above or joined from s1 below)
    const size_t t0 = k0 + kb;
    if (t0 >= n) break;
    const size_t rows_t = n - t0;
    const T *Pt = L + t0 * ldl + k0;            // (rows_t + nx) x kb: the panel rows below the diagonal block
    const size_t nb1 = rows_t < NB ? rows_t : NB;  // width of the next panel
    // s0: update the next panel's block column first (a full-GPU MFMA gemm of its own), then hand it to s1,
    // whose latency-bound factorisation chain (potf2 / substitution / K=128 updates) runs beside the rest of
    // the trailing update
    T *C1 = L + t0 * ldl + t0;                  // (rows_t + nx) x nb1 block column (lower part)
    LGP_CHECK(lgp_gemm_launch<T>(ctx, s0, false, true, rows_t + nx, nb1, kb, T(-1), Pt, ldl, Pt, ldl, T(1), C1, ldl, true));
    LGP_HIP(hipEventRecord(ctx->ev_a, s0));
    LGP_HIP(hipStreamWaitEvent(s1, ctx->ev_a, 0));
    LGP_CHECK(lgp_panel_factor_impl<T>(ctx, s1, rows_t + nx, nb1, C1, ldl, dblk + (t0 / LGP_DB) * (size_t)LGP_DBLK, t0));
    LGP_HIP(hipEventRecord(ctx->ev_b, s1));
    // s0: rest of the trailing update (everything right of the next panel): the N^3/3 flops.  One launch
    // (splitting it into block-column chunks to give s1 more scheduling points was measured 30 % slower).
    if (rows_t > nb1) {
      const size_t rows_r = rows_t - nb1;
      const T *Pa = L + (t0 + nb1) * ldl + k0;          // (rows_r + nx) x kb
      T *Cc = L + (t0 + nb1) * ldl + (t0 + nb1);
      // algorithmic flops of this launch: 2 k per produced entry of the lower trapezoid
      const double entries = 0.5 * (double)rows_r * ((double)rows_r + 1.0) + (double)nx * (double)rows_r;
      // (timed for bench.py when it is the 128-tile instantiation that profilers list as gemm_kernel<..., 128, 256, 1>;
      // the updates of up to lgp_gemm_t64_trailing_max() tiles go to the 64-tile kernel and are left out of both)
      const bool timed = ((rows_r + nx + 127) / 128) * ((rows_r + 127) / 128) > lgp_gemm_t64_trailing_max();   // lgp_gemm_launch's 128-tile criterion
      if (timed) LGP_CHECK(lgp_prof_start(ctx, s0, 2.0 * (double)kb * entries));
      LGP_CHECK(lgp_gemm_launch<T>(ctx, s0, false, true, rows_r + nx, rows_r, kb, T(-1), Pa, ldl, Pa, ldl, T(1), Cc, ldl, true, 3));
      if (timed) LGP_CHECK(lgp_prof_stop(ctx, s0));
    }
    LGP_HIP(hipStreamWaitEvent(s0, ctx->ev_b, 0));
  }
  return LGP_OK;
}

// The driver.  Three streams:
//   s1 (chain, high priority)  per panel k: U1(k-1) = the K = 1024 update of the panel's own diagonal SQUARE, then the square
//                              factored block by block (potf2 -> substitution of the square's rows below -> K = 128 update of
//                              the rest of the square): everything the next diagonal block waits for, and nothing else;
//   s3 (far rows)              the rows below the square (and the carried right-hand sides) follow one block behind: for each
//                              block an event of s1 ("rows of the square solved"), their substitution and their K = 128 update.
//                              s3 waits for s1, NEVER the other way round inside a panel, so no event latency lands on the chain;
//   s0 (main)                  U2(k) = rest of block column k+1, U3(k) = everything right of it (the N^3/3 flops).
// Against the round-1 driver (whole panel, all rows, on the chain stream; kept as potrf_lookahead_v1, LGP_POTRF=v1) the chain
// loses the substitution and the update of up to N rows per block and the part of the block-column update below the square.
static int potrf_chain(lgp_ctx *ctx, size_t n, double *L, size_t ldl, double *dblk, size_t nx, size_t row0, bool early);

template <typename T>
static int potrf_split(lgp_ctx *ctx, size_t n, T *L, size_t ldl, T *dblk, size_t nx) {
  ThinGuard thin_guard(ctx);
  // The tail of a large factorisation is bound by the chain of diagonal blocks, not by the updates: once this few rows are left,
  // the trailing matrix (already carrying every update) is handed to the chain-kernel driver, which factors it faster than the
  // launch-per-block chain here.  Measured (tools/tail_sweep.sh, Cholesky alone): tail 8192 -> N = 32768 181.7 -> 180.9 ms, N = 20480
  // 51.8 -> 50.5 ms; tail 4096: -0.2 / -0.8 ms; 12288: +0.2 / -0.6; 16384: +1.6 / +0.7 ms (the chain kernel keeps 16 CUs from updates that
  // still bound the factorisation there).  Decided before anything is enqueued: the first use of the chain kernel in a context runs a
  // handshake between its streams.
  static const size_t tail_env = getenv("LGP_SPLIT_CHAIN_TAIL") ? (size_t)atol(getenv("LGP_SPLIT_CHAIN_TAIL")) : (size_t)8192;
  size_t chain_tail = 0;
  if constexpr (std::is_same<T, double>::value) {
    if (tail_env >= 2048 && n > tail_env && n % LGP_DB == 0) {
      const size_t cand = (tail_env < ctx->chain_max_n ? tail_env : ctx->chain_max_n) / LGP_NB * LGP_NB;
      if (cand >= 2048 && lgp_chain_usable(ctx, cand, ldl)) chain_tail = cand;
    }
  }
  // chain-bound sizes: the trailing updates (sg) and the far rows (s3) run on CU-masked streams that leave a few CUs to the chain
  const bool reserve = ctx->stream_gm != nullptr && ctx->stream_fm != nullptr && n <= ctx->reserve_max_n;
  hipStream_t s0 = ctx->stream, s1 = ctx->stream2, s3 = reserve ? ctx->stream_fm : ctx->stream3, sg = reserve ? ctx->stream_gm : s0;
  const size_t NB = LGP_NB;
  enum { EV_BLK = 0, EV_SQ = 8, EV_FAR = 9, EV_U2 = 10, EV_U3 = 11, EV_START = 12, EV_COUNT = 13 };
  while (ctx->panel_events.size() < (size_t)EV_COUNT) {
    hipEvent_t e;
    LGP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->panel_events.push_back(e);
  }
  auto E = [&](int i) { return ctx->panel_events[(size_t)i]; };
  auto subst = [&](hipStream_t st, const T *w, size_t jb, T *X, size_t nrows) -> int {
    if (nrows == 0) return LGP_OK;
    hipLaunchKernelGGL((subst_right_mfma_kernel<T>), dim3((unsigned)((nrows + SUBSTM_ROWS - 1) / SUBSTM_ROWS)), dim3(256), 0, st, w, (int)jb, X,
                       (long)ldl, (long)nrows);
    LGP_LAUNCH_CHECK();
    return LGP_OK;
  };
  static const bool right_looking = getenv("LGP_FAR_RIGHT") != nullptr;
  LGP_HIP(hipEventRecord(E(EV_START), s0));             // the matrix as the caller's stream has it
  LGP_HIP(hipStreamWaitEvent(s1, E(EV_START), 0));
  LGP_HIP(hipStreamWaitEvent(s3, E(EV_START), 0));
  if (sg != s0) LGP_HIP(hipStreamWaitEvent(sg, E(EV_START), 0));
  // Round 5: the look-ahead's products (the square's, the rows below it, U1) run at wave priority 3 like the substitutions and the
  // potf2 beside them: they finish sooner, and the large update they share the CUs with loses less to them -- C3 on one box
  // 191.4-192.2 against 193.0-193.1 ms, the large updates at 0.805-0.809 of peak against 0.785-0.787.  LGP_SPLIT_PRIO=0: off.
  static const bool split_prio = !(getenv("LGP_SPLIT_PRIO") != nullptr && atoi(getenv("LGP_SPLIT_PRIO")) == 0);
  bool have_u3 = false, have_far = false;
  for (size_t k0 = 0; k0 < n; k0 += NB) {
    if constexpr (std::is_same<T, double>::value) {
      if (chain_tail && k0 > 0 && n - k0 <= chain_tail && !reserve) {
        // the trailing matrix has every update once U1 (s1) and U2 / U3 (s0, behind the far rows) of the previous panel are done
        // (the square of this panel needs U1 alone, which is in s1's order: the chain kernel starts beside U3)
        static const bool late = getenv("LGP_SPLIT_CHAIN_LATE") != nullptr;
        if (late) {
          LGP_HIP(hipEventRecord(E(EV_SQ), s1));
          LGP_HIP(hipStreamWaitEvent(s0, E(EV_SQ), 0));
        }
        return potrf_chain(ctx, n - k0, L + k0 * ldl + k0, ldl, dblk + (k0 / LGP_DB) * (size_t)LGP_DBLK, nx, k0, !late);
      }
    }
    const size_t kb = (n - k0) < NB ? (n - k0) : NB;
    T *P = L + k0 * ldl + k0;
    T *Yk = dblk + (k0 / LGP_DB) * (size_t)LGP_DBLK;
    const size_t far = n + nx - (k0 + kb);              // rows below the square, carried rows included
    ctx->potf2_late = ((far + 127) / 128) * ((n - k0 - kb + 127) / 128) <= lgp_gemm_t64_trailing_max() ? 1 : 0;   // (this square runs beside a 64-tile update)
    T *Pfar = P + kb * ldl;
    // ---- the square on s1, the far rows one block behind on s3
    int bi = 0;
    for (size_t j0 = 0; j0 < kb; j0 += LGP_DB, ++bi) {
      const size_t jb = (kb - j0) < (size_t)LGP_DB ? (kb - j0) : (size_t)LGP_DB;
      T *w = Yk + (j0 / LGP_DB) * (size_t)LGP_DBLK;
      LGP_CHECK(lgp_potf2_launch<T>(ctx, s1, P + j0 * ldl + j0, ldl, (int)jb, w, k0 + j0));
      const size_t below = kb - (j0 + jb), wc = below;  // rows of the square below the block = columns right of it
      T *Xsq = P + (j0 + jb) * ldl + j0;
      LGP_CHECK(subst(s1, w, jb, Xsq, below));
      if (far > 0) {
        LGP_HIP(hipEventRecord(E(EV_BLK + bi), s1));
        LGP_HIP(hipStreamWaitEvent(s3, E(EV_BLK + bi), 0));
        // left-looking: column block j of the far rows receives the update of all earlier blocks in one product (K = j0: half the
        // memory traffic of eight K = 128 updates of all the remaining columns), then is solved
        // (LGP_FAR_RECURSIVE: 0 = left-looking K = 128 j products of 128 columns; 1 = recursive order, the default: C3 189.9 -> 189.3 ms,
        //  N = 20480 50.7 -> 50.3 ms; 2 = the same on 128-tiles where they fill the GPU: equal)
        static const int far_rec_env = getenv("LGP_FAR_RECURSIVE") ? atoi(getenv("LGP_FAR_RECURSIVE")) : 1;
        const int far_rec = kb % LGP_DB == 0 ? far_rec_env : 0;          // (a ragged last panel keeps the left-looking order)
        ctx->gemm_prio = split_prio ? 3 : 0;
        if (j0 > 0 && !right_looking && !far_rec)
          LGP_CHECK(lgp_gemm_launch<T>(ctx, s3, false, true, far, jb, j0, T(-1), Pfar, ldl, P + j0 * ldl, ldl, T(1), Pfar + j0, ldl, false, 2));
        LGP_CHECK(subst(s3, w, jb, Pfar + j0, far));
        if (wc > 0 && right_looking)
          LGP_CHECK(lgp_gemm_launch<T>(ctx, s3, false, true, far, wc, jb, T(-1), Pfar + j0, ldl, Xsq, ldl, T(1), Pfar + (j0 + jb), ldl, false));
        if (far_rec && !right_looking) {
          // recursive order: after block b the columns [c, c + s) get blocks [c - s, c), c = b + 1, s = its lowest set bit -- the same
          // flops as the left-looking K = 128 j products of 128 columns, 16 of 28 units of them in ONE 512 x 512 product
          const size_t c = (size_t)bi + 1, sbit = c & (~c + 1), nq = kb / LGP_DB;
          if (c < nq) {
            const size_t ncol = (c + sbit <= nq ? sbit : nq - c) * LGP_DB, kk = sbit * LGP_DB;
            LGP_CHECK(lgp_gemm_launch<T>(ctx, s3, false, true, far, ncol, kk, T(-1), Pfar + (c - sbit) * LGP_DB, ldl, P + (c * LGP_DB) * ldl + (c - sbit) * LGP_DB, ldl,
                                         T(1), Pfar + c * LGP_DB, ldl, false, far_rec == 1 ? 2 : 0));
          }
        }
      }
      if (wc > 0) LGP_CHECK(lgp_gemm_launch<T>(ctx, s1, false, true, below, wc, jb, T(-1), Xsq, ldl, Xsq, ldl, T(1), P + (j0 + jb) * ldl + (j0 + jb), ldl, true));
      ctx->gemm_prio = 0;
    }
    LGP_HIP(hipEventRecord(E(EV_SQ), s1));
    have_far = far > 0;
    if (have_far) LGP_HIP(hipEventRecord(E(EV_FAR), s3));
    const size_t t0 = k0 + kb;
    if (t0 >= n) break;
    const size_t rows_t = n - t0;
    const T *Pt = L + t0 * ldl + k0;                    // (rows_t + nx) x kb: the panel below its square
    const size_t nb1 = rows_t < NB ? rows_t : NB;       // width of the next panel
    T *C1 = L + t0 * ldl + t0;
    // ---- s1: U1, the next panel's square (it has every earlier update once U3 of the previous step is done)
    LGP_HIP(hipStreamWaitEvent(s1, E(EV_FAR), 0));
    if (have_u3) LGP_HIP(hipStreamWaitEvent(s1, E(EV_U3), 0));
    ctx->gemm_prio = split_prio ? 3 : 0;
    const int u1st = lgp_gemm_launch<T>(ctx, s1, false, true, nb1, nb1, kb, T(-1), Pt, ldl, Pt, ldl, T(1), C1, ldl, true);
    ctx->gemm_prio = 0;
    LGP_CHECK(u1st);
    // LGP_LOOKAHEAD_DELAY=<fraction> (experiment): the next square -- and with it the rows below it -- starts only that fraction of
    // the large update's estimated duration (its flops at 65 TFLOP/s) later: one wave that sleeps on the chain stream.  A look-ahead
    // that is held back interferes less with the large update (potf2_dev.h); held back too long it makes the update's stream wait.
    static const double la_delay = getenv("LGP_LOOKAHEAD_DELAY") ? atof(getenv("LGP_LOOKAHEAD_DELAY")) : 0.0;
    if (la_delay > 0 && rows_t > nb1) {
      const double rr = (double)(rows_t - nb1);
      const double us = la_delay * (2.0 * (double)kb * (0.5 * rr * rr + (double)nx * rr)) / 65e12 * 1e6;
      if (us >= 50.) {
        hipLaunchKernelGGL(lookahead_delay_kernel, dim3(1), dim3(64), 0, s1, (unsigned long long)(us * 100.));      // (100 MHz clock)
        LGP_LAUNCH_CHECK();
      }
    }
    // ---- sg: U2, the block column below that square; then s3 may touch it
    LGP_HIP(hipStreamWaitEvent(sg, E(EV_SQ), 0));
    LGP_HIP(hipStreamWaitEvent(sg, E(EV_FAR), 0));
    const size_t rows_r = rows_t - nb1;
    // (the last panels are bound by the chain: their updates leave room on every CU, see potrf_chain)
    static const long thin_env = getenv("LGP_THIN_LDS_KB") ? atol(getenv("LGP_THIN_LDS_KB")) : -1;
    static const size_t thin_small = (size_t)(getenv("LGP_THIN_SMALL_KB") ? atol(getenv("LGP_THIN_SMALL_KB")) : 20) * 1024;      // (70 until round 5)
    ctx->gemm_thin_lds = thin_env >= 0 ? (size_t)thin_env * 1024 : (rows_t <= 6144 ? thin_small : (rows_t <= 12288 ? (size_t)20 * 1024 : 0));
    static const size_t u2_t128_min = getenv("LGP_U2_T128_MIN_ROWS") ? (size_t)atol(getenv("LGP_U2_T128_MIN_ROWS")) : (size_t)1 << 40;
    if (rows_r + nx > 0)
      LGP_CHECK(lgp_gemm_launch<T>(ctx, sg, false, true, rows_r + nx, nb1, kb, T(-1), Pt + nb1 * ldl, ldl, Pt, ldl, T(1), C1 + nb1 * ldl, ldl, false,
                                   rows_r >= u2_t128_min ? 0 : 2));
    LGP_HIP(hipEventRecord(E(EV_U2), sg));
    LGP_HIP(hipStreamWaitEvent(s3, E(EV_U2), 0));
    // ---- sg: U3, everything right of the next panel: the N^3/3 flops, one launch
    if (rows_r > 0) {
      const T *Pa = Pt + nb1 * ldl;                     // (rows_r + nx) x kb
      T *Cc = L + (t0 + nb1) * ldl + (t0 + nb1);
      // algorithmic flops of this launch: 2 k per produced entry of the lower trapezoid
      const double entries = 0.5 * (double)rows_r * ((double)rows_r + 1.0) + (double)nx * (double)rows_r;
      // (timed for bench.py when it is the 128-tile instantiation that profilers list as gemm_kernel<..., 128, 256, 1>;
      // the updates of up to lgp_gemm_t64_trailing_max() tiles go to the 64-tile kernel and are left out of both)
      const bool timed = ((rows_r + nx + 127) / 128) * ((rows_r + 127) / 128) > lgp_gemm_t64_trailing_max();   // lgp_gemm_launch's 128-tile criterion
      if (timed) LGP_CHECK(lgp_prof_start(ctx, sg, 2.0 * (double)kb * entries));
      LGP_CHECK(lgp_gemm_launch<T>(ctx, sg, false, true, rows_r + nx, rows_r, kb, T(-1), Pa, ldl, Pa, ldl, T(1), Cc, ldl, true, 3));
      if (timed) LGP_CHECK(lgp_prof_stop(ctx, sg));
    }
    ctx->gemm_thin_lds = 0;
    LGP_HIP(hipEventRecord(E(EV_U3), sg));
    have_u3 = true;
  }
  // the factor is complete on the caller's stream when the chain and the far rows are
  LGP_HIP(hipStreamWaitEvent(s0, E(EV_SQ), 0));
  if (have_far) LGP_HIP(hipStreamWaitEvent(s0, E(EV_FAR), 0));
  if (sg != s0 && have_u3) LGP_HIP(hipStreamWaitEvent(s0, E(EV_U3), 0));
  return LGP_OK;
}

// The driver for fp64 matrices whose order is a multiple of 128: the chain kernel (potf2m.hip) factors the diagonal squares of
// all panels on CUs of its own, from one launch on s1; the rest is enqueued around it, ordered by flag words that the command
// processor polls (hipStreamWaitValue32) or writes (hipStreamWriteValue32) -- no kernel of the dependent chain is ever
// launched, waited for through an event, or run beside a trailing update.  The rows below a panel's square are cut into row
// groups with a stream each: group 0 = the rows of the NEXT diagonal square (the only ones the chain needs back), the others
// share the rest.  Per panel k:
//   group stream g   wait u2[k-1];  per block j:  wait ready[k][j]  ->  substitution and K = 128 update of the group's rows;  grp[k][g] := 1
//   group 0 then     wait u3[k-1],  U1(k) = the K = 1024 update of the next square,  u1[k+1] := 1  (the chain kernel goes on)
//   s0 (main)        wait grp[k][*]  ->  U2(k) (block column k+1 below its square),  u2[k] := 1,  U3(k) (everything right of it),  u3[k] := 1
// One stream for all the rows below was measured first: its 16 dependent launches per panel (8 x (37 + 52) us at N = 8192) were
// slower than the chain (8 x 72 us) and the next square waited ~400 us for them; per group they overlap.
// early (the hand-over from potrf_split): s1 already holds, in its own order, everything the FIRST square needs -- the chain kernel
// is launched there without waiting for the caller's stream, i.e. beside the trailing update that is still running on it
static int potrf_chain(lgp_ctx *ctx, size_t n, double *L, size_t ldl, double *dblk, size_t nx, size_t row0, bool early) {
  typedef double T;
  ThinGuard thin_guard(ctx);
  hipStream_t s0 = ctx->stream, s1 = ctx->stream2;
  const size_t NB = LGP_NB;
  // row groups (flag words LGP_CH_GRP .. + 3).  Two, so that the factorisation runs on four streams in all: the runtime
  // maps streams onto four hardware queues (GPU_MAX_HW_QUEUES), and with more streams than that a stream can end up behind
  // another stream's polling wait in the same queue (measured with four groups: 2 ms per panel hand-over instead of 0.1)
  const int MAXG = ctx->chain_groups;
  // (the far-row streams were created, and each paired with s1, by the self-test behind lgp_chain_usable)
  LGP_REQUIRE(ctx->far_streams.size() >= (size_t)MAXG, "far-row streams missing: lgp_chain_usable not called");
  enum { EV_SQ = 8, EV_START = 12, EV_GRP = 13, EV_COUNT = 13 + 4 };
  while (ctx->panel_events.size() < (size_t)EV_COUNT) {
    hipEvent_t e;
    LGP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->panel_events.push_back(e);
  }
  auto E = [&](int i) { return ctx->panel_events[(size_t)i]; };
  unsigned *st = nullptr;
  LGP_CHECK(lgp_chain_state(ctx, n, &st));
  auto word = [&](size_t k, int which) { return st + LGP_CH_HEADER + LGP_CH_PANEL * k + which; };
  auto wait_ge = [&](hipStream_t s, unsigned *p, unsigned v) { return hipStreamWaitValue32(s, p, v, hipStreamWaitValueGte, 0xffffffffu); };
  auto subst = [&](hipStream_t s, const T *w, T *X, size_t nrows) -> int {
    if (nrows == 0) return LGP_OK;
    hipLaunchKernelGGL((subst_right_mfma_kernel<T>), dim3((unsigned)((nrows + SUBSTM_ROWS - 1) / SUBSTM_ROWS)), dim3(256), 0, s, w, (int)LGP_DB, X,
                       (long)ldl, (long)nrows);
    LGP_LAUNCH_CHECK();
    return LGP_OK;
  };
  // the flag words are zeroed before any stream can look at them: every stream starts behind this point of s0
  if (early) {
    LGP_HIP(hipMemsetAsync(st, 0, lgp_chain_state_bytes(n), s1));
    LGP_HIP(hipEventRecord(E(EV_START), s1));
    LGP_HIP(hipStreamWaitEvent(s0, E(EV_START), 0));
    LGP_HIP(hipEventRecord(E(EV_START), s0));
  } else {
    LGP_HIP(hipMemsetAsync(st, 0, lgp_chain_state_bytes(n), s0));
    LGP_HIP(hipEventRecord(E(EV_START), s0));
    LGP_HIP(hipStreamWaitEvent(s1, E(EV_START), 0));
  }
  for (int g = 0; g < MAXG; ++g) LGP_HIP(hipStreamWaitEvent(ctx->far_streams[(size_t)g], E(EV_START), 0));
  LGP_CHECK(lgp_chain_launch(ctx, s1, n, L, ldl, dblk, row0));
  LGP_HIP(hipEventRecord(E(EV_SQ), s1));
  const size_t npanels = (n + NB - 1) / NB;
  for (size_t k = 0; k < npanels; ++k) {
    const size_t k0 = k * NB, kb = (n - k0) < NB ? (n - k0) : NB, nq = kb / LGP_DB;
    T *P = L + k0 * ldl + k0;
    T *Yk = dblk + (k0 / LGP_DB) * (size_t)LGP_DBLK;
    const size_t t0 = k0 + kb, rows_t = n - t0;         // matrix rows below the square
    const size_t below = rows_t + nx;                   // ... and the carried rows
    const size_t nb1 = rows_t < NB ? rows_t : NB;       // width of the next panel = rows of group 0
    // row groups: [g0[g], g0[g+1]) relative to the first row below the square
    size_t g0[4 + 1];
    int ng = 0;
    g0[0] = 0;
    if (below > 0) {
      // equal shares (multiples of 64 rows), at least 1024 rows each: the groups advance side by side, one block behind the chain
      size_t parts = below / 1024;
      parts = parts < 1 ? 1 : (parts > (size_t)MAXG ? (size_t)MAXG : parts);
      const size_t per = ((below + parts - 1) / parts + 63) / 64 * 64;
      while (g0[ng] < below) {
        const size_t end = g0[ng] + per < below ? g0[ng] + per : below;
        g0[++ng] = end;
      }
    }
    for (int g = 0; g < ng; ++g) {
      hipStream_t sg = ctx->far_streams[(size_t)g];
      const size_t rows_g = g0[g + 1] - g0[g];
      T *Pg = P + (kb + g0[g]) * ldl;                   // the group's rows of the panel
      if (k >= 1) LGP_HIP(wait_ge(sg, word(k - 1, LGP_CH_U2), 1));     // its rows of this block column have panel k-1's update
      // LEFT-looking over the blocks of the square: column block j of the group's rows first receives the update of all
      // earlier blocks in one product (K = 128 j; it needs the square's rows of block j final = every substitution of step
      // j - 1 done, and runs WHILE the chain kernel factors block j), then is solved as soon as block j is factored.  After the
      // last block only a substitution stands between the chain kernel and U1: 75-180 us from the end of a square to the start
      // of the next in the late panels of N = 8192, where right-looking rows (every block followed by a K = 128 update of all
      // the remaining columns: 37 + 52 us per step against the chain's 68) fell 20 us further behind per block (~390 us).
      // (The early panels wait for the previous panel's updates instead: their far rows cannot start before U2.)
      // (Round 5 tried ONE resident launch per panel whose waves poll those words themselves, docs/history/far_resident: never
      // faster -- N = 8192 6.39 against 6.11-6.25 ms, 12288 16.0 against 14.9 -- because what bounds these sizes is the update stream,
      // busy without a gap, and the resident workgroups take CUs and L2 bandwidth from it for as long as they live.)
      for (size_t j = 0; j < nq; ++j) {
        const size_t j0 = j * LGP_DB;
        if (j >= 1) {
          LGP_HIP(wait_ge(sg, word(k, LGP_CH_READY) + (j - 1), (unsigned)(1 + 2 * nq - 2 * (j - 1) - 2)));
          // (round 5: the chain-side products at wave priority 3, like the substitutions and the potf2: N = 8192 6.12-6.19 vs 6.18-6.33 ms,
          //  12288 15.05 vs 15.15-15.20 on one box; LGP_FAR_PRIO=0 switches it off)
          static const bool far_prio = !(getenv("LGP_FAR_PRIO") != nullptr && atoi(getenv("LGP_FAR_PRIO")) == 0);
          ctx->gemm_prio = far_prio ? 3 : 0;
          const int gst = lgp_gemm_launch<T>(ctx, sg, false, true, rows_g, LGP_DB, j0, T(-1), Pg, ldl, P + j0 * ldl, ldl, T(1), Pg + j0, ldl, false, 2);
          ctx->gemm_prio = 0;
          LGP_CHECK(gst);
        }
        LGP_HIP(wait_ge(sg, word(k, LGP_CH_FLAGP) + j, 1));
        LGP_CHECK(subst(sg, Yk + j * (size_t)LGP_DBLK, Pg + j0, rows_g));
      }
      LGP_HIP(hipStreamWriteValue32(sg, word(k, LGP_CH_GRP + g), 1, 0));
    }
    if (t0 >= n) {
      if (ng == 0) {      // nothing below the last square: the caller's stream still has to see it finished
        LGP_HIP(wait_ge(ctx->far_streams[0], word(k, LGP_CH_READY) + (nq - 1), 1));
      }
      break;
    }
    const T *Pt = L + t0 * ldl + k0;                    // (rows_t + nx) x kb: the panel below its square
    T *C1 = L + t0 * ldl + t0;
    // ---- group 0's stream: U1, the next panel's square; it has every earlier update once U3 of the previous step is done
    {
      hipStream_t sg = ctx->far_streams[0];
      for (int g = 1; g < ng; ++g) LGP_HIP(wait_ge(sg, word(k, LGP_CH_GRP + g), 1));      // (the rows of the next square may sit in any group)
      if (k >= 1) LGP_HIP(wait_ge(sg, word(k - 1, LGP_CH_U3), 1));
      static const bool u1_prio = !(getenv("LGP_FAR_PRIO") != nullptr && atoi(getenv("LGP_FAR_PRIO")) == 0);
      ctx->gemm_prio = u1_prio ? 3 : 0;
      const int ust = lgp_gemm_launch<T>(ctx, sg, false, true, nb1, nb1, kb, T(-1), Pt, ldl, Pt, ldl, T(1), C1, ldl, true);
      ctx->gemm_prio = 0;
      LGP_CHECK(ust);
      if (!(ctx->chain_test_stall && k == 0)) LGP_HIP(hipStreamWriteValue32(sg, word(k + 1, LGP_CH_U1), 1, 0));
    }
    // ---- s0: U2, the block column below that square
    for (int g = 0; g < ng; ++g) LGP_HIP(wait_ge(s0, word(k, LGP_CH_GRP + g), 1));
    const size_t rows_r = rows_t - nb1;
    // The updates with this panel leave room on every CU when the chain is what bounds the rest of the factorisation: untouched
    // dynamic LDS thins their workgroups out (two 64-tile workgroups per CU below 12288 remaining rows: 36.9 + 20 KB each of 160), so that the far-row
    // kernels of the next panel are dispatched at once and not squeezed (N = 4096 2.58 -> 2.48 ms, N = 8192 6.92 -> 6.55 ms;
    // thinning the large updates of N = 16384 as well: 30.4 -> 31.4 ms, so they are left alone).  Until round 5 it was ONE
    // workgroup per CU below 6144 rows; since the chain-side products run at wave priority 3 that costs the updates -- whose
    // stream is busy without a gap at N = 8192 -- more than it gives the rows: N = 8192 5.97-6.00 against 6.11-6.25 ms, 12288
    // 14.52 against 14.85 on one box (LGP_THIN_SMALL_KB=70 is the old setting).
    static const long thin_env = getenv("LGP_THIN_LDS_KB") ? atol(getenv("LGP_THIN_LDS_KB")) : -1;
    static const size_t thin_small = (size_t)(getenv("LGP_THIN_SMALL_KB") ? atol(getenv("LGP_THIN_SMALL_KB")) : 20) * 1024;      // (70 until round 5)
    ctx->gemm_thin_lds = thin_env >= 0 ? (size_t)thin_env * 1024 : (rows_t <= 6144 ? thin_small : (rows_t <= 12288 ? (size_t)20 * 1024 : 0));
    // (U2 in blocks of 128 columns with a count in u2[k] behind each, so that step j of the rows below the next square waits for
    // block j only: measured worse, N = 8192 6.35 -> 7.2 ms -- eight 128-column products take 440 us where the one launch takes 290)
    // small matrices: U2 behind U1 (side by side U1 takes 130 us instead of 45 and the next square waits for it; N = 4096 2.44 ->
    // 2.37 ms.  From N = 8192 on it is the rows below the next square, which wait for U2, that must not start later: 6.33 -> 6.47 ms)
    static const size_t u2_after_u1_max_n = getenv("LGP_CHAIN_U2_AFTER_U1_MAX_N") ? (size_t)atol(getenv("LGP_CHAIN_U2_AFTER_U1_MAX_N")) : 4096;
    if (n <= u2_after_u1_max_n && !(ctx->chain_test_stall && k == 0)) LGP_HIP(wait_ge(s0, word(k + 1, LGP_CH_U1), 1));
    if (rows_r + nx > 0)
      LGP_CHECK(lgp_gemm_launch<T>(ctx, s0, false, true, rows_r + nx, nb1, kb, T(-1), Pt + nb1 * ldl, ldl, Pt, ldl, T(1), C1 + nb1 * ldl, ldl, false, 2));
    LGP_HIP(hipStreamWriteValue32(s0, word(k, LGP_CH_U2), 1, 0));
    // ---- s0: the bulk of the flops.  Up to N = 12288 (8192 until round 5) LEFT-looking with a look-ahead of one block column: block column k+2 (its
    // square and everything below) receives the update of ALL panels 0 .. k in one product, K = 1024 (k + 1), while the chain
    // kernel factors the square of panel k+1 (whose block column got the same treatment one panel earlier, and panel k's share
    // from U1 / U2).  Right-looking (everything right of block column k+1 updated with panel k alone) front-loads the work: at
    // N = 8192 the updates with panels 0 / 1 / 2 are 51 / 37 / 26 GFLOP, more than fits beside a square's 0.55 ms, and the
    // square of panel k+2 waits for all of update k (hand-overs of 630 / 530 / 330 us in the task trace) while the last panels
    // leave the GPU idle; left-looking the work beside each square is 33-39 GFLOP for panels 0-3: N = 8192 6.45-6.55 -> 6.30 ms
    // (N = 4096 equal; N = 12288 14.9 -> 16.0, N = 16384 29.8 -> 34.6 ms: there the updates bound the factorisation, and a block
    // column of 1024 is too few tiles per launch -- right-looking stays).  LGP_CHAIN_LEFT_MAX_N.
    // (round 5, with the updates thinned less and 24 workers: left-looking now wins up to N = 12288 -- same box, left against right:
    //  9216 7.28-7.39 / 7.69-7.72 ms, 10240 9.15-9.20 / 9.54-9.62, 11264 11.5-11.7 / 11.9-12.2, 12288 14.31-14.33 / 14.47-14.49;
    //  13312 17.8 / 17.4, 14336 21.8 / 21.0: right-looking from there on)
    static const size_t left_max_n = getenv("LGP_CHAIN_LEFT_MAX_N") ? (size_t)atol(getenv("LGP_CHAIN_LEFT_MAX_N")) : 12288;
    const bool right_looking = n > left_max_n;
    // The only thing that waits for this update through its flag word is U1 of the NEXT panel, i.e. the square of block column k+2
    // (everything else follows in s0's own order).  LGP_CHAIN_U3_SPLIT=1 (round 5, off): that square first, as a launch of its own
    // with the flag right behind it, so that the next square never waits for the bulk of an update.  Measured on one box: N = 8192
    // 6.33-6.44 against 6.18-6.30 ms, N = 12288 16.1 against 14.9 (the rest of a right-looking update as a row slab loses its
    // triangular tile map), N = 16384 equal.  The kernel trace says why it cannot help: in the middle panels of N = 8192 it is the
    // rows below the square (16 dependent launches per panel, 2-3 times slower beside the update than alone: 733 us against the
    // square's 530) that the next square waits for, not this flag (profiles/r05_chain_timeline_8192.txt).
    static const bool u3_split = getenv("LGP_CHAIN_U3_SPLIT") != nullptr && atoi(getenv("LGP_CHAIN_U3_SPLIT")) != 0;
    const size_t nb2 = rows_r < NB ? rows_r : NB;       // width of panel k+2
    bool flagged = false;
    if (rows_r > 0 && right_looking) {
      const T *Pa = Pt + nb1 * ldl;                     // (rows_r + nx) x kb
      T *Cc = L + (t0 + nb1) * ldl + (t0 + nb1);
      size_t top = 0;
      if (u3_split && rows_r + nx > nb2) {
        LGP_CHECK(lgp_gemm_launch<T>(ctx, s0, false, true, nb2, nb2, kb, T(-1), Pa, ldl, Pa, ldl, T(1), Cc, ldl, true, 2));
        LGP_HIP(hipStreamWriteValue32(s0, word(k, LGP_CH_U3), 1, 0));
        flagged = true;
        top = nb2;
      }
      const size_t mrows = rows_r + nx - top;
      const double entries = 0.5 * (double)rows_r * ((double)rows_r + 1.0) + (double)nx * (double)rows_r - 0.5 * (double)top * ((double)top + 1.0);
      // (timed for bench.py when it is the 128-tile instantiation that profilers list as gemm_kernel<..., 128, 256, 1>)
      const bool timed = ((mrows + 127) / 128) * ((rows_r + 127) / 128) > lgp_gemm_t64_trailing_max();   // lgp_gemm_launch's 128-tile criterion
      if (timed) LGP_CHECK(lgp_prof_start(ctx, s0, 2.0 * (double)kb * entries));
      LGP_CHECK(lgp_gemm_launch<T>(ctx, s0, false, true, mrows, rows_r, kb, T(-1), Pa + top * ldl, ldl, Pa, ldl, T(1), Cc + top * ldl, ldl, true, 3, 0, 0, (int)top));
      if (timed) LGP_CHECK(lgp_prof_stop(ctx, s0));
    } else if (rows_r > 0) {
      const size_t t2 = t0 + nb1;
      const T *Pa = L + t2 * ldl;                       // (rows_r + nx) x t0: every panel so far, the rows from square k+2 down
      T *Cc = L + t2 * ldl + t2;
      if (u3_split && rows_r + nx > nb2) {
        LGP_CHECK(lgp_gemm_launch<T>(ctx, s0, false, true, nb2, nb2, t0, T(-1), Pa, ldl, Pa, ldl, T(1), Cc, ldl, true, 6));
        LGP_HIP(hipStreamWriteValue32(s0, word(k, LGP_CH_U3), 1, 0));
        flagged = true;
        LGP_CHECK(lgp_gemm_launch<T>(ctx, s0, false, true, rows_r + nx - nb2, nb2, t0, T(-1), Pa + nb2 * ldl, ldl, Pa, ldl, T(1), Cc + nb2 * ldl, ldl, false, 6));
      } else {
        LGP_CHECK(lgp_gemm_launch<T>(ctx, s0, false, true, rows_r + nx, nb2, t0, T(-1), Pa, ldl, Pa, ldl, T(1), Cc, ldl, true, 6));
      }
    }
    ctx->gemm_thin_lds = 0;
    if (!flagged) LGP_HIP(hipStreamWriteValue32(s0, word(k, LGP_CH_U3), 1, 0));
  }
  // the factor is complete on the caller's stream when the chain kernel has left and every row group is done
  LGP_HIP(hipStreamWaitEvent(s0, E(EV_SQ), 0));
  for (int g = 0; g < MAXG; ++g) {
    LGP_HIP(hipEventRecord(E(EV_GRP + g), ctx->far_streams[(size_t)g]));
    LGP_HIP(hipStreamWaitEvent(s0, E(EV_GRP + g), 0));
  }
  return LGP_OK;
}

// The end of the sharded factorisation when its rank has no peers: the trailing matrix (every update applied; U1 of its first
// square in s1's order, U2 / U3 of the panel before on the context's stream) handed to the chain driver, as potrf_split does.
int lgp_potrf_chain_tail_f64(lgp_ctx *ctx, size_t n, double *L, size_t ldl, double *dblk, size_t nx, size_t row0) {
  return potrf_chain(ctx, n, L, ldl, dblk, nx, row0, true);
}

// ---- pieces of a panel for the sharded factorisation (dist.hip), the same kernels as the drivers above -----------------------
// rows below a panel's diagonal square, solved against it: left-looking over the square's 128-column blocks, as the far rows
// of potrf_split.  Pfar: the rows (column 0 of the panel), Psq: the square, Yk: its diagonal-block workspaces.  blk_events
// (optional): event j = "block j of the square factored and the square's rows below it solved"; block j of these rows then
// starts behind event j - 1 (update) and event j (substitution), i.e. the rows follow the square one block behind instead of
// waiting for all of it.
// X <- X L_jj^-T for `rows` rows (the matrix-core substitution), written down for the schedule tracer of dist.hip when there is one
template <typename T>
static int subst_right_traced(lgp_ctx *ctx, hipStream_t st, const T *w, size_t jb, T *X, size_t ldx, size_t rows) {
  if (rows == 0) return LGP_OK;
  if (ctx->tracer != nullptr) {
    long long sw, ow, sx, ox;
    lgp_tr_ptr(ctx, w, sw, ow); lgp_tr_ptr(ctx, X, sx, ox);
    if (!LGP_TRACED(ctx, LGP_TR_SUBST, st, ow, (long long)jb, sx, ox, (long long)ldx, (long long)rows)) return LGP_OK;
  }
  hipLaunchKernelGGL((subst_right_mfma_kernel<T>), dim3((unsigned)((rows + SUBSTM_ROWS - 1) / SUBSTM_ROWS)), dim3(256), 0, st, w, (int)jb, X, (long)ldx,
                     (long)rows);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

template <typename T>
int lgp_far_rows_impl(lgp_ctx *ctx, hipStream_t st, size_t rows, size_t kb, T *Pfar, size_t ldl, const T *Psq, const T *Yk,
                      const hipEvent_t *blk_events) {
  if (rows == 0) return LGP_OK;
  static const int far_rec = getenv("LGP_FAR_RECURSIVE") ? atoi(getenv("LGP_FAR_RECURSIVE")) : 1;     // (as in potrf_split)
  const bool recursive = far_rec != 0 && kb % LGP_DB == 0;
  const size_t nq = (kb + LGP_DB - 1) / LGP_DB;
  size_t j = 0;
  for (size_t j0 = 0; j0 < kb; j0 += LGP_DB, ++j) {
    const size_t jb = (kb - j0) < (size_t)LGP_DB ? (kb - j0) : (size_t)LGP_DB;
    if (j0 > 0 && !recursive) {
      if (blk_events) LGP_CHECK(lgp_tr_wait(ctx, st, blk_events[j - 1]));
      LGP_CHECK(lgp_gemm_launch<T>(ctx, st, false, true, rows, jb, j0, T(-1), Pfar, ldl, Psq + j0 * ldl, ldl, T(1), Pfar + j0, ldl, false, 2));
    }
    if (blk_events) LGP_CHECK(lgp_tr_wait(ctx, st, blk_events[j]));
    LGP_CHECK(subst_right_traced<T>(ctx, st, Yk + (j0 / LGP_DB) * (size_t)LGP_DBLK, jb, Pfar + j0, ldl, rows));
    if (recursive) {     // after block j the columns [c, c + s) get blocks [c - s, c), c = j + 1, s = its lowest set bit (behind event j: the square's rows of those blocks are final)
      const size_t c = j + 1, sbit = c & (~c + 1);
      if (c < nq) {
        const size_t ncol = (c + sbit <= nq ? sbit : nq - c) * LGP_DB, kk = sbit * LGP_DB;
        LGP_CHECK(lgp_gemm_launch<T>(ctx, st, false, true, rows, ncol, kk, T(-1), Pfar + (c - sbit) * LGP_DB, ldl, Psq + (c * LGP_DB) * ldl + (c - sbit) * LGP_DB, ldl,
                                     T(1), Pfar + c * LGP_DB, ldl, false, far_rec == 1 ? 2 : 0));
      }
    }
  }
  return LGP_OK;
}
template int lgp_far_rows_impl<double>(lgp_ctx *, hipStream_t, size_t, size_t, double *, size_t, const double *, const double *, const hipEvent_t *);

// a panel's diagonal square alone (kb x kb, kb <= LGP_NB).  use_chain: the chain kernel's resident workers (8 blocks in ~0.55 ms
// instead of 8 x (potf2 + substitution + update) launches that share the CUs with the trailing update) when it covers the
// size; otherwise launch per operation, recording blk_events[j] (if given) behind block j's substitution so that the rows below
// the square can follow block by block.  With the chain kernel the events are all recorded behind the kernel.
int lgp_square_factor_f64(lgp_ctx *ctx, hipStream_t st, size_t kb, double *P, size_t ldl, double *Yk, size_t row0, bool use_chain,
                          const hipEvent_t *blk_events) {
  typedef double T;
  const size_t nq = (kb + LGP_DB - 1) / LGP_DB;
  if (use_chain && kb >= 256 && (lgp_tr_dry(ctx) || lgp_chain_usable(ctx, kb, ldl))) {
    bool enqueue = true;
    if (ctx->tracer != nullptr) {
      long long sp, op, sw, ow;
      lgp_tr_ptr(ctx, P, sp, op); lgp_tr_ptr(ctx, Yk, sw, ow);
      enqueue = LGP_TRACED(ctx, LGP_TR_CHAINSQ, st, sp, op, (long long)ldl, (long long)kb, ow);
    }
    if (enqueue) {
      unsigned *state = nullptr;
      LGP_CHECK(lgp_chain_state(ctx, kb, &state));
      LGP_HIP(hipMemsetAsync(state, 0, lgp_chain_state_bytes(kb), st));
      LGP_CHECK(lgp_chain_launch(ctx, st, kb, P, ldl, Yk, row0));
    }
    if (blk_events)
      for (size_t j = 0; j < nq; ++j) LGP_CHECK(lgp_tr_record(ctx, blk_events[j], st));
    return LGP_OK;
  }
  size_t j = 0;
  for (size_t j0 = 0; j0 < kb; j0 += LGP_DB, ++j) {
    const size_t jb = (kb - j0) < (size_t)LGP_DB ? (kb - j0) : (size_t)LGP_DB;
    T *w = Yk + j * (size_t)LGP_DBLK;
    LGP_CHECK(lgp_potf2_launch<T>(ctx, st, P + j0 * ldl + j0, ldl, (int)jb, w, row0 + j0));
    const size_t below = kb - (j0 + jb);
    T *X = P + (j0 + jb) * ldl + j0;
    LGP_CHECK(subst_right_traced<T>(ctx, st, w, jb, X, ldl, below));
    if (blk_events) LGP_CHECK(lgp_tr_record(ctx, blk_events[j], st));
    if (below > 0) LGP_CHECK(lgp_gemm_launch<T>(ctx, st, false, true, below, below, jb, T(-1), X, ldl, X, ldl, T(1), P + (j0 + jb) * ldl + (j0 + jb), ldl, true));
  }
  return LGP_OK;
}

template <typename T>
int lgp_potrf_impl(lgp_ctx *ctx, size_t n, T *L, size_t ldl, T *dblk, size_t nx) {
  LGP_REQUIRE(ldl >= n, "ldl < n");
  static const bool v1 = getenv("LGP_POTRF") != nullptr && strcmp(getenv("LGP_POTRF"), "v1") == 0;
  ctx->potf2_small = n <= 8192;         // (no large update beside the diagonal blocks: the four-pivot potf2, potf2_dev.h)
  if (v1 || ctx->stream3 == nullptr) return potrf_lookahead_v1<T>(ctx, n, L, ldl, dblk, nx);
  if constexpr (std::is_same<T, double>::value) {
    if (lgp_chain_usable(ctx, n, ldl) && n <= ctx->chain_max_n) return potrf_chain(ctx, n, L, ldl, dblk, nx, 0, false);
  }
  return potrf_split<T>(ctx, n, L, ldl, dblk, nx);
}

template <typename T>
int lgp_trsm_impl(lgp_ctx *ctx, hipStream_t stream, size_t n, const T *L, size_t ldl, const T *dblk, int trans, T *B, size_t ldb, size_t nrhs) {
  if (n == 0 || nrhs == 0) return LGP_OK;
  LGP_REQUIRE(ldl >= n && ldb >= nrhs, "leading dimension too small");
  LGP_REQUIRE(nrhs < (1u << 30), "too many right-hand sides");
  // Two-level blocking.  The chain of dependent steps is one substitution (128 unknowns, a few microseconds)
  // plus one SMALL gemm that only touches the rest of the current 1024-row outer block; everything below
  // (above, for the transposed solve) the outer block is updated once per outer block by one deep (K = 1024)
  // MFMA gemm off the per-block chain.
  constexpr size_t OB = 1024;
  constexpr size_t per_wg = (size_t)(SUBST_NT / 64) * SUBST_R;
  const dim3 gs((unsigned)((nrhs + per_wg - 1) / per_wg)), bs(SUBST_NT);
  if (!trans) {
    // forward: B_j <- L_jj^-1 B_j (substitution), B_below -= L[below, j] B_j (MFMA gemm)
    for (size_t o0 = 0; o0 < n; o0 += OB) {
      const size_t o1 = (o0 + OB) < n ? (o0 + OB) : n;
      for (size_t j0 = o0; j0 < o1; j0 += LGP_DB) {
        const size_t jb = (o1 - j0) < (size_t)LGP_DB ? (o1 - j0) : (size_t)LGP_DB;
        T *Bj = B + j0 * ldb;
        hipLaunchKernelGGL((subst_left_kernel<T, false>), gs, bs, 0, stream, dblk + (j0 / LGP_DB) * (size_t)LGP_DBLK, (int)jb, Bj, (long)ldb, (int)nrhs);
        LGP_LAUNCH_CHECK();
        const size_t in_block = o1 - (j0 + jb);
        if (in_block > 0)
          LGP_CHECK(lgp_gemm_launch<T>(ctx, stream, false, false, in_block, nrhs, jb, T(-1), L + (j0 + jb) * ldl + j0, ldl, Bj, ldb, T(1),
                                       B + (j0 + jb) * ldb, ldb, false));
      }
      if (o1 < n)
        LGP_CHECK(lgp_gemm_launch<T>(ctx, stream, false, false, n - o1, nrhs, o1 - o0, T(-1), L + o1 * ldl + o0, ldl, B + o0 * ldb, ldb, T(1),
                                     B + o1 * ldb, ldb, false));
    }
  } else {
    // backward: B_j <- L_jj^-T B_j, B_above -= L[j, above]^T B_j
    const size_t nob = (n + OB - 1) / OB;
    for (size_t ob = nob; ob-- > 0;) {
      const size_t o0 = ob * OB, o1 = (o0 + OB) < n ? (o0 + OB) : n;
      const size_t nin = (o1 - o0 + LGP_DB - 1) / LGP_DB;
      for (size_t jj = nin; jj-- > 0;) {
        const size_t j0 = o0 + jj * LGP_DB, jb = (o1 - j0) < (size_t)LGP_DB ? (o1 - j0) : (size_t)LGP_DB;
        T *Bj = B + j0 * ldb;
        hipLaunchKernelGGL((subst_left_kernel<T, true>), gs, bs, 0, stream, dblk + (j0 / LGP_DB) * (size_t)LGP_DBLK, (int)jb, Bj, (long)ldb, (int)nrhs);
        LGP_LAUNCH_CHECK();
        if (j0 > o0)   // rows o0 .. j0 of the same outer block
          LGP_CHECK(lgp_gemm_launch<T>(ctx, stream, true, false, j0 - o0, nrhs, jb, T(-1), L + j0 * ldl + o0, ldl, Bj, ldb, T(1), B + o0 * ldb, ldb,
                                       false));
      }
      if (o0 > 0)      // everything above the outer block: B[0:o0] -= L[o0:o1, 0:o0]^T B[o0:o1]
        LGP_CHECK(lgp_gemm_launch<T>(ctx, stream, true, false, o0, nrhs, o1 - o0, T(-1), L + o0 * ldl, ldl, B + o0 * ldb, ldb, T(1), B, ldb, false));
    }
  }
  return LGP_OK;
}

template <typename T>
__global__ void __launch_bounds__(256) scale_cols_kernel(T *W, long ldw, int rows, int n, const T *s, const unsigned int *nonunit) {
  if (*nonunit == 0) return;
  // extra rows B^T (rows x n): column j /= s[j], so that they carry (S^-1 B)^T into the factorisation of K/s/s'
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < (long)rows * n; idx += (long)gridDim.x * 256) {
    const long r = idx / n, j = idx % n;
    W[r * ldw + j] /= s[j];
  }
}

template <typename T>
static int chol_factor_impl(lgp_ctx *ctx, size_t n, const T *K, size_t ldk, T epsrel, T epsabs, int flags, T *L, size_t ldl,
                            T *dblk, T *heps_out, size_t nrhs = 0) {
  LGP_REQUIRE(ctx != nullptr, "null context");
  LGP_REQUIRE(n >= 1 && n < (1u << 30), "matrix size out of range");
  LGP_REQUIRE(K != nullptr && L != nullptr && dblk != nullptr, "null pointer");
  LGP_REQUIRE(ldk >= n && ldl >= n, "leading dimension too small");
  DeviceGuard dg(ctx->device);
  hipStream_t st = ctx->stream;
  // workspace: s (n), rows (n), stats (16)
  T *ws = nullptr;
  LGP_CHECK(lgp_ws_alloc(ctx, st, (2 * n + 16) * sizeof(T), reinterpret_cast<void **>(&ws)));
  WsGuard ws_guard(ctx, st, ws);
  T *s = ws, *rows = ws + n, *stats = ws + 2 * n;
  LGP_HIP(hipMemsetAsync(stats, 0, 16 * sizeof(T), st));
  LGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), st));
  const int ni = (int)n;
  hipLaunchKernelGGL((diag_scale_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, K, (long)ldk, ni, s, stats);
  LGP_LAUNCH_CHECK();
  // is the scaling trivial?  The kernels read the count of s_i != 1 from device memory (all s_i == 1 lets the big passes
  // skip two divisions per element, with bit-identical results): no host synchronisation until the final read-back
  const unsigned int *nonunit = reinterpret_cast<const unsigned int *>(stats) + 8;
  hipLaunchKernelGGL((rowabs_kernel<T>), dim3((unsigned)n), dim3(256), 0, st, K, (long)ldk, ni, s, nonunit, rows);
  LGP_LAUNCH_CHECK();
  hipLaunchKernelGGL((eps_kernel<T>), dim3(1), dim3(256), 0, st, rows, s, ni, epsrel, epsabs, flags, stats);
  LGP_LAUNCH_CHECK();
  const int nbx = (int)((n + 2047) / 2048);
  hipLaunchKernelGGL((prep_lower_kernel<T>), dim3((unsigned)((size_t)nbx * n)), dim3(256), 0, st, K, (long)ldk, ni, nbx, s, nonunit,
                     stats, L, (long)ldl);
  LGP_LAUNCH_CHECK();
  if (nrhs > 0) {
    hipLaunchKernelGGL((scale_cols_kernel<T>), dim3((unsigned)((nrhs * n + 255) / 256 < 65536 ? (nrhs * n + 255) / 256 : 65536)), dim3(256), 0, st,
                       L + n * ldl, (long)ldl, (int)nrhs, ni, s, nonunit);
    LGP_LAUNCH_CHECK();
  }
  LGP_CHECK(lgp_potrf_impl<T>(ctx, n, L, ldl, dblk, nrhs));
  hipLaunchKernelGGL((scale_rows_kernel<T>), dim3((unsigned)((size_t)nbx * n)), dim3(256), 0, st, L, (long)ldl, ni, nbx, s, nonunit);
  LGP_LAUNCH_CHECK();
  LGP_CHECK(lgp_extract_dblk<T>(ctx, st, L, ldl, n, dblk));     // D, D^T, 1/diag of every diagonal block (read by the solves)
  T hstats[3];
  int info = 0;
  LGP_HIP(hipMemcpyAsync(hstats, stats, 3 * sizeof(T), hipMemcpyDeviceToHost, st));
  LGP_HIP(hipMemcpyAsync(&info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, st));
  LGP_HIP(hipStreamSynchronize(st));
  if (heps_out) *heps_out = hstats[1];
  return lgp_info_result(ctx, info);
}

// Tail of Chol.__init__ for a matrix that the fused Gram kernel (gram.hip, gram_reg_kernel) has already scaled and written as
// a lower triangle: ws = [s (n) | Gershgorin sums (n) | stats (16) | partial column sums (ntr x n)], owned by the caller.
namespace {
template <typename T>
__global__ void __launch_bounds__(256) colpart_reduce_kernel(size_t n, size_t ntr, const T *part, const T *s, T *rows, unsigned int *nonunit) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  T acc = T(0);
  for (size_t t = 0; t < ntr; ++t) acc += part[t * n + j];      // fixed order: deterministic
  rows[j] = acc;
  if (s[j] != T(1)) atomicAdd(nonunit, 1u);
}
template <typename T>
__global__ void __launch_bounds__(256) add_eps_diag_kernel(size_t n, T *L, size_t ldl, const T *epsp) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) L[i * ldl + i] += epsp[0];
}
}  // namespace

template <typename T>
int lgp_chol_from_prepared(lgp_ctx *ctx, size_t n, T *L, size_t ldl, T *dblk, size_t nrhs, T *ws, size_t ntr, T epsrel, T epsabs, int flags,
                           T *heps_out) {
  hipStream_t st = ctx->stream;
  T *s = ws, *rows = ws + n, *stats = ws + 2 * n, *part = ws + 2 * n + 16;
  const int ni = (int)n;
  LGP_HIP(hipMemsetAsync(stats, 0, 16 * sizeof(T), st));
  LGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), st));
  unsigned int *nonunit = reinterpret_cast<unsigned int *>(stats) + 8;
  hipLaunchKernelGGL((colpart_reduce_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, ntr, part, s, rows, nonunit);
  LGP_LAUNCH_CHECK();
  hipLaunchKernelGGL((eps_kernel<T>), dim3(1), dim3(256), 0, st, rows, s, ni, epsrel, epsabs, flags, stats);
  LGP_LAUNCH_CHECK();
  hipLaunchKernelGGL((add_eps_diag_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, L, ldl, stats);
  LGP_LAUNCH_CHECK();
  if (nrhs > 0) {
    hipLaunchKernelGGL((scale_cols_kernel<T>), dim3((unsigned)((nrhs * n + 255) / 256 < 65536 ? (nrhs * n + 255) / 256 : 65536)), dim3(256), 0, st,
                       L + n * ldl, (long)ldl, (int)nrhs, ni, s, nonunit);
    LGP_LAUNCH_CHECK();
  }
  LGP_CHECK(lgp_potrf_impl<T>(ctx, n, L, ldl, dblk, nrhs));
  const int nbx = (int)((n + 2047) / 2048);
  hipLaunchKernelGGL((scale_rows_kernel<T>), dim3((unsigned)((size_t)nbx * n)), dim3(256), 0, st, L, (long)ldl, ni, nbx, s, nonunit);
  LGP_LAUNCH_CHECK();
  if (ctx->defer_finish) {
    // the caller reads eps and the pivot verdict when it synchronises for its own results, and has the diagonal-block images made
    // after that (they are not on the way to a posterior whose solve rode through the factorisation)
    LGP_HIP(hipMemcpyAsync(static_cast<double *>(ctx->h_pinned) + 32, stats, 3 * sizeof(T), hipMemcpyDeviceToHost, st));
    LGP_HIP(hipMemcpyAsync(static_cast<double *>(ctx->h_pinned) + 64, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, st));
    return LGP_OK;
  }
  LGP_CHECK(lgp_extract_dblk<T>(ctx, st, L, ldl, n, dblk));
  T hstats[3];
  int info = 0;
  LGP_HIP(hipMemcpyAsync(hstats, stats, 3 * sizeof(T), hipMemcpyDeviceToHost, st));
  LGP_HIP(hipMemcpyAsync(&info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, st));
  LGP_HIP(hipStreamSynchronize(st));
  if (heps_out) *heps_out = hstats[1];
  return lgp_info_result(ctx, info);
}
template int lgp_chol_from_prepared<double>(lgp_ctx *, size_t, double *, size_t, double *, size_t, double *, size_t, double, double, int, double *);
template int lgp_chol_from_prepared<float>(lgp_ctx *, size_t, float *, size_t, float *, size_t, float *, size_t, float, float, int, float *);

extern "C" int lgp_chol_factor_rhs_f64(lgp_ctx *ctx, size_t n, const double *dK, size_t ldk, double epsrel, double epsabs, int eps_flags,
                                       double *dL, size_t ldl, size_t nrhs, double *ddblk, double *heps_out) {
  return chol_factor_impl<double>(ctx, n, dK, ldk, epsrel, epsabs, eps_flags, dL, ldl, ddblk, heps_out, nrhs);
}
extern "C" int lgp_chol_factor_rhs_f32(lgp_ctx *ctx, size_t n, const float *dK, size_t ldk, float epsrel, float epsabs, int eps_flags,
                                       float *dL, size_t ldl, size_t nrhs, float *ddblk, float *heps_out) {
  return chol_factor_impl<float>(ctx, n, dK, ldk, epsrel, epsabs, eps_flags, dL, ldl, ddblk, heps_out, nrhs);
}

template <typename T>
static int zero_upper_entry(lgp_ctx *ctx, size_t n, T *L, size_t ldl) {
  LGP_REQUIRE(ctx != nullptr && L != nullptr, "null pointer");
  LGP_REQUIRE(n >= 1 && n < (1u << 30) && ldl >= n, "bad dimensions");
  DeviceGuard dg(ctx->device);
  const int nbx = (int)((n + 2047) / 2048);
  hipLaunchKernelGGL((zero_upper_kernel<T>), dim3((unsigned)((size_t)nbx * n)), dim3(256), 0, ctx->stream, L, (long)ldl, (int)n, nbx);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}
extern "C" int lgp_zero_upper_f64(lgp_ctx *ctx, size_t n, double *dL, size_t ldl) { return zero_upper_entry<double>(ctx, n, dL, ldl); }
extern "C" int lgp_zero_upper_f32(lgp_ctx *ctx, size_t n, float *dL, size_t ldl) { return zero_upper_entry<float>(ctx, n, dL, ldl); }

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
  LGP_CHECK(lgp_extract_dblk<T>(ctx, ctx->stream, L, ldl, n, dblk));
  int info = 0;
  LGP_HIP(hipMemcpyAsync(&info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  LGP_HIP(hipStreamSynchronize(ctx->stream));
  return lgp_info_result(ctx, info);
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
  ctx->potf2_small = rows <= 8192;
  LGP_CHECK(lgp_panel_factor_impl<double>(ctx, ctx->stream, rows, nb, dP, ldp, ddblk, row0));
  return lgp_extract_dblk<double>(ctx, ctx->stream, dP, ldp, nb, ddblk);
}

extern "C" int lgp_get_info(lgp_ctx *ctx, int *hinfo) {
  LGP_REQUIRE(ctx != nullptr && hinfo != nullptr, "null pointer");
  DeviceGuard dg(ctx->device);
  LGP_HIP(hipMemcpyAsync(hinfo, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  LGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), ctx->stream));
  LGP_HIP(hipStreamSynchronize(ctx->stream));
  return lgp_info_result(ctx, *hinfo) < 0 ? LGP_EHIP : LGP_OK;
}

template int lgp_panel_factor_impl<double>(lgp_ctx *, hipStream_t, size_t, size_t, double *, size_t, double *, size_t);
template int lgp_panel_factor_impl<float>(lgp_ctx *, hipStream_t, size_t, size_t, float *, size_t, float *, size_t);
template int lgp_potrf_impl<double>(lgp_ctx *, size_t, double *, size_t, double *, size_t);
template int lgp_potrf_impl<float>(lgp_ctx *, size_t, float *, size_t, float *, size_t);
template int lgp_trsm_impl<double>(lgp_ctx *, hipStream_t, size_t, const double *, size_t, const double *, int, double *, size_t, size_t);
template int lgp_trsm_impl<float>(lgp_ctx *, hipStream_t, size_t, const float *, size_t, const float *, int, float *, size_t, size_t);

int lgp_extract_dblk_public(lgp_ctx *ctx, hipStream_t stream, const double *L, size_t ldl, size_t n, double *dblk) {
  return lgp_extract_dblk<double>(ctx, stream, L, ldl, n, dblk);
}

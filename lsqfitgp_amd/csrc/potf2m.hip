// potf2m.hip -- the diagonal-block Cholesky (potf2) of the blocked factorisation on the matrix cores of gfx950, fp64.
// Replaces the LAPACK potrf that jlinalg.cholesky lowers to (src/lsqfitgp/_linalg/_decomp.py:388) for the 128x128 diagonal
// blocks; the driver is in potrf.hip.
#include "potf2_dev.h"
#include <algorithm>
#include <cstdlib>

namespace {

// potf2 on the matrix cores (fp64): ONE WAVE factors the 128x128 diagonal block, held in registers as the 36 upper
// 16x16 tiles of U = L^T in MFMA ACCUMULATOR layout: tile (j, i), j <= i (block row j, block column i), register r, lane
// (n = lane & 15, g = lane >> 4) holds U[16 j + 4 r + g][16 i + n].  In that layout the accumulator registers of a tile
// ARE matrix-core operands: register r as A operand is the k-slab r of the tile TRANSPOSED, as B operand the k-slab
// itself, so   sum_r mfma(A = P[r], B = Q[r])  =  P^T Q   with no data movement at all.  Block step s (right-looking,
// U^T U form):
//   1. the 16 rows of block row s (diagonal tile and the tiles right of it, all at once) are eliminated in slabs of 4
//      rows = one register: per pivot row k  a readlane of the pivot, 1 / sqrt on every lane, two ds_bpermute
//      broadcasts (row k to every lane group, the pivot row's entries at the slab's own columns) and ONE fma per tile;
//      the UNSCALED row is broadcast, so the broadcasts run beside the reciprocal square root and the dependent chain
//      of a pivot is  readlane -> rsq + 2 Newton steps -> square -> fma;
//      after its 4 pivots a slab is applied to the later rows of every tile of the block row by ONE mfma per tile
//      (A = the diagonal tile's register, masked to the later rows, negated; B = the tile's register);
//   2. trailing tiles (j, i), s < j <= i:  t_ji -= U_sj^T U_si = 4 mfma with A = -t_sj[r], B = t_si[r].
// No LDS memory, no barrier, ~130 pivot chains of ~10 dependent operations and 444 MFMAs in all; the VALU potf2_kernel
// (potrf.hip, still used for fp32) needs a workgroup barrier and an LDS round trip per pivot (450 ns each).  The kernel also writes the fragment
// image that subst_right_mfma_kernel reads (the former pack_frag_kernel): off-diagonal fragments are tile registers
// times -1/diag of the lane's row, in lane order already.
// The lower part of a diagonal tile (column < row) is never read as data: the in-slab update only uses columns right of
// the pivot, the slab MFMA masks its A operand to later rows, and the stores mask it.

template <int S>
__device__ __forceinline__ void potf2m_step(acc_t (&t)[36], double (&rdn)[8], int n, int g, int nb, int *info, long row0) {
  constexpr int DI = S * (S + 1) / 2 + S;
  constexpr int NT = 8 - S;                                        // tiles of the block row
  double dn = 1.0;                                                 // pivot of row n of the tile, in lane n (every group)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    // the slab (register r of every tile of the block row) as plain scalars for its four pivots: the tiles themselves
    // may live in accumulator registers, which vector instructions cannot address
    double w[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) w[i] = t[(i + S) * (i + S + 1) / 2 + S][r];
    double dg = 1.0;                                               // pivot of the lane group's own row of the slab
#pragma unroll 1
    for (int g0 = 0; g0 < 4; ++g0) {
      const int kk = 4 * r + g0;                                   // pivot row inside the tile
      const double d = readlane_d(w[0], g0 * 16 + kk);
      // every broadcast of the (unscaled) pivot row is issued before anything waits for one: the LDS crossbar round
      // trips overlap each other and the reciprocal below
      const int arow = (g0 * 16 + n) * 4;                          // row kk, column n        -> all lane groups
      const int acol = (g0 * 16 + 4 * r + g) * 4;                  // row kk, column 4 r + g  -> lane group g
      const double cf = MmaD::bpermute(acol, w[0]);
      double rn[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) rn[i] = MmaD::bpermute(arow, w[i]);
      // the dependent chain of a pivot: readlane -> rcp + 3 fma -> fma on the diagonal tile.  (An in-order wave pays the
      // latency of every dependent sequence it issues, so everything that can wait is done once per slab instead.)
      const double rc = rcp_chain(d);
      const double cfm = (g > g0) ? -cf : 0.0;                     // only the rows of the slab below the pivot row
      w[0] = __builtin_fma(cfm * rn[0], rc, w[0]);
      const double cfs = cfm * rc;                                 // -a_k,k' / d_k
#pragma unroll
      for (int i = 1; i < NT; ++i) w[i] = __builtin_fma(cfs, rn[i], w[i]);
      dg = (g == g0) ? d : dg;
      dn = (n == kk) ? d : dn;
    }
    // once per slab, all four pivots at once (one per lane group): validity, 1 / sqrt, scaling of the finished rows
    const bool badv = !(dg > 0.0) || (dg + dg == dg);              // <= 0, NaN, or +inf
    const unsigned long long badm = __builtin_amdgcn_ballot_w64(badv);
    if (badm != 0) {                                               // rare; wave-uniform
      const int gb = (__builtin_ctzll(badm) >> 4);                 // first bad pivot of the slab
      if (n == 0 && g == 0 && 16 * S + 4 * r + gb < nb) atomicCAS(info, 0, (int)(row0 + 16 * S + 4 * r + gb + 1));
    }
    const double sclv = rsqrt_d(badv ? (double)NAN : dg);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      w[i] *= sclv;
      t[(i + S) * (i + S + 1) / 2 + S][r] = w[i];
    }
    if (r < 3) {
      // the finished slab -> later rows of every tile of the block row
      const double am = (n > 4 * r + 3) ? -w[0] : 0.0;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int ti = (i + S) * (i + S + 1) / 2 + S;
        t[ti] = MmaD::run(am, w[i], t[ti]);
      }
    }
  }
  rdn[S] = rsqrt_d(dn);                             // 1 / L_kk for lane n == k (any group)
  // trailing tiles
#pragma unroll
  for (int j = S + 1; j < 8; ++j) {
    const int tj = j * (j + 1) / 2 + S;
    const acc_t neg = -t[tj];
#pragma unroll
    for (int i = j; i < 8; ++i) {
      const int ti = i * (i + 1) / 2 + S, tt = i * (i + 1) / 2 + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) t[tt] = MmaD::run(neg[r], t[ti][r], t[tt]);
    }
  }
}

#ifdef LGP_POTF2_TRACE
__device__ unsigned long long lgp_potf2_trace[16];
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
potf2_mfma_kernel(double *A, long lda, int nb, double *dblk, int *info, long row0) {
  __builtin_amdgcn_s_setprio(LGP_CHAIN_PRIO);       // head of the dependent chain, see potf2_kernel in potrf.hip
  const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
  acc_t t[36];
  const __amdgpu_buffer_rsrc_t abuf = __builtin_amdgcn_make_buffer_rsrc(A, 0, 0x7fffff00, 0x00020000);
  const int lbase = (int)(((long)n * lda + g) * 8), ldstep = (int)(16 * lda * 8);   // per-lane offset, per-tile-row (scalar) offset
  // lower triangle of the block, transposed into the tiles (identity padding beyond nb; zero below the diagonal
  // of the diagonal tiles): t[j, i][r] <- A[16 i + n][16 j + 4 r + g]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * i + n, col = 16 * j + 4 * r + g;
        const bool in = row < nb && col <= row;
        // raw buffer load, out-of-range offset (-> 0) for the masked lanes: no branch around the load, all 144 loads of a
        // lane are in flight together
        const double v = MmaD::load(abuf, in ? lbase + (16 * j + 4 * r) * 8 : 0x7ffffff0, i * ldstep);
        t[i * (i + 1) / 2 + j][r] = (!in && row == col) ? 1.0 : v;
      }
  double rdn[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) rdn[c] = 1.0;
#ifdef LGP_POTF2_TRACE
#define LGP_PSTAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (lane == 0) lgp_potf2_trace[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LGP_PSTAMP(k) do { } while (0)
#endif
  LGP_PSTAMP(0);
  potf2m_step<0>(t, rdn, n, g, nb, info, row0);
  LGP_PSTAMP(1);
  potf2m_step<1>(t, rdn, n, g, nb, info, row0);
  LGP_PSTAMP(2);
  potf2m_step<2>(t, rdn, n, g, nb, info, row0);
  LGP_PSTAMP(3);
  potf2m_step<3>(t, rdn, n, g, nb, info, row0);
  LGP_PSTAMP(4);
  potf2m_step<4>(t, rdn, n, g, nb, info, row0);
  LGP_PSTAMP(5);
  potf2m_step<5>(t, rdn, n, g, nb, info, row0);
  LGP_PSTAMP(6);
  potf2m_step<6>(t, rdn, n, g, nb, info, row0);
  LGP_PSTAMP(7);
  potf2m_step<7>(t, rdn, n, g, nb, info, row0);
  LGP_PSTAMP(8);

  // ---- L block back to the matrix (lower part only)
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * i + n, col = 16 * j + 4 * r + g;
        if (row < nb && col <= row) A[(long)row * lda + col] = t[i * (i + 1) / 2 + j][r];   // (buffer stores here crash ROCm 7.2's clang)
      }
  LGP_PSTAMP(9);
#ifdef LGP_POTF2_TRACE
  // experiment: the last three block steps again, three times in a rolled loop (same code: instruction cache hot from the 2nd pass)
#pragma unroll 1
  for (int rep = 0; rep < 3; ++rep) {
    potf2m_step<5>(t, rdn, n, g, nb, info, row0);
    potf2m_step<6>(t, rdn, n, g, nb, info, row0);
    potf2m_step<7>(t, rdn, n, g, nb, info, row0);
    LGP_PSTAMP(11 + rep);
  }
#endif
  // ---- fragment image for subst_right_mfma_kernel (layout: see the table above pack_frag_kernel); M = diag^-1 L
  const __amdgpu_buffer_rsrc_t fbuf = __builtin_amdgcn_make_buffer_rsrc(dblk + LGP_FRAG0, 0, LGP_NFRAG * 64 * 8, 0x00020000);
  const int fl = lane * 8;
  auto putf = [&](int f, double v) {       // fragment f: lane order, 512 contiguous bytes; scalar offset, no address arithmetic
    typedef unsigned int u2_t __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, v), fbuf, fl, f * 512, 0);
  };
#pragma unroll
  for (int c = 1; c < 8; ++c) {
    const double nrd = -rdn[c];
#pragma unroll
    for (int i = 0; i < c; ++i)
#pragma unroll
      for (int s = 0; s < 4; ++s)          // off(c, i, s) = -M[16 c + n][16 i + 4 s + g] = -U[16 i + 4 s + g][16 c + n] / L_(16c+n)
        putf((c * (c - 1) / 2 + i) * 4 + s, t[c * (c + 1) / 2 + i][s] * nrd);
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int dc = c * (c + 1) / 2 + c;
    const double nrd = -rdn[c];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double rdg = MmaD::bpermute((4 * r + g) * 4, rdn[c]);     // 1 / L at row 16 c + 4 r + g
      if (r < 3) putf(112 + 3 * c + r, (n >= 4 * (r + 1)) ? t[dc][r] * nrd : 0.0);
#pragma unroll
      for (int a = 0; a < 3; ++a) {        // coef(c, r, a) = -M[16 c + 4 r + g][16 c + 4 r + a] = -U[.. + a][.. + g] / L_(..+g), g > a
        const double u = MmaD::bpermute((a * 16 + 4 * r + g) * 4, t[dc][r]);
        putf(136 + (4 * c + r) * 3 + a, (g > a) ? -(u * rdg) : 0.0);
      }
      putf(232 + 4 * c + r, rdg);
    }
  }
  LGP_PSTAMP(10);
}


// ------------------------------------------------------------------------------------------------
// The same algorithm on FOUR waves (one workgroup), so that it fits where the chain runs: beside the trailing update every
// SIMD holds two 224-register waves and a kernel that needs a whole register file (the one-wave kernel above) waits for a
// SIMD to drain -- 0.58 ms on average in the N = 32768 profile.  Four waves of ~150 registers and 18 KB of LDS fit the slot
// of ONE retiring trailing-update workgroup.
//   * wave w owns block columns w and 7 - w of U (9 tiles): tA[j] = tile (j, w), j <= w; tB[j] = tile (j, 7 - w), j <= 7 - w;
//   * block step s: the owner of the diagonal tile publishes it in LDS; EVERY wave runs the 16-pivot chain on its copy
//     (redundantly: no hand-over inside the chain) and eliminates its own (at most two) tiles of block row s beside it;
//     the eliminated tiles go to LDS; every wave updates its trailing tiles with A operands read from there.
//   Two barriers per block step, 16 in all.
template <bool QUAD>
__global__ void __launch_bounds__(256) potf2_mfma4_kernel(double *A, long lda, int nb, double *dblk, int *info, long row0) {
  __shared__ double rowbuf[P4_ROWBUF];
  __shared__ double diagbuf[4 * 64];
  __builtin_amdgcn_s_setprio(LGP_CHAIN_PRIO);
#define P4_QUAD_ARG QUAD
#include "potf2m4_block.inc"
#undef P4_QUAD_ARG
}

}  // namespace

int lgp_potf2_mfma_launch(lgp_ctx *ctx, hipStream_t stream, double *A, size_t lda, int nb, double *dblk, size_t row0) {
  LGP_REQUIRE(lda < (1u << 20), "leading dimension too large for the 32-bit offsets of the diagonal-block kernel");
  if (ctx->potf2_mode != 3) {
    // Which pivot scheme beside the updates of a large factorisation (potf2m4_step says why): the quad beside the 128-tile updates,
    // where it holds the look-ahead back; one by one beside the 64-tile updates of the end, where the look-ahead is what the step
    // waits for (C3 191.6 vs 191.9 ms, N = 20480 51.0-51.1 vs 51.3-51.4 on one box).  LGP_POTF2_INSITU=quad | pivot | late (default).
    static const int insitu = getenv("LGP_POTF2_INSITU") == nullptr ? 2 : (strcmp(getenv("LGP_POTF2_INSITU"), "pivot") == 0 ? 1 : (strcmp(getenv("LGP_POTF2_INSITU"), "late") == 0 ? 2 : 0));
    const bool pivot_now = !ctx->potf2_small && (insitu == 1 || (insitu == 2 && ctx->potf2_late));
    if (LGP_P4_QUAD && !pivot_now) hipLaunchKernelGGL(potf2_mfma4_kernel<true>, dim3(1), dim3(256), 0, stream, A, (long)lda, nb, dblk, ctx->d_info, (long)row0);
    else hipLaunchKernelGGL(potf2_mfma4_kernel<false>, dim3(1), dim3(256), 0, stream, A, (long)lda, nb, dblk, ctx->d_info, (long)row0);
    LGP_LAUNCH_CHECK();
    return LGP_OK;
  }
  hipLaunchKernelGGL(potf2_mfma_kernel, dim3(1), dim3(64), 0, stream, A, (long)lda, nb, dblk, ctx->d_info, (long)row0);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}


// potf2_dev.h -- device pieces shared by the diagonal-block kernels (potf2m.hip) and the chain kernel (chaint.hip).
#pragma once
#include "chain.h"

namespace {

typedef Mma<double> MmaD;
typedef MmaD::acc_t acc_t;

__device__ __forceinline__ double readlane_d(double v, int lane) {
  const long long bits = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)bits, lane), hi = __builtin_amdgcn_readlane((int)(bits >> 32), lane);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// 1 / sqrt(d): hardware estimate + two Newton steps, y <- y (1.5 - 0.5 d y^2)
__device__ __forceinline__ double rsqrt_d(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  y = y * (1.5 - h * y * y);
  y = y * (1.5 - h * y * y);
  return y;
}

__device__ __forceinline__ int ut_idx(int j, int i) { return i * (i + 1) / 2 + j; }   // tile (block row j, block column i), j <= i

// 1 / d to working accuracy with a short dependent chain: hardware estimate (~2^-24 relative) and ONE cubic step,
// y (1 + e + e^2) with e = 1 - d y  (error e^3): rcp, fma, fma, fma
__device__ __forceinline__ double rcp_chain(double d) {
  const double y = __builtin_amdgcn_rcp(d);
  const double e = __builtin_fma(-d, y, 1.0);
  const double e2 = __builtin_fma(e, e, e);
  return __builtin_fma(y, e2, y);
}

#ifdef LGP_POTF2_TRACE
__device__ unsigned long long lgp_p4_trace[64];
#define LGP_P4STAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (threadIdx.x == 0) lgp_p4_trace[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LGP_P4STAMP(k) do { } while (0)
#endif

// pivot 4 r + G0 of a slab (register r of the tiles of a block row): w0 the diagonal tile's register, w1 / w2 the wave's own tiles'.
// Rows g <= G0 of the slab are not touched (their multiplier stays zero); the last pivot of a slab eliminates nothing in it.
template <int G0>
__device__ __forceinline__ void p4_pivot(int r, double &w0, double &w1, double &w2, double &dg, double &dn, int n, int g, bool hasA, bool hasB) {
  const int kk = 4 * r + G0;
  const double d = readlane_d(w0, G0 * 16 + kk);
  if constexpr (G0 < 3) {
    // the pivot row, in every row: on the chain (the diagonal tile) by lane swaps; for the wave's own tiles, which nothing waits
    // for inside the slab, through the LDS crossbar (one instruction per half instead of four; skipped for tiles it does not own)
    const double rn0 = lgp_rowbcast(w0, G0);
    const int arow = (G0 * 16 + n) * 4;
    const double rn1 = hasA ? MmaD::bpermute(arow, w1) : 0.0, rn2 = hasB ? MmaD::bpermute(arow, w2) : 0.0;
    // its entries at the columns 4 r + g of the slab's later rows g > G0: lane 4 r + g of the own row, by DPP moves that do not
    // depend on one another (disjoint rows, zero elsewhere), then or-ed together
    double c1 = 0.0, c2 = 0.0, c3 = 0.0;
    switch (r) {                                        // (r is a constant after unrolling; the DPP lane has to be an immediate)
      case 0:
        if constexpr (G0 < 1) c1 = lgp_row_lane<1, 0x2>(0.0, rn0);
        if constexpr (G0 < 2) c2 = lgp_row_lane<2, 0x4>(0.0, rn0);
        c3 = lgp_row_lane<3, 0x8>(0.0, rn0);
        break;
      case 1:
        if constexpr (G0 < 1) c1 = lgp_row_lane<5, 0x2>(0.0, rn0);
        if constexpr (G0 < 2) c2 = lgp_row_lane<6, 0x4>(0.0, rn0);
        c3 = lgp_row_lane<7, 0x8>(0.0, rn0);
        break;
      case 2:
        if constexpr (G0 < 1) c1 = lgp_row_lane<9, 0x2>(0.0, rn0);
        if constexpr (G0 < 2) c2 = lgp_row_lane<10, 0x4>(0.0, rn0);
        c3 = lgp_row_lane<11, 0x8>(0.0, rn0);
        break;
      default:
        if constexpr (G0 < 1) c1 = lgp_row_lane<13, 0x2>(0.0, rn0);
        if constexpr (G0 < 2) c2 = lgp_row_lane<14, 0x4>(0.0, rn0);
        c3 = lgp_row_lane<15, 0x8>(0.0, rn0);
        break;
    }
    const double cf = __builtin_bit_cast(double, __builtin_bit_cast(long long, c1) | __builtin_bit_cast(long long, c2) | __builtin_bit_cast(long long, c3));
    const double rc = rcp_chain(d);
    const double cfm = -cf;
    w0 = __builtin_fma(cfm * rn0, rc, w0);
    const double cfs = cfm * rc;
    if (hasA) w1 = __builtin_fma(cfs, rn1, w1);
    if (hasB) w2 = __builtin_fma(cfs, rn2, w2);
  }
  dg = (g == G0) ? d : dg;
  dn = (n == kk) ? d : dn;
}

// ---- the four pivots 4 r .. 4 r + 3 of a slab TOGETHER (round 5; -DLGP_P4_QUAD=0 compiles the pivot-by-pivot code above instead).
// The 4 x 4 pivot block is read out of the diagonal tile once (10 values, uniform) and factored as U' D U in uniform arithmetic:
// the six multipliers and four pivots of the quad are known after three reciprocals, none of which waits for a lane crossing.
// The rows are then eliminated in the usual order -- row G of every tile as it is after the steps before it, broadcast to the
// other lane groups through the LDS crossbar, times the lane group's own multiplier -- so the arithmetic on the rows is that of
// Gaussian elimination (a first version applied the accumulated transformation T = inverse of the unit lower factor to the
// original rows in one go: equal speed, but for smooth kernels with a small jitter the residual |L L' - A| of a block went
// from 1e-15 to 1e-13, tools/p4_probe.hip).
__device__ __forceinline__ double lgp_opaque(double v) {        // (keeps the compiler from sinking a uniform computation into the lanes that select it)
  asm volatile("" : "+v"(v));
  return v;
}

// XBAR: the row broadcasts through the LDS crossbar (ds_bpermute: two instructions per broadcast); not XBAR: by lane swaps (eight).
template <bool XBAR>
__device__ __forceinline__ double p4_bcast(double v, int G, int addr) {
  if constexpr (XBAR) return MmaD::bpermute(addr, v);
  else return lgp_rowbcast(v, G);
}

template <bool XBAR>
__device__ __forceinline__ void p4_quad(int r, double &w0, double &w1, double &w2, double &dg, double &dn, int n, int g) {
  const int c0 = 4 * r;
  // (row 0 of the quad is final as it stands: its broadcasts are asked for first, so that the crossbar's ~100 cycles pass beside the
  // uniform arithmetic below and not behind it)
  const int a0 = n * 4, a1 = (16 + n) * 4, a2 = (32 + n) * 4;
  double x = p4_bcast<XBAR>(w0, 0, a0), y = p4_bcast<XBAR>(w1, 0, a0), z = p4_bcast<XBAR>(w2, 0, a0);
  // the 4 x 4 pivot block, upper part: entry (G, G') of the slab's register r sits at lane 16 G + 4 r + G' of the diagonal tile
  const double b00 = readlane_d(w0, c0), b01 = readlane_d(w0, c0 + 1), b02 = readlane_d(w0, c0 + 2), b03 = readlane_d(w0, c0 + 3);
  const double b11 = readlane_d(w0, 16 + c0 + 1), b12 = readlane_d(w0, 16 + c0 + 2), b13 = readlane_d(w0, 16 + c0 + 3);
  const double b22 = readlane_d(w0, 32 + c0 + 2), b23 = readlane_d(w0, 32 + c0 + 3);
  const double b33 = readlane_d(w0, 48 + c0 + 3);
  // U' D U of it (the multiplier m_kG = u_kG / d_k eliminates row G with pivot row k)
  const double rc0 = rcp_chain(b00);
  const double m01 = lgp_opaque(b01 * rc0), m02 = lgp_opaque(b02 * rc0), m03 = lgp_opaque(b03 * rc0);
  const double d1 = __builtin_fma(-m01, b01, b11), c12 = __builtin_fma(-m01, b02, b12), c13 = __builtin_fma(-m01, b03, b13);
  const double c22 = __builtin_fma(-m02, b02, b22), c23 = __builtin_fma(-m02, b03, b23), c33 = __builtin_fma(-m03, b03, b33);
  const double rc1 = rcp_chain(d1);
  const double m12 = lgp_opaque(c12 * rc1), m13 = lgp_opaque(c13 * rc1);
  const double d2 = __builtin_fma(-m12, c12, c22), e23 = __builtin_fma(-m12, c13, c23), e33 = __builtin_fma(-m13, c13, c33);
  const double rc2 = rcp_chain(d2);
  const double m23 = lgp_opaque(e23 * rc2);
  const double d3 = lgp_opaque(__builtin_fma(-m23, e23, e33));
  const double d1o = lgp_opaque(d1), d2o = lgp_opaque(d2);
  // the lane group's own multiplier in each of the three elimination steps (zero where a step leaves the row alone)
  const bool g1 = g == 1, g2 = g == 2, g3 = g == 3;
  double q0 = g1 ? m01 : 0.0; q0 = g2 ? m02 : q0; q0 = g3 ? m03 : q0;
  double q1 = g2 ? m12 : 0.0; q1 = g3 ? m13 : q1;
  const double q2 = g3 ? m23 : 0.0;
  w0 = __builtin_fma(-q0, x, w0); w1 = __builtin_fma(-q0, y, w1); w2 = __builtin_fma(-q0, z, w2);
  x = p4_bcast<XBAR>(w0, 1, a1); y = p4_bcast<XBAR>(w1, 1, a1); z = p4_bcast<XBAR>(w2, 1, a1);
  w0 = __builtin_fma(-q1, x, w0); w1 = __builtin_fma(-q1, y, w1); w2 = __builtin_fma(-q1, z, w2);
  x = p4_bcast<XBAR>(w0, 2, a2); y = p4_bcast<XBAR>(w1, 2, a2); z = p4_bcast<XBAR>(w2, 2, a2);
  w0 = __builtin_fma(-q2, x, w0); w1 = __builtin_fma(-q2, y, w1); w2 = __builtin_fma(-q2, z, w2);
  double dsel = g1 ? d1o : b00; dsel = g2 ? d2o : dsel; dsel = g3 ? d3 : dsel;
  dg = dsel;
  const int nq = n & 3;
  double dcol = nq == 1 ? d1o : b00; dcol = nq == 2 ? d2o : dcol; dcol = nq == 3 ? d3 : dcol;
  dn = ((n >> 2) == r) ? dcol : dn;
}

#ifndef LGP_P4_QUAD
#define LGP_P4_QUAD 1
#endif
#ifndef P4_XBAR
#define P4_XBAR 1          // the quad's row broadcasts through the LDS crossbar (0: lane swaps -- 426 instead of 300 cycles per pivot alone,
#endif                     // and no faster beside the large update: what costs there is the register file, not the crossbar)

// ---- the four-wave potf2 (potf2m.hip: the stand-alone kernel; chaint.hip: task P of the task-graph chain kernel)
constexpr int P4_ROWBUF = 8 * 4 * 64;   // block row s: tile of block column i, register r, lane  (doubles)

// QUAD: the four pivots of a register together (`p4_quad`) or one by one (`p4_pivot`).  The quad is the faster block alone (36.5
// against 40.7 us) but needs 256 + 128 registers where the pivot-by-pivot code needs 236: beside the large update, whose waves hold
// 2 x 224 of a SIMD's 512, a 236-register workgroup fits the slot of ONE retiring workgroup, a 384-register one has to find a CU
// with BOTH slots free -- 650-730 us per block on average in the C3 step (10 ms at worst) against 186-192.  Measured on one box
// (LGP_POTF2_INSITU=pivot), the C3 step is nevertheless FASTER with the quad everywhere: 187.1-187.4 against 187.9-188.0 ms.  With
// the pivot-by-pivot blocks the look-ahead gets in at once, the stream of the large updates idles 0.6 instead of 3.5-4.7 ms
// between them -- and the large updates take 150.8 instead of 146.0 ms, because the look-ahead's kernels then run beside them for
// longer.  A look-ahead that is held back interferes less: the quad is the default beside the large (128-tile) updates; beside the
// 64-tile updates of the last panels, where the step waits for the look-ahead, the pivot-by-pivot kernel is (potf2m.hip).
template <int S, bool QUAD>
__device__ __forceinline__ void potf2m4_step(acc_t (&tA)[4], acc_t (&tB)[8], double (&rdn)[8], double *rowbuf, double *diagbuf, int w, int lane,
                                             int n, int g, int nb, int *info, long row0) {
  constexpr bool DIAG_IN_A = S <= 3;               // the diagonal tile (S, S) is tile S of block column S: column w = S or 7 - w = S
  const int wd = DIAG_IN_A ? S : 7 - S;            // its owner
  const bool hasA = S < w;                         // this wave owns a tile of block row S in its column A (right of the diagonal)
  const bool hasB = S < 7 - w;                     // ... in its column B
  // ---- 1. publish the diagonal tile
  if (w == wd) {
#pragma unroll
    for (int r = 0; r < 4; ++r) diagbuf[r * 64 + lane] = DIAG_IN_A ? tA[S < 4 ? S : 0][r] : tB[S][r];
  }
  __syncthreads();
  acc_t dt;
#pragma unroll
  for (int r = 0; r < 4; ++r) dt[r] = diagbuf[r * 64 + lane];
  LGP_P4STAMP(2 + 4 * S);
  // (QUAD: a wave that owns no tile of this block row in a column works on whatever that register holds: never stored, and cheaper
  // than the zeroing selects, which the compiler rematerialised in front of every matrix-core instruction)
  acc_t ta = (QUAD || hasA) ? tA[S < 4 ? S : 0] : acc_t{0, 0, 0, 0};
  acc_t tb = (QUAD || hasB) ? tB[S] : acc_t{0, 0, 0, 0};
  double dn = 1.0;
  // ---- 2. the 16 pivots: diagonal tile copy + the wave's own tiles of the block row.  Lane crossing by VALU swaps and DPP
  // (chain.h): per pivot  row broadcast of the pivot row -> its entry at the lane's own row index -> multiply -> fma.
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double w0 = dt[r], w1 = ta[r], w2 = tb[r];
    double dg = 1.0;
    if constexpr (QUAD) {
      p4_quad<P4_XBAR != 0>(r, w0, w1, w2, dg, dn, n, g);
    } else {
      p4_pivot<0>(r, w0, w1, w2, dg, dn, n, g, hasA, hasB);
      p4_pivot<1>(r, w0, w1, w2, dg, dn, n, g, hasA, hasB);
      p4_pivot<2>(r, w0, w1, w2, dg, dn, n, g, hasA, hasB);
      p4_pivot<3>(r, w0, w1, w2, dg, dn, n, g, hasA, hasB);
    }
    const bool badv = !(dg > 0.0) || (dg + dg == dg);
    const unsigned long long badm = __builtin_amdgcn_ballot_w64(badv);
    if (badm != 0 && w == wd) {
      const int gb = (__builtin_ctzll(badm) >> 4);
      if (n == 0 && g == 0 && 16 * S + 4 * r + gb < nb) atomicCAS(info, 0, (int)(row0 + 16 * S + 4 * r + gb + 1));
    }
    const double sclv = rsqrt_d(badv ? (double)NAN : dg);
    w0 *= sclv; w1 *= sclv; w2 *= sclv;
    dt[r] = w0; ta[r] = w1; tb[r] = w2;
    // the finished rows of the wave's own tiles -> LDS (A operands of everybody's trailing updates)
    if (hasA) rowbuf[(w * 4 + r) * 64 + lane] = w1;
    if (hasB) rowbuf[((7 - w) * 4 + r) * 64 + lane] = w2;
    if (r < 3) {
      const double am = (n > 4 * r + 3) ? -w0 : 0.0;
      dt = MmaD::run(am, w0, dt);
      ta = MmaD::run(am, w1, ta);
      tb = MmaD::run(am, w2, tb);
    }
  }
  rdn[S] = rsqrt_d(dn);
  if (w == wd) {
    if (DIAG_IN_A) tA[S < 4 ? S : 0] = dt; else tB[S] = dt;
  }
  if (hasA) tA[S < 4 ? S : 0] = ta;
  if (hasB) tB[S] = tb;
  LGP_P4STAMP(3 + 4 * S);
  __syncthreads();
  LGP_P4STAMP(4 + 4 * S);
  // ---- 3. trailing tiles of the wave's two block columns: t(j, i) -= U_sj^T U_si, the A operand from LDS
#pragma unroll
  for (int j = S + 1; j < 8; ++j) {
    const bool doA = hasA && j <= w, doB = hasB && j <= 7 - w;      // wave-uniform
    if (!(doA || doB)) continue;
    double neg[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) neg[r] = -rowbuf[(j * 4 + r) * 64 + lane];
    if (j < 4) {
      if (doA) {
#pragma unroll
        for (int r = 0; r < 4; ++r) tA[j < 4 ? j : 0] = MmaD::run(neg[r], ta[r], tA[j < 4 ? j : 0]);
      }
    }
    if (doB) {
#pragma unroll
      for (int r = 0; r < 4; ++r) tB[j] = MmaD::run(neg[r], tb[r], tB[j]);
    }
  }
  LGP_P4STAMP(5 + 4 * S);
}

}  // namespace

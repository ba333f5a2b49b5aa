// chain.h -- device code shared by the kernels on the dependent chain of the blocked factorisation (potrf.hip, potf2m.hip):
// the matrix-core wrappers and the right-side substitution of one wave.
#pragma once
#include "common.h"

#ifndef LGP_CHAIN_PRIO
#define LGP_CHAIN_PRIO 3
#endif

namespace {

template <typename T> struct Mma;
template <> struct Mma<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ double load(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    typedef unsigned int u2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(double, (u2_t)__builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0));
  }
  static __device__ __forceinline__ double bpermute(int addr, double v) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_ds_bpermute(addr, (int)bits), hi = __builtin_amdgcn_ds_bpermute(addr, (int)(bits >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
  }
};
template <> struct Mma<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ float load(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    return __builtin_bit_cast(float, (unsigned int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
  }
  static __device__ __forceinline__ float bpermute(int addr, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
  }
};

// ---- lane crossing without LDS.  ds_bpermute goes through the LDS crossbar of the CU: 98 cycles of dependent latency, and the
// four waves of a workgroup that all broadcast at once queue behind one another (the four-wave potf2 spent 512 cycles per
// pivot there, tools/potf2_probe.hip).  gfx950 has VALU lane swaps instead: a 16-lane ROW of a register reaches all four rows
// in two steps (v_permlane16_swap, v_permlane32_swap: 72 cycles dependent, per SIMD, no shared resource), and v_mov_dpp
// row_newbcast with a row mask gives every row a different lane of itself (40 cycles).  tools/permlane_probe.hip.
typedef unsigned int lgp_u2_t __attribute__((ext_vector_type(2)));
// row g0 (lanes 16 g0 .. 16 g0 + 15) of x, in every row
__device__ __forceinline__ unsigned int lgp_rowbcast_b32(unsigned int x, int g0) {
  const lgp_u2_t a = __builtin_amdgcn_permlane16_swap(x, x, false, false);       // .x = rows [0 0 2 2], .y = rows [1 1 3 3]
  const unsigned int h = (g0 & 1) ? a.y : a.x;
  const lgp_u2_t b = __builtin_amdgcn_permlane32_swap(h, h, false, false);       // .x = lower rows everywhere, .y = upper rows
  return (g0 & 2) ? b.y : b.x;
}
__device__ __forceinline__ double lgp_rowbcast(double v, int g0) {
  const long long bits = __builtin_bit_cast(long long, v);
  const unsigned int lo = lgp_rowbcast_b32((unsigned int)bits, g0), hi = lgp_rowbcast_b32((unsigned int)(bits >> 32), g0);
  return __builtin_bit_cast(double, ((long long)hi << 32) | lo);
}
__device__ __forceinline__ float lgp_rowbcast(float v, int g0) {
  return __builtin_bit_cast(float, lgp_rowbcast_b32(__builtin_bit_cast(unsigned int, v), g0));
}
// rows selected by ROWMASK <- lane LANE of the own row of src; the other rows keep old
template <int LANE, int ROWMASK>
__device__ __forceinline__ double lgp_row_lane(double old, double src) {
  const long long o = __builtin_bit_cast(long long, old), s = __builtin_bit_cast(long long, src);
  const unsigned int lo = __builtin_amdgcn_update_dpp((unsigned int)o, (unsigned int)s, 0x150 + LANE, ROWMASK, 0xf, false);
  const unsigned int hi = __builtin_amdgcn_update_dpp((unsigned int)(o >> 32), (unsigned int)(s >> 32), 0x150 + LANE, ROWMASK, 0xf, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | lo);
}

constexpr int SUBSTM_ROWS = 64;   // rows per 256-thread workgroup (16 per wave)

#ifdef LGP_SUBST_TRACE
__device__ unsigned long long lgp_subst_trace[16];
#endif
// One wave: 16 rows (r0 .. r0 + 15, those below `rows` only) of X <- X L_jj^-T, the block given by its fragment image.
template <typename T>
__device__ __forceinline__ void subst_right_wave(const T *__restrict__ w, int jb, T *X, long ldx, long rows, long r0) {
  typedef typename Mma<T>::acc_t acc_t;
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  if (r0 >= rows) return;     // wave-uniform
  const long row = r0 + n;
  const bool rok = row < rows;
  T *xrow = X + (rok ? row : r0) * ldx;     // clamped: out-of-range lanes read a valid row and never store
  const __amdgpu_buffer_rsrc_t frag = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(w + LGP_FRAG0), 0, LGP_NFRAG * 64 * (int)sizeof(T), 0x00020000);
  const int voff = lane * (int)sizeof(T);
  auto F = [&](int id) -> T { return Mma<T>::load(frag, voff, id * 64 * (int)sizeof(T)); };

  acc_t t[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = 16 * c + 4 * r + g;
      t[c][r] = (col < jb) ? xrow[col] : T(0);
    }

  // Right-looking over the eight 16-column tiles.  The dependent chain of a tile is its in-tile solve (four slabs of three
  // broadcast + fma steps and one MFMA) plus the four MFMAs that bring in the tile just finished; everything else -- the
  // updates of the later tiles with a finished tile, and above all the FRAGMENT LOADS (L2 round trips of several hundred
  // cycles each) -- is issued a whole tile step ahead of its use: the fragments of step c + 1 are requested at the top of
  // step c, into registers.
#ifdef LGP_SUBST_TRACE
#define LGP_SSTAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (threadIdx.x == 0 && r0 == 0) lgp_subst_trace[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LGP_SSTAMP(k) do { } while (0)
#endif
  int baddr[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) baddr[a] = (a * 16 + n) * 4;
  LGP_SSTAMP(0);
  T rdv[8][4];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) rdv[c][r] = F(232 + 4 * c + r);
  T coef[2][4][3], dg[2][3];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int a = 0; a < 3; ++a) coef[0][r][a] = F(136 + r * 3 + a);
    if (r < 3) dg[0][r] = F(112 + r);
  }
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) t[c][r] *= rdv[c][r];                       // x~ = x / diag, every tile up front
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int cur = c & 1, nxt = cur ^ 1;
    // requests for the next step: its in-tile coefficients, and the fragments that apply THIS tile to the later ones
    T off[7][4];
#pragma unroll
    for (int cc = c + 1; cc < 8; ++cc)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) off[cc - c - 1][s2] = F((cc * (cc - 1) / 2 + c) * 4 + s2);
    if (c < 7) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int a = 0; a < 3; ++a) coef[nxt][r][a] = F(136 + (4 * (c + 1) + r) * 3 + a);
        if (r < 3) dg[nxt][r] = F(112 + 3 * (c + 1) + r);
      }
    }
    // in-tile solve
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const T yb = Mma<T>::bpermute(baddr[a], t[c][r]);                 // unknown 4r + a of this row, to every lane group
                                                                          // (lane swaps instead -- lgp_rowbcast -- were measured slower here: 21.6 vs 19.6 us)
        t[c][r] = __builtin_fma(coef[cur][r][a], yb, t[c][r]);            // zero coefficient for groups <= a
      }
      if (r < 3) t[c] = Mma<T>::run(dg[cur][r], t[c][r], t[c]);           // finished unknowns -> later registers of the tile
    }
    // the finished tile -> every later tile (nearest first: it is the next one on the chain)
#pragma unroll
    for (int cc = c + 1; cc < 8; ++cc)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) t[cc] = Mma<T>::run(off[cc - c - 1][s2], t[c][s2], t[cc]);
    LGP_SSTAMP(1 + c);
  }

  if (rok) {
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = 16 * c + 4 * r + g;
        if (col < jb) xrow[col] = t[c][r];
      }
  }
  LGP_SSTAMP(9);
}

}  // namespace

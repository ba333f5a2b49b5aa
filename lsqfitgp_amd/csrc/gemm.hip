// gemm.hip -- LDS-tiled MFMA GEMM for gfx950 (CDNA4), fp64 (v_mfma_f64_16x16x4_f64) and fp32
// (v_mfma_f32_16x16x4_f32).  This one kernel family is the dense-contraction workhorse of the hot path:
//   * syrk/gemm trailing update of the blocked Cholesky (replaces the LAPACK potrf that
//     jlinalg.cholesky lowers to, src/lsqfitgp/_linalg/_decomp.py:388),
//   * the gemm updates of the blocked triangular solves (solve_triangular, _decomp.py:402-439),
//   * `invLA.T @ invLr`, `invLA.T @ invLA`, `L @ x`, `L.T @ X` (_decomp.py:409,420,431,435).
//
// Geometry: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA
// tiles, 16 accumulators x 4 values), K-step 16, double-buffered LDS (73.7 KB -> 2 workgroups per CU),
// register-staged global->LDS prefetch one K-step ahead, one barrier per K-step.  An fp64 MFMA occupies
// the SIMD for 64 cycles (measured, tools/f64_rate_probe.hip), so per K-step each wave has 4096 cycles
// of matrix work against 32 ds_read_b64 and 8 global loads: the kernel is MFMA-bound by design.
//
// LDS images are chosen per operand so that the staging store is a straight 16-byte copy and the
// fragment read (one element per lane: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15]) is
// bank-conflict free for ds_read_b64:
//   * operand stored k-contiguous in memory  -> image [row][BK + 2]   (stride 18 doubles = 36 dwords)
//   * operand stored row-contiguous in memory -> image [k][128 + 16]  (stride 144 doubles, 288 = 32 mod 64)
//
// Workgroup -> tile map: tiles are grouped in 8x8 super-tiles; super-tile s goes to the XCD that the
// dispatcher's round-robin gives to blockIdx % 8 == s % 8, so the 64 co-resident workgroups of an XCD
// share 8 A-panels and 8 B-panels through that XCD's L2 while they walk K together.  Placement only
// affects speed, never results.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

#ifndef LGP_GEMM_STORE_AT
#define LGP_GEMM_STORE_AT 3   // K-step quarter after which the next operand tiles are written to LDS (4: after the MFMAs;
                              // 3 measured +1.5 %: 70.5 vs 69.4 TFLOP/s at 30720 rows, 2 the same)
#endif
constexpr int BK = 16;        // K-step; the output tile is TS x TS with TS = 128 (bulk) or 64 (latency-bound small products)

template <typename T> struct Mfma;
template <> struct Mfma<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg (verified on device)
  static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
  static constexpr int SMK = BK + 2;   // [row][k] image stride (elements)
};
template <> struct Mfma<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D map of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
  static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
  static constexpr int SMK = BK + 4;
};
template <int TS> struct Skm { static constexpr int value = TS + 16; };   // [k][row] image stride: 2*stride = 32 mod 64

template <typename T>
struct GemmArgs {
  const T *A;
  const T *B;
  T *C;
  long lda, ldb, ldc;
  int m, n, k;
  T alpha, beta;
  int lower;
  // block-column-cyclic lower mode (lower != 0, cyc_nb > 0): local column block q of width cyc_nb holds the global
  // block (q0 + q) * G + rank, i.e. global column = local column + cyc_nb * cyc_g1 * (local column / cyc_nb), cyc_g1 = G - 1;
  // the B operand rows and the lower mask follow the global column (0, 0: plain lower)
  int cyc_nb, cyc_g1;
  // row offset of the lower mask (lower != 0): row i of this launch is row i + roff of the trapezoid, i.e. the launch produces
  // a horizontal slab of a larger lower update (the sharded factorisation updates its block columns slab by slab as the
  // pieces of a panel arrive, csrc/dist.hip); the operand pointers are the slab's own
  int roff;
  int tm, tn;          // tile counts
  int nsm, nsn;        // super-tile counts (0: direct mapping)
  int nbulk;           // super-tiles mapped XCD-wise (a multiple of 8); the rest is dealt out slot by slot
  int alignedA, alignedB;
  int prio;            // wave priority (s_setprio) of the launch: 0 leaves it alone
  int nsplit, kchunk;  // split-K: blockIdx.y = K slice, partial tiles go to Cpart[slice] (dense m x n), reduced afterwards
  T *Cpart;
#ifdef LGP_GEMM_TRACE
  unsigned long long *trace;   // tools/gemm_trace.hip: s_memtime stamps of wave 0 of each traced workgroup
  int noload;                  // experiment: skip the global loads of the main loop (timing only, wrong results)
#endif
};

#ifdef LGP_GEMM_TRACE
#define LGP_STAMP(slot)                                                                   \
  do {                                                                                    \
    if (g.trace && threadIdx.x == 0 && blockIdx.x < 8) {                                  \
      unsigned long long now_;                                                            \
      asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");         \
      g.trace[blockIdx.x * 1024 + (slot)] = now_;                                         \
    }                                                                                     \
  } while (0)
#else
#define LGP_STAMP(slot) do { } while (0)
#endif

// One operand tile: 128 "rows" (m or n index) x BK k-values.
// TR = false: memory is [row][k] (k contiguous);  TR = true: memory is [k][row] (row contiguous).
template <typename T, bool TR, int TS, int NTHREADS>
struct Tile {
  static constexpr int VEC = 16 / sizeof(T);
  static constexpr int NV = TS * BK / VEC / NTHREADS;
  static constexpr int SKM = Skm<TS>::value;
  static constexpr int IMG = TR ? BK * SKM : TS * Mfma<T>::SMK;   // elements per LDS image
  static_assert(NV >= 1, "tile too small for 256 threads");
  typedef T vec_t __attribute__((ext_vector_type(VEC)));

  vec_t reg[NV];
  unsigned goff[NV];   // fast path: byte offset of this thread's vectors from the (uniform) tile origin

  // Fast path (interior tile, 16-byte aligned operand, 128 * ld * sizeof(T) < 4 GB): the address is a uniform 64-bit
  // base that advances with k (scalar ALU) plus a per-thread 32-bit offset that never changes, i.e. the
  // global_load saddr + voffset form: no vector ALU work per K-step.  An fp64 MFMA stream does not share its SIMD
  // with other waves' vector instructions (measured with tools/gemm_trace.hip: every non-MFMA VALU instruction of
  // the loop shows up as idle matrix-core time), so the hot loop is kept free of them.
  __device__ __forceinline__ void init_fast(long ld) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = t + NTHREADS * i;
      if (!TR) {
        constexpr int VPR = BK / VEC;
        goff[i] = (unsigned)(((long)(v / VPR) * ld + (v % VPR) * VEC) * (long)sizeof(T));
      } else {
        constexpr int VPR = TS / VEC;
        goff[i] = (unsigned)(((long)(v / VPR) * ld + (v % VPR) * VEC) * (long)sizeof(T));
      }
    }
  }
  __device__ __forceinline__ void load_fast(const char *base) {
    // raw buffer load: scalar resource (base advanced with s_add) + per-thread 32-bit offset, no address VALU at all
    typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, -1, 0x00020000);
#pragma unroll
    for (int i = 0; i < NV; ++i) reg[i] = __builtin_bit_cast(vec_t, (u4_t)__builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[i], 0, 0));
  }

  __device__ __forceinline__ void load(const T *base, long ld, int row0, int k0, int R, int K, bool aligned) {
    const int t = threadIdx.x;
    if (!TR) {
      constexpr int VPR = BK / VEC;   // vectors per row
      const bool fast = aligned && (row0 + TS <= R) && (k0 + BK <= K);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int v = t + NTHREADS * i;
        const int row = v / VPR, kv = (v % VPR) * VEC;
        const T *p = base + (long)(row0 + row) * ld + (k0 + kv);
        if (fast) {
          reg[i] = *reinterpret_cast<const vec_t *>(p);
        } else {
          const bool rok = (row0 + row) < R;
#pragma unroll
          for (int e = 0; e < VEC; ++e) reg[i][e] = (rok && (k0 + kv + e) < K) ? p[e] : T(0);
        }
      }
    } else {
      constexpr int VPR = TS / VEC;  // vectors per k-row
      const bool fast = aligned && (row0 + TS <= R) && (k0 + BK <= K);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int v = t + NTHREADS * i;
        const int kr = v / VPR, rv = (v % VPR) * VEC;
        const T *p = base + (long)(k0 + kr) * ld + (row0 + rv);
        if (fast) {
          reg[i] = *reinterpret_cast<const vec_t *>(p);
        } else {
          const bool kok = (k0 + kr) < K;
#pragma unroll
          for (int e = 0; e < VEC; ++e) reg[i][e] = (kok && (row0 + rv + e) < R) ? p[e] : T(0);
        }
      }
    }
  }

  __device__ __forceinline__ void store(T *lds) const {
    const int t = threadIdx.x;
    if (!TR) {
      constexpr int VPR = BK / VEC;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int v = t + NTHREADS * i;
        const int row = v / VPR, kv = (v % VPR) * VEC;
        *reinterpret_cast<vec_t *>(lds + row * Mfma<T>::SMK + kv) = reg[i];
      }
    } else {
      constexpr int VPR = TS / VEC;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int v = t + NTHREADS * i;
        const int kr = v / VPR, rv = (v % VPR) * VEC;
        *reinterpret_cast<vec_t *>(lds + kr * SKM + rv) = reg[i];
      }
    }
  }

  // fragment element for MFMA: row index r (0..TS-1), k index kk (0..BK-1)
  // volatile: keeps each fragment a separate ds_read_b64.  Left alone, the compiler pairs them into
  // ds_read2_b64, whose banking (32 banks, 16-lane groups) turns the stride-18 image into a 2-way conflict
  // (measured: SQ_LDS_BANK_CONFLICT = 40 % of SQ_LDS_IDX_ACTIVE); ds_read_b64 (64 banks, 32-lane groups) is
  // conflict-free on both images.
  static __device__ __forceinline__ T frag(const T *lds, int r, int kk) {
    typedef const volatile T __attribute__((address_space(3))) * lds_ptr_t;   // explicit LDS address space: a
    const T *q = TR ? lds + kk * SKM + r : lds + r * Mfma<T>::SMK + kk;       // volatile generic access would
    return *(lds_ptr_t)(q);                                                   // become a flat_load
  }
};

__device__ __forceinline__ bool map_tile(int tm, int tn, int nsm, int nsn, int nbulk, int lower, int &ti, int &tj) {
  const int bid = blockIdx.x;
  if (nsm == 0) {   // direct map, small grids
    ti = bid / tn;
    tj = bid % tn;
    return !(lower && ti < tj);
  }
  // bulk: whole super-tiles per XCD, the same number on each.  The remaining (< 8) super-tiles are dealt out slot by
  // slot over all XCDs, so the last round of a launch is spread over the whole chip instead of leaving the XCDs
  // without a super-tile idle for a full tile time.
  int sidx, slot;
  if (bid < nbulk * 64) {
    const int xcd = bid & 7, local = bid >> 3;
    sidx = (local >> 6) * 8 + xcd;
    slot = local & 63;
  } else {
    const int t = bid - nbulk * 64;
    sidx = nbulk + (t >> 6);
    slot = t & 63;
  }
  int I, J;
  if (!lower) {
    if (sidx >= nsm * nsn) return false;
    I = sidx / nsn;
    J = sidx % nsn;
  } else {
    // column-major over J of super-tiles with I >= J
    int rem = sidx;
    J = 0;
    const int jmax = nsn < nsm ? nsn : nsm;
    while (J < jmax && rem >= nsm - J) {
      rem -= nsm - J;
      ++J;
    }
    if (J >= jmax) return false;
    I = J + rem;
  }
  ti = I * 8 + (slot >> 3);
  tj = J * 8 + (slot & 7);
  if (ti >= tm || tj >= tn) return false;
  return !(lower && ti < tj);
}

// NTHREADS = 256: 2 x 2 waves, 64 x 64 (TS = 128) or 32 x 32 (TS = 64) per wave.
// (NTHREADS = 512, 4 x 2 waves of 32 x 64 at 4 waves per SIMD, is supported by the code and was measured equal.)
// TAG only names the instantiation: TAG = 1 is the Cholesky trailing update, so that profilers list it as its own row.
template <typename T, bool TA, bool TB, int TS, int NTHREADS, int TAG = 0>
__global__ void __launch_bounds__(NTHREADS, 2) gemm_kernel(const GemmArgs<T> g) {
  constexpr int BM = TS, BN = TS;
  constexpr int WGM = NTHREADS / 128;   // wave grid WGM x 2
  constexpr int WTM = TS / WGM, WTN = TS / 2;          // per-wave output tile
  constexpr int NFM = WTM / 16, NFN = WTN / 16;        // MFMA tiles per wave in m and n
  // A operand: op(A) is m x k.  TA=false: A stored [m][k] -> Tile<T,false>; TA=true: A stored [k][m] -> Tile<T,true>
  // B operand: op(B) is k x n.  TB=false: B stored [k][n] -> Tile<T,true>;  TB=true: B stored [n][k] -> Tile<T,false>
  typedef Tile<T, TA, TS, NTHREADS> TileA;
  typedef Tile<T, !TB, TS, NTHREADS> TileB;
  typedef typename Mfma<T>::acc_t acc_t;
  constexpr int IMG_A = TileA::IMG, IMG_B = TileB::IMG;
  __shared__ __attribute__((aligned(16))) T lds[2 * IMG_A + 2 * IMG_B];

  if (g.prio) __builtin_amdgcn_s_setprio(3);
  int ti, tj;
  LGP_STAMP(0);
  if (!map_tile(g.tm, g.tn, g.nsm, g.nsn, g.nbulk, g.lower && g.roff == 0, ti, tj)) return;
  const int row0 = ti * BM, col0 = tj * BN;
  const int cshift = g.cyc_nb ? g.cyc_nb * g.cyc_g1 * (col0 / g.cyc_nb) : 0;   // global column - local column of this tile
  const int mshift = cshift - g.roff;                                          // ... as the lower mask sees it (row offset folded in)
  if (g.lower && row0 + BM - 1 < col0 + mshift) return;                        // entirely above the (global) diagonal

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * WTM, wn = (wave & 1) * WTN;
  const int fr = lane & 15, fk = lane >> 4;

  // Accumulate-in-place mode (C = C +- A.B, the Cholesky trailing update and the solve updates): the C tile is
  // loaded straight into the accumulators while the first operand tiles are in flight, so the epilogue is stores
  // only.  (A read-modify-write epilogue on the same pointer serialises into 64 dependent global round trips per
  // lane.)  alpha = -1 costs nothing in the loop: the accumulators start at -C and the epilogue stores -acc.
  const bool interior = (row0 + BM <= g.m) && (col0 + BN <= g.n) && !(g.lower && row0 < col0 + mshift + BN - 1);
  const bool accinit = g.nsplit <= 1 && g.beta == T(1) && (g.alpha == T(1) || g.alpha == T(-1));
  const bool negC = accinit && g.alpha == T(-1);

  acc_t acc[NFM][NFN];
#pragma unroll
  for (int i = 0; i < NFM; ++i)
#pragma unroll
    for (int j = 0; j < NFN; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
  TileA ta;
  TileB tb;
  // K range of this workgroup (split-K slices are multiples of BK, so vector loads stay aligned)
  const int kbeg = g.nsplit > 1 ? (int)blockIdx.y * g.kchunk : 0;
  const int kend = g.nsplit > 1 ? ((kbeg + g.kchunk) < g.k ? (kbeg + g.kchunk) : g.k) : g.k;
  const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  T *ldsA = lds, *ldsB = lds + 2 * IMG_A;

  // whole-tile fast path: both operand tiles inside the matrices, aligned, K range a multiple of BK
  const bool allfast = g.alignedA && g.alignedB && (row0 + BM <= g.m) && (col0 + BN <= g.n) && ((kend - kbeg) % BK == 0) &&
                       g.lda < (1l << 21) && g.ldb < (1l << 21);
  const int bcol0 = col0 + cshift;        // B operand rows of this tile (cyclic mode: the global column)
  const int bn = g.n + cshift;            // ... and their end
#ifdef LGP_GEMM_TRACE
  const int lrow0 = g.noload == 2 ? 0 : row0, lcol0 = g.noload == 2 ? 0 : bcol0;   // experiment: every tile reads the same operand tiles
#else
  const int lrow0 = row0, lcol0 = bcol0;
#endif
  const char *pa = reinterpret_cast<const char *>(TA ? g.A + (long)kbeg * g.lda + lrow0 : g.A + (long)lrow0 * g.lda + kbeg);
  const char *pb = reinterpret_cast<const char *>(!TB ? g.B + (long)kbeg * g.ldb + lcol0 : g.B + (long)lcol0 * g.ldb + kbeg);
  const long stepA = (TA ? (long)BK * g.lda : (long)BK) * (long)sizeof(T);
  const long stepB = (!TB ? (long)BK * g.ldb : (long)BK) * (long)sizeof(T);
  if (allfast) {
    ta.init_fast(g.lda);
    tb.init_fast(g.ldb);
  }

  if (nk > 0) {
    if (allfast) {
      ta.load_fast(pa);
      tb.load_fast(pb);
    } else {
      ta.load(g.A, g.lda, row0, kbeg, g.m, kend, g.alignedA);
      tb.load(g.B, g.ldb, bcol0, kbeg, bn, kend, g.alignedB);
    }
  }
  if (accinit) {
    if (interior) {   // uniform branch: 4 * NFM * NFN independent loads in flight, no per-element control flow
#pragma unroll
      for (int i = 0; i < NFM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int j = 0; j < NFN; ++j)
            acc[i][j][r] = g.C[(row0 + wm + i * 16 + Mfma<T>::crow(lane, r)) * g.ldc + (col0 + wn + j * 16 + fr)];
    } else {
#pragma unroll
      for (int i = 0; i < NFM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row0 + wm + i * 16 + Mfma<T>::crow(lane, r);
#pragma unroll
          for (int j = 0; j < NFN; ++j) {
            const int col = col0 + wn + j * 16 + fr;
            const bool ok = row < g.m && col < g.n && !(g.lower && col + mshift > row);
            const T v = g.C[ok ? row * g.ldc + col : 0];   // clamped address instead of a branch around the load
            acc[i][j][r] = ok ? v : T(0);
          }
        }
      }
    }
    if (negC) {
#pragma unroll
      for (int i = 0; i < NFM; ++i)
#pragma unroll
        for (int j = 0; j < NFN; ++j) acc[i][j] = -acc[i][j];
    }
  }
  if (nk > 0) {
    ta.store(ldsA);
    tb.store(ldsB);
  }
  __syncthreads();
  LGP_STAMP(1);

  // One K-step with a compile-time LDS buffer index (the loop is unrolled by two), so that every LDS address of
  // the step is a per-thread constant plus an immediate.
  auto kstep = [&](auto curc, int t) {
    constexpr int cur = decltype(curc)::value;
    LGP_STAMP(4 + 4 * t);
    const bool more = (t + 1) < nk;
#ifdef LGP_GEMM_TRACE
    if (more && g.noload != 1) {
#else
    if (more) {
#endif
      if (allfast) {
        pa += stepA;
        pb += stepB;
        ta.load_fast(pa);
        tb.load_fast(pb);
      } else {
        ta.load(g.A, g.lda, row0, kbeg + (t + 1) * BK, g.m, kend, g.alignedA);
        tb.load(g.B, g.ldb, bcol0, kbeg + (t + 1) * BK, bn, kend, g.alignedB);
      }
    }
    const T *la = ldsA + cur * IMG_A;
    const T *lb = ldsB + cur * IMG_B;
#pragma unroll
    for (int ks = 0; ks < BK; ks += 4) {
      T af[NFM], bf[NFN];
#pragma unroll
      for (int i = 0; i < NFM; ++i) af[i] = TileA::frag(la, wm + i * 16 + fr, ks + fk);
#pragma unroll
      for (int j = 0; j < NFN; ++j) bf[j] = TileB::frag(lb, wn + j * 16 + fr, ks + fk);
#pragma unroll
      for (int i = 0; i < NFM; ++i)
#pragma unroll
        for (int j = 0; j < NFN; ++j) acc[i][j] = Mfma<T>::run(af[i], bf[j], acc[i][j]);
      if (ks / 4 + 1 == LGP_GEMM_STORE_AT && LGP_GEMM_STORE_AT < BK / 4) {
        // staging stores inside the wave's own MFMA phase: while the partner workgroup streams MFMAs no
        // VGPR-reading instruction of this wave (the ds_write included) gets issued
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
          ta.store(ldsA + (cur ^ 1) * IMG_A);
          tb.store(ldsB + (cur ^ 1) * IMG_B);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    LGP_STAMP(5 + 4 * t);
    if (LGP_GEMM_STORE_AT >= BK / 4 && more) {
      ta.store(ldsA + (cur ^ 1) * IMG_A);
      tb.store(ldsB + (cur ^ 1) * IMG_B);
    }
    LGP_STAMP(6 + 4 * t);
    __syncthreads();
    LGP_STAMP(7 + 4 * t);
  };
  {
    int t = 0;
    for (; t + 2 <= nk; t += 2) {
      kstep(std::integral_constant<int, 0>{}, t);
      kstep(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < nk) kstep(std::integral_constant<int, 0>{}, t);
  }
  LGP_STAMP(2);

  // epilogue: C = alpha * acc + beta * C   (split-K: raw partial sums into this slice's slab)
  const bool split = g.nsplit > 1;
  const T alpha = split ? T(1) : (accinit ? (negC ? T(-1) : T(1)) : g.alpha), beta = (split || accinit) ? T(0) : g.beta;
  T *Cbase = split ? g.Cpart + (long)blockIdx.y * g.m * g.n : g.C;
  const long ldc = split ? (long)g.n : g.ldc;
  if (interior && beta == T(0)) {   // stores only (accumulate-in-place, split-K slabs, beta = 0)
#pragma unroll
    for (int i = 0; i < NFM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < NFN; ++j)
          Cbase[(row0 + wm + i * 16 + Mfma<T>::crow(lane, r)) * ldc + (col0 + wn + j * 16 + fr)] = alpha * acc[i][j][r];
    return;
  }
#pragma unroll
  for (int i = 0; i < NFM; ++i) {
    T cv[4][NFN];   // general beta: all loads of a 16-row band before its stores (same pointer, the compiler must keep order)
    if (beta != T(0)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + wm + i * 16 + Mfma<T>::crow(lane, r);
#pragma unroll
        for (int j = 0; j < NFN; ++j) {
          const int col = col0 + wn + j * 16 + fr;
          const bool ok = interior || (row < g.m && col < g.n && !(g.lower && col + mshift > row));
          cv[r][j] = Cbase[ok ? row * ldc + col : 0];   // clamped address instead of a branch around the load
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + wm + i * 16 + Mfma<T>::crow(lane, r);
#pragma unroll
      for (int j = 0; j < NFN; ++j) {
        const int col = col0 + wn + j * 16 + fr;
        if (interior || (row < g.m && col < g.n && !(g.lower && col + mshift > row))) {
          T v = alpha * acc[i][j][r];
          if (beta != T(0)) v += beta * cv[r][j];
          Cbase[row * ldc + col] = v;
        }
      }
    }
  }
}

// C = alpha * sum_s part[s] + beta * C
template <typename T>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const T *part, int nsplit, long mn, int n, T alpha, T beta, T *C, long ldc) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= mn) return;
  T acc = T(0);
  for (int s = 0; s < nsplit; ++s) acc += part[(long)s * mn + idx];
  T *p = C + (idx / n) * ldc + (idx % n);
  T v = alpha * acc;
  if (beta != T(0)) v += beta * (*p);
  *p = v;
}

}  // namespace

template <typename T, int TS, int NTHREADS>
static int gemm_launch_ts(lgp_ctx *ctx, hipStream_t stream, bool ta, bool tb, size_t m, size_t n, size_t k, T alpha, const T *A, size_t lda,
                          const T *B, size_t ldb, T beta, T *C, size_t ldc, bool lower, int tag, int cyc_nb, int cyc_g1, int roff) {
  constexpr size_t VEC = 16 / sizeof(T);
  GemmArgs<T> g;
  g.A = A; g.B = B; g.C = C;
  g.lda = (long)lda; g.ldb = (long)ldb; g.ldc = (long)ldc;
  g.m = (int)m; g.n = (int)n; g.k = (int)k;
  g.alpha = alpha; g.beta = beta;
  g.lower = lower ? 1 : 0;
  g.cyc_nb = cyc_nb;
  g.cyc_g1 = cyc_g1;
  g.roff = roff;
  g.prio = ctx->gemm_prio;
  g.tm = (int)((m + TS - 1) / TS);
  g.tn = (int)((n + TS - 1) / TS);
  g.alignedA = ((reinterpret_cast<uintptr_t>(A) % 16) == 0 && (lda % VEC) == 0) ? 1 : 0;
  g.alignedB = ((reinterpret_cast<uintptr_t>(B) % 16) == 0 && (ldb % VEC) == 0) ? 1 : 0;
  size_t grid;
  // super-tiles only pay when both tile dimensions are large; tall-skinny grids (panel updates, solves with a
  // few hundred right-hand sides) are 3x faster with the plain row-major map (measured: 30000x101x1024)
  if ((size_t)g.tm * g.tn <= 64 || g.tm < 16 || g.tn < 16) {
    g.nsm = g.nsn = 0;
    g.nbulk = 0;
    grid = (size_t)g.tm * g.tn;
  } else {
    g.nsm = (g.tm + 7) / 8;
    g.nsn = (g.tn + 7) / 8;
    size_t nsuper;
    if (!lower || roff != 0) {
      nsuper = (size_t)g.nsm * g.nsn;
    } else {
      const int jmax = g.nsn < g.nsm ? g.nsn : g.nsm;
      nsuper = 0;
      for (int J = 0; J < jmax; ++J) nsuper += (size_t)(g.nsm - J);
    }
    g.nbulk = (int)((nsuper / 8) * 8);
    grid = nsuper * 64;
  }
  LGP_REQUIRE(grid < (1ull << 31), "gemm grid too large");
  // split-K for few-tile, deep-K products (X^T X, X^T y over N rows): deterministic slab reduction
  g.nsplit = 1; g.kchunk = 0; g.Cpart = nullptr;
  const size_t tiles = (size_t)g.tm * g.tn;
  // (from k = 512 on for the tiniest outputs: the X'X and X'y of a posterior over a thousand points were two 59 us launches of 2-4
  // workgroups walking K = 1000 alone; in slabs of 128 they are ~15 us each, docs/userguide/optim.rst's N = 1000 benchmark)
  if (!lower && tiles * 4 <= (size_t)ctx->cus && (k >= 2048 || (k >= 512 && tiles <= 4))) {
    size_t ns = (2 * (size_t)ctx->cus) / tiles;
    const size_t maxs = k >= 2048 ? (k + 255) / 256 : (k + 127) / 128;
    if (ns > maxs) ns = maxs;
    if (ns > 1) {
      size_t kc = (k + ns - 1) / ns;
      kc = ((kc + BK - 1) / BK) * BK;
      ns = (k + kc - 1) / kc;
      g.nsplit = (int)ns; g.kchunk = (int)kc;
      LGP_CHECK(lgp_ws_alloc(ctx, stream, ns * m * n * sizeof(T), reinterpret_cast<void **>(&g.Cpart)));
    }
  }
  WsGuard cpart_guard(ctx, stream, g.Cpart);
  dim3 gd((unsigned)grid, (unsigned)g.nsplit), bd(NTHREADS);
  // untouched dynamic LDS that thins out the workgroups of a trailing update per CU (experiment, see potrf_chain)
  const size_t thin = (!ta && tb) ? ctx->gemm_thin_lds : 0;
  if (!ta && tb && tag == 1) hipLaunchKernelGGL((gemm_kernel<T, false, true, TS, NTHREADS, 1>), gd, bd, thin, stream, g);
  else if (!ta && tb) hipLaunchKernelGGL((gemm_kernel<T, false, true, TS, NTHREADS>), gd, bd, thin, stream, g);
  else if (!ta && !tb) hipLaunchKernelGGL((gemm_kernel<T, false, false, TS, NTHREADS>), gd, bd, 0, stream, g);
  else if (ta && !tb) hipLaunchKernelGGL((gemm_kernel<T, true, false, TS, NTHREADS>), gd, bd, 0, stream, g);
  else hipLaunchKernelGGL((gemm_kernel<T, true, true, TS, NTHREADS>), gd, bd, 0, stream, g);
  LGP_LAUNCH_CHECK();
  if (g.nsplit > 1) {
    const long mn = (long)m * (long)n;
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, stream, g.Cpart, g.nsplit, mn, (int)n,
                       alpha, beta, C, (long)ldc);
    LGP_LAUNCH_CHECK();
  }
  return LGP_OK;
}

size_t lgp_gemm_t64_trailing_max() {
  static const size_t v = getenv("LGP_GEMM_T64_TRAILING_MAX") ? (size_t)atol(getenv("LGP_GEMM_T64_TRAILING_MAX")) : 10000;
  return v;
}

template <typename T>
int lgp_gemm_launch(lgp_ctx *ctx, hipStream_t stream, bool ta, bool tb, size_t m, size_t n, size_t k, T alpha,
                    const T *A, size_t lda, const T *B, size_t ldb, T beta, T *C, size_t ldc, bool lower, int tag, int cyc_nb, int cyc_g1, int roff) {
  if (m == 0 || n == 0) return LGP_OK;
  LGP_REQUIRE(roff == 0 || (lower && roff > 0), "row offset: lower updates only");
  LGP_REQUIRE(cyc_nb == 0 || (lower && cyc_nb % 128 == 0 && cyc_g1 >= 0 && !ta && tb), "cyclic mode: lower NT update, block width a multiple of 128");
  LGP_REQUIRE(m < (1u << 30) && n < (1u << 30) && k < (1u << 30), "gemm dimension too large");
  LGP_REQUIRE(C != nullptr && (k == 0 || (A != nullptr && B != nullptr)), "null matrix pointer");
  LGP_REQUIRE(ldc >= n, "ldc < n");
  LGP_REQUIRE(k == 0 || lda >= (ta ? m : k), "lda too small");
  LGP_REQUIRE(k == 0 || ldb >= (tb ? k : n), "ldb too small");
  if (ctx->tracer != nullptr) {            // the sharded factorisation's schedule, written down (dist.hip); dry: nothing is launched
    long long sa, oa, sb, ob, sc, oc;
    lgp_tr_ptr(ctx, A, sa, oa); lgp_tr_ptr(ctx, B, sb, ob); lgp_tr_ptr(ctx, C, sc, oc);
    if (!LGP_TRACED(ctx, LGP_TR_GEMM, stream, (long long)m, (long long)n, (long long)k, sa, oa, (long long)lda, sb, ob, (long long)ldb, sc, oc, (long long)ldc,
                    lower ? 1 : 0, cyc_nb, cyc_g1, roff, (ta ? 1 : 0) | (tb ? 2 : 0)))
      return LGP_OK;
  }
  // Small outputs sit on the dependent chains of the panel factorisation and of the triangular solves: with
  // 128x128 tiles they would occupy a handful of CUs for 8+ K-steps of 4096 cycles each; 64x64 tiles quarter
  // the per-workgroup latency and spread the product over 4x more CUs.  In-place products (C aliasing an
  // operand) rely on one tile covering the aliased dimension, which the callers guarantee for 128 only.
  const size_t t128 = ((m + 127) / 128) * ((n + 127) / 128);
  const bool aliased = (static_cast<const void *>(A) == static_cast<const void *>(C)) || (static_cast<const void *>(B) == static_cast<const void *>(C));
  // tag & 2 (the updates of the factorisation, K = 1024): 64 x 64 tiles up to 10000 tiles of 128 as well.  Their workgroups
  // live 27 us instead of 200, so the latency-bound kernels of the other streams find CUs and are not squeezed for as long:
  // N = 8192 7.2 -> 6.8 ms, N = 16384 32.2 -> 30.4 ms, N = 32768 185.7 -> 183.7 ms (all 64-tile: 187.8 ms).
  const size_t t64big = (tag & 2) ? lgp_gemm_t64_trailing_max() : 64;
  // ... and for outputs of at most four 128-tiles whatever K (X^T X of the posterior: 101 x 101 over N rows): with split-K below
  // they become ~100 short workgroups instead of 8 long ones (N = 4096: 90 -> 15 us per product)
  // tag & 4 (the left-looking block-column updates of the chain driver, K up to N): 64 x 64 tiles while 128-tiles would not fill the GPU
  static const size_t deep64 = getenv("LGP_GEMM_DEEP64") ? (size_t)atol(getenv("LGP_GEMM_DEEP64")) : 512;
  static const bool force64 = getenv("LGP_GEMM_FORCE64") != nullptr;      // (measurements: every product that may on 64 x 64 tiles)
  if (force64 && !aliased) return gemm_launch_ts<T, 64, 256>(ctx, stream, ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, lower, 0, cyc_nb, cyc_g1, roff);
  if (!aliased && ((t128 <= t64big && k <= 2048) || (t128 <= 512 && k <= 256) || t128 <= 4 || ((tag & 4) && t128 <= deep64)))
    return gemm_launch_ts<T, 64, 256>(ctx, stream, ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, lower, 0, cyc_nb, cyc_g1, roff);
  // (a 512-thread variant, 4 x 2 waves of 32 x 64 at 4 waves per SIMD, was measured equal: 61.3 vs 61.7 TFLOP/s)
  return gemm_launch_ts<T, 128, 256>(ctx, stream, ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, lower, tag & 1, cyc_nb, cyc_g1, roff);
}

template int lgp_gemm_launch<double>(lgp_ctx *, hipStream_t, bool, bool, size_t, size_t, size_t, double, const double *, size_t,
                                     const double *, size_t, double, double *, size_t, bool, int, int, int, int);
template int lgp_gemm_launch<float>(lgp_ctx *, hipStream_t, bool, bool, size_t, size_t, size_t, float, const float *, size_t,
                                    const float *, size_t, float, float *, size_t, bool, int, int, int, int);

extern "C" int lgp_gemm_f64(lgp_ctx *ctx, int ta, int tb, size_t m, size_t n, size_t k, double alpha, const double *dA,
                            size_t lda, const double *dB, size_t ldb, double beta, double *dC, size_t ldc) {
  LGP_REQUIRE(ctx != nullptr, "null context");
  DeviceGuard dg(ctx->device);
  return lgp_gemm_launch<double>(ctx, ctx->stream, ta != 0, tb != 0, m, n, k, alpha, dA, lda, dB, ldb, beta, dC, ldc, false);
}

extern "C" int lgp_gemm_f32(lgp_ctx *ctx, int ta, int tb, size_t m, size_t n, size_t k, float alpha, const float *dA,
                            size_t lda, const float *dB, size_t ldb, float beta, float *dC, size_t ldc) {
  LGP_REQUIRE(ctx != nullptr, "null context");
  DeviceGuard dg(ctx->device);
  return lgp_gemm_launch<float>(ctx, ctx->stream, ta != 0, tb != 0, m, n, k, alpha, dA, lda, dB, ldb, beta, dC, ldc, false);
}

extern "C" int lgp_update_nt_cyclic_f64(lgp_ctx *ctx, size_t rows, size_t cols, size_t k, const double *dP, size_t ldp, double *dC, size_t ldc,
                                        size_t nb, size_t nranks) {
  LGP_REQUIRE(ctx != nullptr, "null context");
  LGP_REQUIRE(nb >= 128 && nb % 128 == 0 && nranks >= 1 && nb < (1u << 24) && nranks < 4096, "bad block layout");
  DeviceGuard dg(ctx->device);
  return lgp_gemm_launch<double>(ctx, ctx->stream, false, true, rows, cols, k, -1.0, dP, ldp, dP, ldp, 1.0, dC, ldc, true, 0, (int)nb,
                                 (int)(nranks - 1));
}

extern "C" int lgp_update_nt_slab_f64(lgp_ctx *ctx, size_t rows, size_t cols, size_t k, const double *dA, size_t lda, const double *dB, size_t ldb,
                                      double *dC, size_t ldc, size_t nb, size_t nranks, size_t roff) {
  LGP_REQUIRE(ctx != nullptr, "null context");
  LGP_REQUIRE(nb >= 128 && nb % 128 == 0 && nranks >= 1 && nb < (1u << 24) && nranks < 4096 && roff < (1u << 30), "bad block layout");
  DeviceGuard dg(ctx->device);
  return lgp_gemm_launch<double>(ctx, ctx->stream, false, true, rows, cols, k, -1.0, dA, lda, dB, ldb, 1.0, dC, ldc, true, 0, (int)nb,
                                 (int)(nranks - 1), (int)roff);
}

extern "C" int lgp_update_nt_f64(lgp_ctx *ctx, size_t rows, size_t cols, size_t k, const double *dA, size_t lda,
                                 const double *dB, size_t ldb, double *dC, size_t ldc, int lower) {
  LGP_REQUIRE(ctx != nullptr, "null context");
  DeviceGuard dg(ctx->device);
  return lgp_gemm_launch<double>(ctx, ctx->stream, false, true, rows, cols, k, -1.0, dA, lda, dB, ldb, 1.0, dC, ldc, lower != 0);
}

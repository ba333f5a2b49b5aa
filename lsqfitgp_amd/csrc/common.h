// Internal definitions shared by the HIP translation units of liblsqfitgp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include <type_traits>
#include <vector>
#include "../../include/lsqfitgp_hip.h"

struct lgp_tracer;

struct lgp_ctx {
  int device = 0;
  lgp_tracer *tracer = nullptr;       // schedule tracer / jitter of the sharded factorisation (dist.hip), normally absent
  hipStream_t stream = nullptr;       // main stream: all public entry points enqueue here
  hipStream_t stream2 = nullptr;      // look-ahead stream of the blocked factorisation
  hipStream_t stream3 = nullptr;      // far-row stream of the factorisation (rows below a panel's diagonal square)
  hipStream_t stream_comm = nullptr;  // the broadcasts of the sharded factorisation (dist.hip), created on first use
  std::vector<hipStream_t> far_streams;   // one per row group below the square (potrf_chain), high priority
  int prio_hi = 0;
  int chain_groups = 1;               // row groups = streams for the rows below a square (env LGP_CHAIN_GROUPS, 1..4)
  // CU-masked twins for chain-bound sizes (potrf.hip, potrf_split): the trailing updates keep off `reserve_chain +
  // reserve_far` CUs, the far rows off the first `reserve_chain`, so that the dependent chain always finds idle CUs
  hipStream_t stream_gm = nullptr, stream_fm = nullptr;
  int reserve_chain = 8, reserve_far = 8;      // env LGP_RESERVE="chain,far" (CUs; mask bit i is a CU of XCD i mod 8)
  size_t reserve_max_n = 0;                    // env LGP_RESERVE_MAX_N: largest matrix that reserves CUs (0: never, the default: measured slower)
  std::vector<hipEvent_t> panel_events;   // per 128-block of a panel: potf2 done, look-ahead rows solved, bulk done
  std::vector<hipEvent_t> dist_events;    // the events of the sharded factorisation's schedule (dist.hip), kept between calls: several hundred per call
  // lgp_dist_stats: the last sharded factorisation measured from inside -- timing event pairs around the compute stream's waits for
  // arriving pieces and around the packs -- off unless asked for (the timed steps of bench.py run without)
  int dist_stats_on = 0;
  std::vector<hipEvent_t> dist_timing_events;
  double dist_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  int *d_info = nullptr;              // pivot info word (0 = ok, else 1-based index), device
  void *d_scratch = nullptr;          // small reductions (>= 64 KiB)
  size_t scratch_bytes = 0;
  void *h_pinned = nullptr;           // pinned host staging for scalar results (4 KiB)
  int defer_finish = 0;               // lgp_chol_from_prepared leaves its read-back (eps, pivot verdict -> h_pinned + 32 / + 64 doubles), its synchronisation
                                      // and the diagonal-block images to the caller (lgp_predict_isotropic: one synchronisation per prediction)
  void *h_small = nullptr;            // pinned, device-mapped staging of the one-launch small prediction (1 MiB, created on first use; gram.hip)
  void *comm = nullptr;               // ncclComm_t
  // alternative transport: collectives done by a caller-supplied host function on a staged copy (lgp_comm_init_host)
  int (*hostfn)(void *user, int op, void *hbuf, size_t count, int root) = nullptr;
  void *hostuser = nullptr;
  int rank = 0, nranks = 1;
  // lgp_bcast: 0 = auto (RCCL: scatter + all-gather over the direct links for >= 3 ranks and >= bcast_min bytes; host transport:
  // plain), 1 = always the transport's own broadcast, 2 = scatter + all-gather whenever there are >= 3 ranks (the host transport
  // then needs op 5, see lgp_comm_init_host).  lgp_comm_set_bcast / env LGP_BCAST_ALGO=auto|plain|scatter, LGP_BCAST_MIN_BYTES
  int bcast_algo = 0;
  size_t bcast_min = (size_t)1 << 20;
  int cus = 256;
  int swapped = 0;                    // lgp_select_stream(1): `stream` and `stream2` exchanged
  int potf2_mode = 0;                 // diagonal-block kernel: 0 = auto, 1 = VALU, 2 = matrix cores (env LGP_POTF2=valu|mfma)
  int gemm_prio = 0;                  // wave priority of the next products (0: default; 3 for the latency-bound products on the chain's side, LGP_FAR_PRIO)
  size_t gemm_thin_lds = 0;           // bytes of untouched dynamic LDS added to NT gemm launches (set around the trailing updates)
  int potf2_late = 0;                 // potrf_split: the panel in hand belongs to the end of a large factorisation (its updates run on 64-tiles)
  int potf2_small = 1;                // auto mode: the factorisation in progress is small enough for the matrix-core kernel
  // the dependent chain of the factorisation as one launch of resident workers (potf2m.hip): schedules per matrix size,
  // flag/counter words (LGP_CH_* layout)
  struct ChainSched { unsigned long long key; void *tasks; int ntasks, nchain; };
  std::vector<ChainSched> chain_sched;
  void *chain_state = nullptr;
  size_t chain_state_bytes = 0;
  void *chain_trace = nullptr;        // LGP_CHAIN_TRACE=<file>: per-task times of the last launch
  size_t chain_trace_tasks = 0;
  const void *chain_trace_sched = nullptr;
  int chain_mode = 0;                 // 0 = not yet checked, 1 = off (env LGP_CHAIN=0, or the self-test failed), 2 = available
  size_t chain_max_n = 12288;         // largest matrix factored this way (env LGP_CHAIN_MAX_N; 14336 until round 5: N = 14336 20.3-20.6 ms through the split driver
                                      // with its chain tail against 20.75-20.84 here, 13312 17.13-17.25 against 17.30); beyond, the trailing update bounds the
                                      // factorisation and the workers' CUs are worth more to it (N = 32768: 186 ms split driver, 204 ms chain;
                                      // N = 16384: 30.2-30.5 ms here against 28.9-29.3 for the split driver with a chain tail of 8192 rows, N = 14336 equal)
  size_t chain_hog_max_n = 8192;      // ... and largest matrix for which the workers keep other workgroups off their CUs
  int chain_workers = 24;             // resident worker workgroups = CUs taken from the trailing update (env LGP_CHAIN_WORKERS); 16 until round 5:
                                      // same box N = 2048 1.15 -> 1.09 ms, 4096 2.31 -> 2.22, 8192 5.89 -> 5.78, 12288 14.4 -> 14.25, 14336 and C3 equal; 32 equal, 48 worse
  int chain_lds_kb = 100;             // untouched dynamic LDS per worker that keeps other workgroups off its CU (env LGP_CHAIN_LDS_KB)
  unsigned long long chain_timeout = 1000000000ull;   // wall_clock64 ticks (100 MHz) a worker waits for one input: 10 s (env LGP_CHAIN_TIMEOUT_MS)
  bool chain_attr_set = false;        // the chain kernel's dynamic-LDS limit has been raised on this context's device
  int chain_test_stall = 0;           // env LGP_CHAIN_TEST_STALL=1 (tests): the driver withholds u1[1], so that the workers time out
  // workspace blocks of the library's own (lgp_ws_alloc / lgp_ws_free): plain hipMalloc blocks kept for reuse, each tagged
  // with the stream that used it last and an event recorded there when it was given back
  struct WsBlock { void *ptr; size_t bytes; hipStream_t last; hipEvent_t ev; bool busy; };
  std::vector<WsBlock> ws_blocks;
  size_t ws_keep_bytes = (size_t)1 << 30;   // idle blocks beyond this total are returned to the device (env LGP_WS_KEEP_MB)
  // live kernel timing of the dominant kernel (trailing-update syrk) for bench.py: HIP event pairs on the
  // stream the kernel is launched on, enabled by lgp_profile_begin
  int prof_on = 0;
  std::vector<hipEvent_t> prof_events;   // start/stop pairs
  std::vector<double> prof_flops;        // algorithmic flops of each timed launch
  size_t prof_used = 0;
};

// ---- schedule tracer of the sharded factorisation (dist.hip) ------------------------------------------------------------
// What lgp_chol_factor_dist_f64 enqueues -- every event wait, event record, kernel launch, copy and broadcast with the stream it
// goes to and the operand ranges it was given -- written down as it is enqueued (LGP_DIST_TRACE=<path>: one file per rank), or
// instead of being enqueued (`dry`: lgp_chol_dist_trace, host only, no device), so that the CPU suite can check the schedule the
// C++ really produces: every pair of conflicting accesses ordered by stream order + event edges, and the factor it computes when
// a simulator runs it in any order those edges allow (tests/dist_trace.py).  `jitter` (LGP_DIST_JITTER=<seed>): a spin kernel of
// random 0-300 us in front of randomly chosen enqueues, so that an edge missing from the schedule shows within tens of runs.
constexpr int LGP_TR_FIELDS = 20;      // int64 words per record: kind, stream, 18 arguments
enum {
  LGP_TR_WAIT = 1,      // event
  LGP_TR_RECORD = 2,    // event
  LGP_TR_MARK = 3,      // what, b, i, r0, r1   (what: 0 lane, 1 comm, 2 bulk, 3 u1; context for the records that follow)
  LGP_TR_GEMM = 10,     // m, n, k, spA, offA, lda, spB, offB, ldb, spC, offC, ldc, lower, cyc_nb, cyc_g1, roff, ta | tb << 1
  LGP_TR_SUBST = 11,    // off(dblk workspace), jb, spX, offX, ldx, rows          X <- X L_jj^-T
  LGP_TR_POTF2 = 12,    // spA, offA, lda, nb, off(dblk workspace)
  LGP_TR_COPY2D = 13,   // spD, offD, ldd, spS, offS, lds, rows, cols
  LGP_TR_BCAST = 14,    // sp, off, elements, root
  LGP_TR_CHAINSQ = 15,  // spP, offP, ldl, kb, off(dblk workspace)               a square factored by the chain kernel
  LGP_TR_TAIL = 16,     // n, spL, offL, ldl, off(dblk), nx, row0                 the trailing matrix handed to the chain driver (one rank)
  LGP_TR_EXTRACT = 17,  // spL, offL, ldl, n, off(dblk workspace)
  LGP_TR_SCALE = 18,    // rows, cols, spA, offA, lda
  LGP_TR_MEMSET = 19,   // (the info word)
};
struct lgp_tracer {
  bool dry = false, log = false;
  unsigned long long jitter = 0;          // xorshift state, 0 = off
  long long slab = -1;                    // dry runs: slab height (-1: the library's default)
  int flags = -1;                         // dry runs: bit 0 owner sends through the communication stream, bit 1 chain kernel usable, bit 2 force comm
  std::vector<long long> rec;
  std::vector<const void *> streams, events;
  const char *base[3] = {nullptr, nullptr, nullptr};      // spaces: 0 the rank's block columns, 1 the panel slots, 2 the diagonal-block workspaces
  size_t bytes[3] = {0, 0, 0};
};
void lgp_tr_log(lgp_ctx *ctx, int kind, hipStream_t st, std::initializer_list<long long> args);
void lgp_tr_ptr(lgp_ctx *ctx, const void *p, long long &space, long long &off);       // (offset in doubles)
inline bool lgp_tr_dry(const lgp_ctx *ctx) { return ctx->tracer != nullptr && ctx->tracer->dry; }
int lgp_tr_wait(lgp_ctx *ctx, hipStream_t st, hipEvent_t ev);        // hipStreamWaitEvent, logged / jittered / skipped as the tracer says
int lgp_tr_record(lgp_ctx *ctx, hipEvent_t ev, hipStream_t st);      // hipEventRecord, likewise
// a traced operation: logs it (with jitter in front) and says whether it is to be enqueued at all
#define LGP_TRACED(ctx, kind, st, ...) ((ctx)->tracer == nullptr || (lgp_tr_log((ctx), (kind), (st), {__VA_ARGS__}), !(ctx)->tracer->dry))

int lgp_prof_start(lgp_ctx *ctx, hipStream_t stream, double flops);
int lgp_prof_stop(lgp_ctx *ctx, hipStream_t stream);

void lgp_set_error(const char *fmt, ...);

// Workspaces of the library (api.hip).  NOT hipMallocAsync: with the stream-ordered pool of this ROCm release a workspace of
// a few hundred MB that was freed and allocated again by the next call gave wrong results every other call (measured on
// lgp_mlnd_grad at n = 4096, round 3; the same code on hipMalloc blocks, zeroed or poisoned, is exact).  A block handed out for
// `stream` after it was last used on another stream makes `stream` wait for the event recorded when it was given back.
int lgp_ws_alloc(lgp_ctx *ctx, hipStream_t stream, size_t bytes, void **out);
int lgp_ws_free(lgp_ctx *ctx, hipStream_t stream, void *ptr);
// gives the block back when the scope is left on any path
struct WsGuard {
  lgp_ctx *c; hipStream_t s; void *p;
  WsGuard(lgp_ctx *ctx, hipStream_t stream, void *ptr) : c(ctx), s(stream), p(ptr) {}
  ~WsGuard() { if (p) (void)lgp_ws_free(c, s, p); }
  WsGuard(const WsGuard &) = delete;
  WsGuard &operator=(const WsGuard &) = delete;
};

#define LGP_HIP(expr)                                                                       \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      lgp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return LGP_EHIP;                                                                      \
    }                                                                                       \
  } while (0)

#define LGP_REQUIRE(cond, msg)                                        \
  do {                                                                \
    if (!(cond)) {                                                    \
      lgp_set_error("invalid argument: %s (%s)", msg, #cond);         \
      return LGP_EINVAL;                                              \
    }                                                                 \
  } while (0)

#define LGP_CHECK(expr)          \
  do {                           \
    int _s = (expr);             \
    if (_s != 0) return _s;      \
  } while (0)

#define LGP_LAUNCH_CHECK() LGP_HIP(hipGetLastError())

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (dev < 0) return;                // (a dry run of a schedule: no device, no runtime call)
    (void)hipGetDevice(&prev);
    if (prev != dev) (void)hipSetDevice(dev);
    else prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// ---- blocked-algorithm geometry -------------------------------------------------------
constexpr int LGP_DB = 128;    // diagonal block factored (and inverted) by one workgroup
constexpr int LGP_NB = 1024;    // outer panel width of the right-looking factorisation
constexpr int LGP_NFRAG = 264;   // MFMA operand fragments of one diagonal block (potrf.hip, pack_frag_kernel), 64 elements each
constexpr int LGP_FRAG0 = 2 * LGP_DB * LGP_DB + LGP_DB;   // offset of the fragment image inside a block's workspace
constexpr int LGP_DBLK = LGP_FRAG0 + LGP_NFRAG * 64;   // per-diagonal-block workspace: D, D^T, 1/diag, fragment image

// ---- GEMM (gemm.hip) ---------------------------------------------------------------------
// C (m x n) = alpha * op(A) op(B) + beta * C, row-major; ta/tb: operand stored transposed.
// lower != 0: only entries with row >= col are produced (C(0,0) is a diagonal element); tiles
// entirely above the diagonal are skipped.  C may alias A only when n <= 128 (one tile column),
// and may alias B only when m <= 128 (one tile row) and the aliased operand is read completely
// (k covers it) before the epilogue -- used by the in-place triangular kernels.
template <typename T>
int lgp_gemm_launch(lgp_ctx *ctx, hipStream_t stream, bool ta, bool tb, size_t m, size_t n, size_t k, T alpha,
                    const T *A, size_t lda, const T *B, size_t ldb, T beta, T *C, size_t ldc, bool lower, int tag = 0, int cyc_nb = 0,
                    int cyc_g1 = 0, int roff = 0);
// roff > 0 (lower only): the launch is the horizontal slab of a lower update that starts roff rows below its diagonal
// tag & 2: an update of the factorisation that shares the GPU with latency-bound kernels: 64 x 64 tiles up to lgp_gemm_t64_trailing_max()
// tiles of 128 x 128 (the tag & 1 label then only applies to the launches that still get 128 x 128 tiles)
size_t lgp_gemm_t64_trailing_max();
// cyc_nb > 0: block-column-cyclic lower update (see GemmArgs in gemm.hip); only for the NT lower form with A == B rows
// tag = 1 launches the separately named instantiation used for the Cholesky trailing update (profiling label only)

// ---- factorisation pieces (potrf.hip) ------------------------------------------------------
template <typename T>
int lgp_potf2_launch(lgp_ctx *ctx, hipStream_t stream, T *A, size_t lda, int nb, T *dblk, size_t row0);
int lgp_potf2_mfma_launch(lgp_ctx *ctx, hipStream_t stream, double *A, size_t lda, int nb, double *dblk, size_t row0);   // potf2m.hip
// the chain kernel (potf2m.hip): the diagonal squares of all panels, factored by resident workers
bool lgp_chain_usable(lgp_ctx *ctx, size_t n, size_t ldl);
size_t lgp_chain_state_bytes(size_t n);
int lgp_chain_state(lgp_ctx *ctx, size_t n, unsigned **out);        // device words, LGP_CH_* layout; the caller zeroes them
int lgp_chain_launch(lgp_ctx *ctx, hipStream_t stream, size_t n, double *L, size_t ldl, double *dblk, size_t row0 = 0);
// layout of the state words: a header, then LGP_CH_PANEL words per panel
constexpr int LGP_CH_NEXT = 0, LGP_CH_ABORT = 1, LGP_CH_NEXT_BULK = 2, LGP_CH_HEADER = 16, LGP_CH_PANEL = 512;
constexpr int LGP_CH_U1 = 0;      // stream -> kernel: the panel's square has the updates of all earlier panels
constexpr int LGP_CH_U2 = 1;      // stream -> stream: the block column below the NEXT panel's square updated with this panel
constexpr int LGP_CH_U3 = 2;      // stream -> stream: everything right of the next panel updated with this panel
constexpr int LGP_CH_GRP = 4;     // [4] stream -> stream: row group g of this panel's rows below its square is final
constexpr int LGP_CH_READY = 8;   // [8] kernel -> stream: 1 (block j factored) + the square's row units solved against it
constexpr int LGP_CH_FLAGP = 16;  // [8]
constexpr int LGP_CH_FLAGS = 32;  // [16][8] row unit r solved against block j
constexpr int LGP_CH_CNT = 160;   // [16][16] steps applied to tile (r, c)
// info word written when the workers of the chain kernel gave up waiting (a stall, not a pivot): reported as an error
constexpr int LGP_INFO_STALLED = 0x7fffffff;
int lgp_info_result(lgp_ctx *ctx, int info);      // info, or LGP_EHIP with the error text set for LGP_INFO_STALLED (potrf.hip)
template <typename T>
int lgp_panel_factor_impl(lgp_ctx *ctx, hipStream_t stream, size_t rows, size_t nb, T *P, size_t ldp, T *dblk, size_t row0);
template <typename T>
int lgp_potrf_impl(lgp_ctx *ctx, size_t n, T *L, size_t ldl, T *dblk, size_t nx = 0);
// nx: extra rows below the n x n matrix (right-hand sides B^T carried through the factorisation: on exit (L^-1 B)^T)
template <typename T>
int lgp_trsm_impl(lgp_ctx *ctx, hipStream_t stream, size_t n, const T *L, size_t ldl, const T *dblk, int trans, T *B, size_t ldb, size_t nrhs);

int lgp_extract_dblk_public(lgp_ctx *ctx, hipStream_t stream, const double *L, size_t ldl, size_t n, double *dblk);   // potrf.hip
// the whole prediction of a small problem in one launch (gram.hip); lgp_predict_isotropic_f64 takes it when it fits
bool lgp_predict_small_fits(int kernel, int d, size_t n, size_t m);
int lgp_predict_small_f64(lgp_ctx *ctx, int kernel, int p, int npar, const double *hpar, int d, const double *hpts, size_t n, size_t m, int has_loc, double loc,
                          int has_scale, double scale, int nops, const int *hops_code, const double *hops_val, const double *hy, const double *hnoise,
                          double *dP, double *dL, size_t ldl, size_t nx, double *ddblk, double *dnoise, double *hout, double *heps_out, int *info_out);

// ---- reductions (elementwise.hip) -------------------------------------------------------------
template <typename T>
int lgp_colsumsq_impl(lgp_ctx *ctx, hipStream_t stream, size_t rows, size_t cols, const T *A, size_t lda, T *out);

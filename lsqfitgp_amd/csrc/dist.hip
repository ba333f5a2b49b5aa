// dist.hip -- the 1-D block-column-cyclic Cholesky across the GPUs of one node, driven from the library: no host language
// between the panels (the reference has no distributed code).  One process per GPU; block column b
// (columns [b nb, (b+1) nb), all rows) lives on rank b % G.
//
// Round 3: the panels travel and are consumed in PIECES.  Panel b is cut by rows into
//     piece 0        its diagonal square, rows [b nb, (b+1) nb)
//     piece 1, 2 ..  the rows below in slabs that end at absolute multiples of `slab` rows (the carried right-hand sides
//                    ride with the last one);  one rank: a single piece with all the rows below
// because everything in a right-looking Cholesky depends on earlier columns only ROW-WISE: rows r of panel b+1 need rows r of
// panel b and the two squares, nothing else.  A slab therefore moves from panel to panel -- update of block column b+1 with
// the slab's rows of panel b, substitution against square b+1, broadcast -- independently of the slabs below it, and the
// dependent chain of the factorisation is no longer  "factor a whole (N - b nb) x nb panel, broadcast all of it, update a whole
// block column"  per panel but that of one slab; the slabs below follow as a pipeline (a wavefront over (panel, slab)).
// With whole panels and one panel of look-ahead (rounds 1-2) the chain was, for C4 on 8 GPUs, ~2.3 ms (block-column update)
// + ~3.5 ms (panel) + 3-11 ms (545 MB broadcast) at the first panel, falling only linearly with b, against 9 ms of trailing
// update per rank that fall quadratically: chain-bound after the first third of the panels.
//
// Every piece (b, i) is one unit of work for three kinds of stream, and ALL streams of ALL ranks take the pieces in the same
// order, by the key 2 b + i (then b) -- a topological order of the dependencies below, so nothing can wait on something
// that is queued behind it, on any stream, whatever the transport (the host-staged transport of the tests blocks the host):
//     owner(b), piece 0     [s1]  U:  square(b) -= P(b-1)[rows of b] P(b-1)[rows of b]'   needs recv(b-1, 1), bulk(b-2, .)
//                                 factor the square (chain kernel: resident workers, potf2m.hip); it is NOT sent: the updates
//                                 of the other block columns only read rows below a panel's diagonal block
//     owner(b), piece i>0   [s3]  (behind U2 of those rows, below) substitution against square b (left-looking over its 128-column blocks, each behind the
//                                 event of that block of the square), pack, ready(b, i)
//     owner(b), piece i>0   [s3]  ... and sent: its broadcast is issued right there, in stream order (see `comm`)
//     every other rank      [comm] broadcast piece (b, i) from owner(b) into panel slot b % nslots; recv(b, i)
//     owner(b+1)            [sc]  U2: the look-ahead column b + 1 below its square, rows of the piece -- in FRONT of:
//     every rank            [sc]  its block columns c >= b + 2, rows of the piece: one staircase MFMA launch (row offset =
//                                 the piece's first row below the diagonal of the first of them); bulk(b, i)
// The panel slots hold (rows x ldp) with a PADDED leading dimension (ldp = nb + 16: with ld = nb = 1024 every operand row of
// the update starts on one HBM channel, 62 instead of 68.7 TFLOP/s); a slot is reused for panel b + nslots once this rank's
// updates with panel b are through (free events of sc and of the look-ahead streams).
#include "common.h"
#include <algorithm>
#include <chrono>

template <typename T>
int lgp_far_rows_impl(lgp_ctx *ctx, hipStream_t st, size_t rows, size_t kb, T *Pfar, size_t ldl, const T *Psq, const T *Yk,
                      const hipEvent_t *blk_events);                                                                               // potrf.hip
int lgp_square_factor_f64(lgp_ctx *ctx, hipStream_t st, size_t kb, double *P, size_t ldl, double *Yk, size_t row0, bool use_chain,
                          const hipEvent_t *blk_events);                                                                           // potrf.hip
int lgp_potrf_chain_tail_f64(lgp_ctx *ctx, size_t n, double *L, size_t ldl, double *dblk, size_t nx, size_t row0);                 // potrf.hip

namespace {

struct Piece { size_t b, i, r0, r1; };

// default slab height (rows): one rank needs none; several: 8 panels' worth (env LGP_DIST_SLAB overrides)
size_t default_slab(size_t nb, size_t nranks) {
  if (const char *e = getenv("LGP_DIST_SLAB")) {
    const size_t v = (size_t)strtoull(e, nullptr, 10);
    return v == 0 ? 0 : ((v + nb - 1) / nb) * nb;
  }
  return nranks <= 1 ? 0 : 8 * nb;
}

// the pieces of panel b in row order
void panel_pieces(size_t n, size_t nb, size_t nx, size_t slab, size_t b, std::vector<Piece> &out) {
  out.clear();
  const size_t top = b * nb, sq_end = std::min(top + nb, n), end = n + nx;
  out.push_back({b, 0, top, sq_end});
  size_t r = sq_end;
  while (r < end) {
    size_t r1 = end;
    if (slab > 0) {
      const size_t cut = (r / slab + 1) * slab;          // next absolute multiple of the slab height
      if (cut < n) r1 = cut;                             // (the last slab takes the ragged end and the carried rows)
    }
    out.push_back({b, out.size(), r, r1});
    r = r1;
  }
}

void build_plan(size_t n, size_t nb, size_t nx, size_t slab, std::vector<Piece> &plan, size_t &maxpieces) {
  const size_t nblocks = (n + nb - 1) / nb;
  plan.clear();
  maxpieces = 0;
  std::vector<Piece> tmp;
  for (size_t b = 0; b < nblocks; ++b) {
    panel_pieces(n, nb, nx, slab, b, tmp);
    maxpieces = std::max(maxpieces, tmp.size());
    plan.insert(plan.end(), tmp.begin(), tmp.end());
  }
  std::stable_sort(plan.begin(), plan.end(), [](const Piece &x, const Piece &y) {
    const size_t kx = 2 * x.b + x.i, ky = 2 * y.b + y.i;
    return kx != ky ? kx < ky : x.b < y.b;
  });
}

size_t plan_slots(size_t maxpieces) { return maxpieces / 2 + 2; }

// the events of one schedule: taken from the context's store (a C3-sized call uses ~400: creating and destroying them every call
// was 2-3 ms of host time in front of and behind the factorisation), which only grows
struct EventPool {
  std::vector<hipEvent_t> &store;
  size_t used = 0;
  explicit EventPool(std::vector<hipEvent_t> &s) : store(s) {}
  bool dry = false;
  hipEvent_t get() {
    if (dry) return reinterpret_cast<hipEvent_t>((uintptr_t)0x1000 + 16 * used++);       // (never handed to the runtime)
    if (used < store.size()) return store[used++];
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    store.push_back(e);
    ++used;
    return e;
  }
};

// the context's collectives enqueue on ctx->stream: they go on the communication stream by swapping it in for the call
struct OnStream {
  lgp_ctx *c;
  hipStream_t saved;
  OnStream(lgp_ctx *ctx, hipStream_t s) : c(ctx), saved(ctx->stream) { c->stream = s; }
  ~OnStream() { c->stream = saved; }
};

// a few hundred microseconds of nothing on a stream (LGP_DIST_JITTER): every wave leaves after `ticks` of the 100 MHz clock
__global__ void jitter_spin_kernel(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// the tracer of one call: made from the environment unless the caller (a dry run) brought its own
struct TracerScope {
  lgp_ctx *ctx;
  lgp_tracer own;
  bool mine = false;
  const char *path = nullptr;
  explicit TracerScope(lgp_ctx *c) : ctx(c) {
    if (ctx->tracer != nullptr) return;
    path = getenv("LGP_DIST_TRACE");
    const char *jit = getenv("LGP_DIST_JITTER");
    if ((path == nullptr || !*path) && (jit == nullptr || !*jit)) return;
    own.log = path != nullptr && *path;
    if (jit != nullptr && *jit) {
      static unsigned long long calls = 0;        // (another sequence for every call of a process, and for every rank)
      own.jitter = (strtoull(jit, nullptr, 10) + 1) * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull * (unsigned long long)(ctx->rank + 1) + 0x632BE59BD9B4E019ull * ++calls;
      if (own.jitter == 0) own.jitter = 1;
    }
    ctx->tracer = &own;
    mine = true;
  }
  void dump(int status) {
    if (!mine || !own.log) return;
    char name[1024];
    snprintf(name, sizeof name, "%s.rank%d", path, ctx->rank);
    if (FILE *f = fopen(name, "wb")) {
      const long long head[4] = {0x3143525450474cll /* "LGPTRC1" */, LGP_TR_FIELDS, (long long)(own.rec.size() / LGP_TR_FIELDS), status};
      fwrite(head, sizeof head, 1, f);
      fwrite(own.rec.data(), sizeof(long long), own.rec.size(), f);
      fclose(f);
    }
  }
  ~TracerScope() {
    if (mine) ctx->tracer = nullptr;
  }
};

}  // namespace

void lgp_tr_ptr(lgp_ctx *ctx, const void *p, long long &space, long long &off) {
  const lgp_tracer *tr = ctx->tracer;
  const char *q = static_cast<const char *>(p);
  space = -1;
  off = 0;
  if (tr == nullptr) return;
  for (int sp = 0; sp < 3; ++sp)
    if (tr->base[sp] != nullptr && q >= tr->base[sp] && q < tr->base[sp] + tr->bytes[sp]) {
      space = sp;
      off = (long long)((q - tr->base[sp]) / (long long)sizeof(double));
      return;
    }
}

static long long tr_stream_id(lgp_tracer *tr, hipStream_t st) {
  for (size_t i = 0; i < tr->streams.size(); ++i)
    if (tr->streams[i] == static_cast<const void *>(st)) return (long long)i;
  tr->streams.push_back(st);
  return (long long)tr->streams.size() - 1;
}

void lgp_tr_log(lgp_ctx *ctx, int kind, hipStream_t st, std::initializer_list<long long> args) {
  lgp_tracer *tr = ctx->tracer;
  if (tr == nullptr) return;
  if (tr->jitter != 0 && !tr->dry && kind != LGP_TR_MARK) {
    unsigned long long x = tr->jitter;                  // xorshift64
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    tr->jitter = x ? x : 1;
    if ((x >> 60) < 6) {                                // 3 enqueues in 8: up to 300 us on this stream first
      const unsigned long long ticks = (x >> 20) % 30000ull;
      hipLaunchKernelGGL(jitter_spin_kernel, dim3(1), dim3(64), 0, st, ticks);
      (void)hipGetLastError();
    }
  }
  if (!tr->log && !tr->dry) return;
  const size_t at = tr->rec.size();
  tr->rec.resize(at + LGP_TR_FIELDS, 0);
  tr->rec[at] = kind;
  tr->rec[at + 1] = tr_stream_id(tr, st);
  size_t k = 2;
  for (long long a : args)
    if (k < (size_t)LGP_TR_FIELDS) tr->rec[at + k++] = a;
}

static long long tr_event_id(lgp_tracer *tr, hipEvent_t ev) {
  for (size_t i = 0; i < tr->events.size(); ++i)
    if (tr->events[i] == static_cast<const void *>(ev)) return (long long)i;
  tr->events.push_back(ev);
  return (long long)tr->events.size() - 1;
}

int lgp_tr_wait(lgp_ctx *ctx, hipStream_t st, hipEvent_t ev) {
  if (ctx->tracer != nullptr) {
    lgp_tr_log(ctx, LGP_TR_WAIT, st, {tr_event_id(ctx->tracer, ev)});
    if (ctx->tracer->dry) return LGP_OK;
  }
  LGP_HIP(hipStreamWaitEvent(st, ev, 0));
  return LGP_OK;
}

int lgp_tr_record(lgp_ctx *ctx, hipEvent_t ev, hipStream_t st) {
  if (ctx->tracer != nullptr) {
    lgp_tr_log(ctx, LGP_TR_RECORD, st, {tr_event_id(ctx->tracer, ev)});
    if (ctx->tracer->dry) return LGP_OK;
  }
  LGP_HIP(hipEventRecord(ev, st));
  return LGP_OK;
}

// The schedule of lgp_chol_factor_dist_f64 for one rank WITHOUT a device: the very function runs with a tracer that writes every
// enqueue down instead of performing it (fake stream / event handles, operand addresses inside made-up ranges).  `rec` receives
// LGP_TR_FIELDS int64 words per record (layout in csrc/common.h, decoded by tests/dist_trace.py); returns the number of records,
// or a negative status.  slab: (size_t)-1 = the default for `nranks`; flags: bit 0 the owner hands its pieces to the communication
// stream (the default with several ranks), bit 1 the squares go to the chain kernel where the library would use it, bit 2 the
// broadcasts are issued with one rank too; -1 = the library's defaults for `nranks`.  Host only.
extern "C" long lgp_chol_dist_trace(size_t n, size_t nb, size_t nx, int rank, int nranks, size_t slab, int flags, long long *rec, size_t capacity) {
  if (n == 0 || nb == 0 || nranks < 1 || rank < 0 || rank >= nranks) {
    lgp_set_error("invalid argument: empty problem or bad rank");
    return LGP_EINVAL;
  }
  lgp_ctx fake;
  lgp_tracer tr;
  tr.dry = true;
  tr.log = true;
  tr.slab = slab == (size_t)-1 ? -1 : (long long)(((slab + nb - 1) / nb) * nb);
  tr.flags = flags < 0 ? (nranks > 1 ? 1 : 0) : flags;
  fake.tracer = &tr;
  fake.device = -1;
  fake.rank = rank;
  fake.nranks = nranks;
  fake.stream = reinterpret_cast<hipStream_t>((uintptr_t)0x10);
  fake.stream2 = reinterpret_cast<hipStream_t>((uintptr_t)0x20);
  fake.stream3 = reinterpret_cast<hipStream_t>((uintptr_t)0x30);
  fake.stream_comm = reinterpret_cast<hipStream_t>((uintptr_t)0x40);
  const size_t G = (size_t)nranks, nblocks = (n + nb - 1) / nb, nlocal = (nblocks + G - 1 - (size_t)rank) / G;
  const size_t ldl = std::max<size_t>(nlocal, 1) * nb, ldp = nb + 16;
  const size_t use_slab = tr.slab >= 0 ? (size_t)tr.slab : default_slab(nb, G);
  std::vector<Piece> plan;
  size_t maxp = 0;
  build_plan(n, nb, nx, use_slab, plan, maxp);
  const size_t nslots = plan_slots(maxp);
  // made-up, disjoint address ranges (never dereferenced)
  double *L = reinterpret_cast<double *>((uintptr_t)1 << 40), *P = reinterpret_cast<double *>((uintptr_t)2 << 40), *W = reinterpret_cast<double *>((uintptr_t)3 << 40);
  int info = 0;
  const int st = lgp_chol_factor_dist_f64(&fake, n, nb, nx, L, ldl, W, P, ldp, nslots, nullptr, &info);
  fake.tracer = nullptr;
  fake.stream = fake.stream2 = fake.stream3 = fake.stream_comm = nullptr;
  if (st != LGP_OK) return st < 0 ? st : LGP_EINVAL;
  const size_t count = tr.rec.size() / LGP_TR_FIELDS;
  if (rec != nullptr)
    for (size_t t = 0; t < tr.rec.size() && t < capacity * LGP_TR_FIELDS; ++t) rec[t] = tr.rec[t];
  return (long)count;
}

// The last lgp_chol_factor_dist_f64 of this context measured from inside.  enable: 1 / 0 switch the measurement on / off for the
// calls that follow (timing event pairs on the compute stream and around the packs: not for timed runs), -1 leaves it.
// out (8 values, may be NULL): wall ms of the call, ms the compute stream waited for pieces, ms of packs, panel bytes read
// from other ranks, panel bytes of its own, pieces, broadcast algorithm (0 none, 1 the transport's broadcast, 2 scatter +
// all-gather), 1 if own pieces were sent from the communication stream.
extern "C" int lgp_dist_stats(lgp_ctx *ctx, int enable, double *out) {
  LGP_REQUIRE(ctx != nullptr, "null context");
  if (enable >= 0) ctx->dist_stats_on = enable != 0;
  if (out != nullptr) memcpy(out, ctx->dist_stats, sizeof ctx->dist_stats);
  return LGP_OK;
}

// The plan of the sharded factorisation, for callers that allocate its panel slots and for the CPU restatement of the
// schedule (tests/test_dist_gloo.py executes this very list over numpy + gloo).  Host only.
extern "C" long lgp_chol_dist_plan(size_t n, size_t nb, size_t nx, size_t nranks, size_t slab, unsigned long long *pieces, size_t capacity,
                                   size_t *nslots, size_t *slab_out) {
  if (n == 0 || nb == 0 || nranks == 0) {
    lgp_set_error("invalid argument: empty problem");
    return LGP_EINVAL;
  }
  if (slab == (size_t)-1) slab = default_slab(nb, nranks);
  else if (slab > 0) slab = ((slab + nb - 1) / nb) * nb;
  std::vector<Piece> plan;
  size_t maxp = 0;
  build_plan(n, nb, nx, slab, plan, maxp);
  if (pieces != nullptr)
    for (size_t t = 0; t < plan.size() && t < capacity; ++t) {
      pieces[4 * t] = plan[t].b; pieces[4 * t + 1] = plan[t].i; pieces[4 * t + 2] = plan[t].r0; pieces[4 * t + 3] = plan[t].r1;
    }
  if (nslots) *nslots = plan_slots(maxp);
  if (slab_out) *slab_out = slab;
  return (long)plan.size();
}

extern "C" int lgp_chol_factor_dist_f64(lgp_ctx *ctx, size_t n, size_t nb, size_t nx, double *dLloc, size_t ldl, double *ddblk,
                                        double *dpanels, size_t ldp, size_t nslots, const double *dsrow, int *hinfo) {
  LGP_REQUIRE(ctx != nullptr && dLloc != nullptr && ddblk != nullptr, "null pointer");
  LGP_REQUIRE(dpanels != nullptr || ctx->nranks == 1, "null pointer: panel slots (only a single rank goes without)");
  LGP_REQUIRE(n >= 1 && nb >= 1 && nb <= n + nb, "bad sizes");
  LGP_REQUIRE(ldp >= nb, "ldp < nb");
  const size_t G = (size_t)ctx->nranks, me = (size_t)ctx->rank;
  const size_t nblocks = (n + nb - 1) / nb;
  const size_t nlocal = (nblocks + G - 1 - me) / G;                   // blocks b with b % G == me
  LGP_REQUIRE(nlocal == 0 || ldl >= nlocal * nb, "ldl smaller than the local block columns");
  const bool dry = lgp_tr_dry(ctx);
  DeviceGuard dg(dry ? -1 : ctx->device);
  // LGP_DIST_TRACE=<path> / LGP_DIST_JITTER=<seed>: the schedule written down as it is enqueued / perturbed (see common.h)
  TracerScope tracer_scope(ctx);
  lgp_tracer *const tr = ctx->tracer;
  const size_t slab = (tr != nullptr && tr->slab >= 0) ? (size_t)tr->slab : default_slab(nb, G);
  std::vector<Piece> plan;
  size_t maxp = 0;
  build_plan(n, nb, nx, slab, plan, maxp);
  LGP_REQUIRE(nslots >= plan_slots(maxp), "too few panel slots: ask lgp_chol_dist_plan");
  static const bool force_comm_env = getenv("LGP_DIST_FORCE_COMM") != nullptr;      // (experiments: the communication stream's traffic with one rank)
  // Who issues the broadcast of a rank's own piece.  "order": the stream that solved and packed it (s3), in stream order -- measured
  // with ONE rank (LGP_DIST_FORCE_COMM, nothing is sent) to save the 5.5 ms per C3 step that the event hand-over costs.  "stream": the
  // communication stream behind an event.  With real peers the in-order send puts the root's collective on the solve lane (the
  // solve of the next piece waits until every peer has launched its receive) and makes one communicator live on two streams:
  // until a multi-GPU run has shown what that costs, several ranks hand over to the communication stream.
  static const char *owner_env = getenv("LGP_DIST_OWNER_COMM");
  bool force_comm = force_comm_env;
  bool owner_sends_in_order = owner_env != nullptr ? strcmp(owner_env, "stream") != 0 : G <= 1;
  if (tr != nullptr && tr->flags >= 0) {
    owner_sends_in_order = (tr->flags & 1) == 0;
    force_comm = (tr->flags & 4) != 0;
  }
  LGP_REQUIRE(dpanels != nullptr || !force_comm, "LGP_DIST_FORCE_COMM needs the panel slots");
  LGP_REQUIRE(ctx->stream2 != nullptr && ctx->stream3 != nullptr, "context without auxiliary streams");
  if (ctx->stream_comm == nullptr) LGP_HIP(hipStreamCreateWithPriority(&ctx->stream_comm, hipStreamNonBlocking, ctx->prio_hi));
  hipStream_t sc = ctx->stream, s1 = ctx->stream2, s3 = ctx->stream3, sx = ctx->stream_comm;
  const size_t rows_all = n + nx;
  const size_t dblk_per_block = lgp_chol_dblk_size(nb);
  ctx->potf2_small = n <= 8192;
  // whether the squares go to the chain kernel is settled HERE: the first question of a context runs a self-test that
  // synchronises the device, which must not happen between the broadcasts of a schedule that the peers are already inside
  const bool chain_ok = nb >= 256 && (dry ? (tr->flags >= 0 && (tr->flags & 2) != 0) : lgp_chain_usable(ctx, nb, ldl));
  // Without peers the end of the factorisation is what it is on the single-GPU driver (potrf_split): once this few rows are left
  // the dependent chain of squares bounds the rest, and the trailing matrix goes to the chain driver in one piece -- the last
  // eight panels of C3 in 6.3 ms instead of 10.4 (kernel trace, round 4).  With peers the trailing matrix is spread over them.
  size_t chain_tail = 0;
  if (G == 1 && slab == 0 && n % LGP_DB == 0 && nb % LGP_DB == 0 && lgp_chol_dblk_size(nb) == (nb / LGP_DB) * (size_t)LGP_DBLK) {
    static const size_t tail_env = getenv("LGP_SPLIT_CHAIN_TAIL") ? (size_t)atol(getenv("LGP_SPLIT_CHAIN_TAIL")) : (size_t)8192;
    const size_t cand = (tail_env < ctx->chain_max_n ? tail_env : ctx->chain_max_n) / LGP_NB * LGP_NB;
    if (cand >= 2048 && n > cand && (dry ? chain_ok : lgp_chain_usable(ctx, cand, ldl))) chain_tail = cand;
  }

  auto owner = [&](size_t b) { return b % G; };
  auto width = [&](size_t b) { return (n - b * nb) < nb ? (n - b * nb) : nb; };
  auto block = [&](size_t b) { return dLloc + (b * nb) * ldl + (b / G) * nb; };   // local block column b from its diagonal down
  auto dblk = [&](size_t b) { return ddblk + (b / G) * dblk_per_block; };
  auto slot_of = [&](size_t b) { return dpanels + (b % nslots) * rows_all * ldp; };
  auto panel_rows = [&](size_t b, size_t r) { return slot_of(b) + (r - b * nb) * ldp; };      // row r (absolute) of panel b in its slot

  // piece tables: events per piece, the pieces of every panel in row order
  std::vector<std::vector<Piece>> pieces(nblocks);
  for (size_t b = 0; b < nblocks; ++b) panel_pieces(n, nb, nx, slab, b, pieces[b]);
  auto cover = [&](size_t b, size_t r) -> size_t {      // the piece of panel b that holds row r
    const auto &v = pieces[b];
    for (size_t i = 0; i < v.size(); ++i)
      if (r >= v[i].r0 && r < v[i].r1) return i;
    return v.size() - 1;
  };
  EventPool pool(ctx->dist_events);
  pool.dry = dry;
  // lgp_dist_stats: pairs of timing events (wait_pairs around the compute stream's waits for a piece, pack_pairs around the packs)
  const bool stats = ctx->dist_stats_on != 0 && !dry;
  std::vector<hipEvent_t> wait_pairs, pack_pairs;
  size_t timing_used = 0;
  double bytes_recv = 0, bytes_own = 0;
  auto stamp = [&](hipStream_t st, std::vector<hipEvent_t> &into) {
    if (timing_used == ctx->dist_timing_events.size()) {
      hipEvent_t e = nullptr;
      if (hipEventCreate(&e) != hipSuccess) return;
      ctx->dist_timing_events.push_back(e);
    }
    hipEvent_t e = ctx->dist_timing_events[timing_used++];
    if (hipEventRecord(e, st) == hipSuccess) into.push_back(e);
  };
  const auto wall0 = std::chrono::steady_clock::now();
  if (tr != nullptr) {
    tr->base[0] = reinterpret_cast<const char *>(dLloc); tr->bytes[0] = rows_all * ldl * sizeof(double);
    tr->base[1] = reinterpret_cast<const char *>(dpanels); tr->bytes[1] = dpanels ? nslots * rows_all * ldp * sizeof(double) : 0;
    tr->base[2] = reinterpret_cast<const char *>(ddblk); tr->bytes[2] = std::max<size_t>(nlocal, 1) * dblk_per_block * sizeof(double);
    // the streams by name: 0 compute (sc), 1 squares (s1), 2 rows below them (s3), 3 communication (sx)
    tr->streams = {sc, s1, s3, sx};
    tr->events.clear();
    if (tr->log) lgp_tr_log(ctx, LGP_TR_MARK, sc, {-1, (long long)n, (long long)nb, (long long)nx, (long long)me, (long long)G, (long long)slab, (long long)nslots,
                                                    (long long)ldl, (long long)ldp, (long long)dblk_per_block, owner_sends_in_order ? 0 : 1});
  }
  std::vector<std::vector<hipEvent_t>> ev_recv(nblocks), ev_ready(nblocks), ev_bulk(nblocks), ev_u2(nblocks), ev_solved(nblocks);
  std::vector<hipEvent_t> ev_free_sc(nslots, nullptr), ev_free_l1(nslots, nullptr);
  std::vector<std::vector<hipEvent_t>> ev_blk(nblocks);       // per 128-column block of an owned square: factored, the square's rows solved
  bool ok = true;
  auto make = [&](hipEvent_t &e) { e = pool.get(); ok = ok && e != nullptr; };
  for (size_t b = 0; b < nblocks; ++b) {
    const size_t np = pieces[b].size();
    ev_recv[b].resize(np); ev_bulk[b].resize(np);
    for (size_t i = 0; i < np; ++i) { make(ev_recv[b][i]); make(ev_bulk[b][i]); }
    if (b + 1 < nblocks && owner(b + 1) == me) {
      ev_u2[b].resize(np);
      for (size_t i = 0; i < np; ++i) make(ev_u2[b][i]);
    }
    if (owner(b) == me) {
      ev_ready[b].resize(np); ev_solved[b].resize(np);
      for (size_t i = 0; i < np; ++i) { make(ev_ready[b][i]); make(ev_solved[b][i]); }
      ev_blk[b].resize((width(b) + LGP_DB - 1) / LGP_DB);
      for (auto &e : ev_blk[b]) make(e);
    }
  }
  for (size_t s = 0; s < nslots; ++s) { make(ev_free_sc[s]); make(ev_free_l1[s]); }
  hipEvent_t ev_start = nullptr, ev_end1 = nullptr, ev_end3 = nullptr, ev_endx = nullptr;
  make(ev_start); make(ev_end1); make(ev_end3); make(ev_endx);
  // From here on the peers may already be inside the broadcasts of this plan: a failure of this rank -- events, the first
  // enqueues, anything in the middle -- still issues every broadcast of the plan (contents no longer matter) before it is reported
  auto bare_broadcasts = [&](size_t from) {
    if (G <= 1) return;
    OnStream on(ctx, sx);
    for (size_t t = from; t < plan.size(); ++t)
      if (plan[t].i > 0 && !dry) (void)lgp_bcast(ctx, panel_rows(plan[t].b, plan[t].r0), (plan[t].r1 - plan[t].r0) * ldp * sizeof(double), (int)owner(plan[t].b));
    if (!dry) (void)hipStreamSynchronize(sx);
  };
  auto start = [&]() -> int {
    if (!ok) {
      lgp_set_error("hipEventCreate failed");
      return LGP_EHIP;
    }
    if (LGP_TRACED(ctx, LGP_TR_MEMSET, sc, 0)) LGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), sc));
    LGP_CHECK(lgp_tr_record(ctx, ev_start, sc));                               // (everything the caller enqueued on the compute stream)
    LGP_CHECK(lgp_tr_wait(ctx, s1, ev_start));
    LGP_CHECK(lgp_tr_wait(ctx, s3, ev_start));
    LGP_CHECK(lgp_tr_wait(ctx, sx, ev_start));
    return LGP_OK;
  };
  if (const int st0 = start(); st0 != LGP_OK) {
    bare_broadcasts(0);
    if (!dry) (void)hipStreamSynchronize(sc);
    return st0;
  }

  // a slot is written by the owner's pack (s1 / s3) or by the arriving broadcast (comm stream): both behind its last readers
  auto wait_slot_free = [&](size_t b, hipStream_t st) -> int {
    if (b < nslots) return LGP_OK;
    const size_t s = b % nslots;
    LGP_CHECK(lgp_tr_wait(ctx, st, ev_free_sc[s]));
    LGP_CHECK(lgp_tr_wait(ctx, st, ev_free_l1[s]));
    // a panel of this rank's own is read by its updates IN PLACE (below): what still reads its slot is the broadcast
    const size_t prev = b - nslots;
    if (owner(prev) == me && pieces[prev].size() > 1) LGP_CHECK(lgp_tr_wait(ctx, st, ev_recv[prev][pieces[prev].size() - 1]));
    return LGP_OK;
  };
  // The rows of a panel as an update reads them: a panel this rank factored itself is read where it lies, in the factor (its rows
  // are final once solved: `ev_solved`), so that neither the copy into the slot nor the broadcast stands between the solve and
  // this rank's own updates; a panel of another rank from its slot, once received
  auto prow = [&](size_t b, size_t r, size_t &ld) -> const double * {
    if (owner(b) == me) { ld = ldl; return block(b) + (r - b * nb) * ldl; }
    ld = ldp;
    return panel_rows(b, r);
  };
  auto avail = [&](size_t b, size_t i) { return owner(b) == me ? ev_solved[b][i] : ev_recv[b][i]; };

  // ---- owner(b): update its piece of block column b with panel b - 1, factor / solve it, pack it
  // U1: the square of block column b gets panel b - 1, on the stream that factors it (nothing else stands between the arrival of
  // those rows and the next square)
  auto u1 = [&](size_t b) -> int {
    if (b == 0) return LGP_OK;
    const size_t kb = width(b);
    if (tr != nullptr && tr->log) lgp_tr_log(ctx, LGP_TR_MARK, s1, {3, (long long)b, 0, (long long)(b * nb), (long long)(b * nb + kb)});
    LGP_CHECK(lgp_tr_wait(ctx, s1, avail(b - 1, cover(b - 1, b * nb))));
    if (b >= 2) LGP_CHECK(lgp_tr_wait(ctx, s1, ev_bulk[b - 2][cover(b - 2, b * nb)]));   // ... and has every earlier update
    size_t lda = 0;
    const double *A = prow(b - 1, b * nb, lda);
    return lgp_gemm_launch<double>(ctx, s1, false, true, kb, kb, width(b - 1), -1.0, A, lda, A, lda, 1.0, block(b), ldl, true);
  };
  auto lane = [&](const Piece &p) -> int {
    const size_t b = p.b, kb = width(b);
    hipStream_t st = p.i == 0 ? s1 : s3;
    double *C = block(b) + (p.r0 - b * nb) * ldl;
    if (p.i == 0) LGP_CHECK(u1(b));
    if (tr != nullptr && tr->log) lgp_tr_log(ctx, LGP_TR_MARK, st, {0, (long long)p.b, (long long)p.i, (long long)p.r0, (long long)p.r1});
    // (U2, the rows below the square, was applied on the compute stream in front of the bulk update with the same piece of panel
    // b - 1, see `bulk`: the two do not share the CUs, as in the single-GPU driver)
    if (b >= 1 && p.i >= 1) LGP_CHECK(lgp_tr_wait(ctx, st, ev_u2[b - 1][cover(b - 1, p.r0)]));
    if (p.i == 0) {
      // the resident workers of the chain kernel take CUs from the trailing update: worth it once this rank's share of what is
      // left of the matrix is small enough for the squares to bound the factorisation (the single-GPU rule: N <= 16384)
      const bool use_chain = chain_ok && (n - b * nb) / G <= ctx->chain_max_n;
      LGP_CHECK(lgp_square_factor_f64(ctx, st, kb, block(b), ldl, dblk(b), b * nb, use_chain, ev_blk[b].data()));
    } else {
      LGP_CHECK(lgp_far_rows_impl<double>(ctx, st, p.r1 - p.r0, kb, C, ldl, block(b), dblk(b), ev_blk[b].data()));
    }
    // the look-ahead stream of the squares is through with panel b - 1
    if (b >= 1 && p.i == 0) LGP_CHECK(lgp_tr_record(ctx, ev_free_l1[(b - 1) % nslots], st));
    // the square itself stays at home: the trailing updates only read rows BELOW a panel's diagonal block
    if (p.i == 0) return LGP_OK;
    LGP_CHECK(lgp_tr_record(ctx, ev_solved[b][p.i], st));
    if (G > 1) {                                          // (without peers nobody reads the slot)
      LGP_CHECK(wait_slot_free(b, st));
      bool enqueue = true;
      if (tr != nullptr) {
        long long sd, od, ss, os;
        lgp_tr_ptr(ctx, panel_rows(b, p.r0), sd, od); lgp_tr_ptr(ctx, C, ss, os);
        enqueue = LGP_TRACED(ctx, LGP_TR_COPY2D, st, sd, od, (long long)ldp, ss, os, (long long)ldl, (long long)(p.r1 - p.r0), (long long)kb);
      }
      if (stats) stamp(st, pack_pairs);
      if (enqueue)
        LGP_HIP(hipMemcpy2DAsync(panel_rows(b, p.r0), ldp * sizeof(double), C, ldl * sizeof(double), kb * sizeof(double), p.r1 - p.r0,
                                 hipMemcpyDeviceToDevice, st));
      if (stats) stamp(st, pack_pairs);
    }
    if (!owner_sends_in_order) LGP_CHECK(lgp_tr_record(ctx, ev_ready[b][p.i], st));
    return LGP_OK;
  };
  // ---- every rank: the piece travels
  auto comm = [&](const Piece &p) -> int {
    // A piece of this rank's own is sent from the stream that solved and packed it (s3), in stream order.  Handing it to the
    // communication stream through an event (rounds 2-3) cost ~170 us per panel, 5.5 ms at C3, measured with one rank
    // (LGP_DIST_FORCE_COMM; LGP_DIST_OWNER_COMM=stream brings it back): the one wait that sits on the otherwise idle stream until
    // the rows are solved delays the dispatches of the compute stream behind ITS event waits (U2 by ~80 us, the large update by
    // ~50 us in the kernel trace), whether the wait is an event or a lane spinning on a flag word, with 4 or 8 hardware queues.
    // What the communication stream keeps are the pieces of the OTHER ranks' panels: their slot has long been free when they are
    // enqueued, and the receiving kernel itself waits for the data.  (One communicator on two streams: RCCL orders the kernels of
    // a communicator in issue order through its own stream, whatever streams they were issued on.)
    hipStream_t sq = (owner(p.b) == me && owner_sends_in_order) ? s3 : sx;
    if (tr != nullptr && tr->log) lgp_tr_log(ctx, LGP_TR_MARK, sq, {1, (long long)p.b, (long long)p.i, (long long)p.r0, (long long)p.r1});
    if (owner(p.b) != me) LGP_CHECK(wait_slot_free(p.b, sx));
    else if (sq == sx) LGP_CHECK(lgp_tr_wait(ctx, sx, ev_ready[p.b][p.i]));
    {
      bool enqueue = true;
      if (tr != nullptr) {
        long long sp, op;
        lgp_tr_ptr(ctx, panel_rows(p.b, p.r0), sp, op);
        enqueue = LGP_TRACED(ctx, LGP_TR_BCAST, sq, sp, op, (long long)((p.r1 - p.r0) * ldp), (long long)owner(p.b));
      }
      OnStream on(ctx, sq);
      if (enqueue) LGP_CHECK(lgp_bcast(ctx, panel_rows(p.b, p.r0), (p.r1 - p.r0) * ldp * sizeof(double), (int)owner(p.b)));
    }
    LGP_CHECK(lgp_tr_record(ctx, ev_recv[p.b][p.i], sq));
    return LGP_OK;
  };
  // ---- every rank: its block columns c >= b + 2, rows of the piece, in one staircase launch
  auto bulk = [&](const Piece &p) -> int {
    const size_t b = p.b;
    if (tr != nullptr && tr->log) lgp_tr_log(ctx, LGP_TR_MARK, sc, {2, (long long)p.b, (long long)p.i, (long long)p.r0, (long long)p.r1});
    if (stats) stamp(sc, wait_pairs);
    LGP_CHECK(lgp_tr_wait(ctx, sc, avail(b, p.i)));
    if (stats) stamp(sc, wait_pairs);
    (owner(b) == me ? bytes_own : bytes_recv) += (double)((p.r1 - p.r0) * width(b) * sizeof(double));
    size_t ldq = 0;
    (void)prow(b, p.r0, ldq);                              // the leading dimension of this panel's rows for this rank
    if (b + 1 < nblocks && owner(b + 1) == me) {
      // U2: the look-ahead column b + 1 below its square, rows of this piece, FIRST: its owner solves those rows next (s3)
      const size_t r2 = std::max(p.r0, std::min((b + 2) * nb, n));
      if (r2 < p.r1) {
        LGP_CHECK(lgp_tr_wait(ctx, sc, avail(b, cover(b, (b + 1) * nb))));
        LGP_CHECK(lgp_gemm_launch<double>(ctx, sc, false, true, p.r1 - r2, width(b + 1), width(b), -1.0, prow(b, r2, ldq), ldq, prow(b, (b + 1) * nb, ldq), ldq,
                                          1.0, block(b + 1) + (r2 - (b + 1) * nb) * ldl, ldl, false, 2));
      }
      LGP_CHECK(lgp_tr_record(ctx, ev_u2[b][p.i], sc));
    }
    size_t c0 = b + 2;
    while (c0 < nblocks && owner(c0) != me) ++c0;
    if (c0 < nblocks) {
      const size_t rs = std::max(p.r0, c0 * nb);
      if (rs < p.r1) {
        LGP_CHECK(lgp_tr_wait(ctx, sc, avail(b, cover(b, c0 * nb))));                    // (arrived earlier: same stream order; explicit)
        size_t ncols = 0;
        for (size_t c = c0; c < nblocks; c += G) ncols += width(c);
        const double *A = prow(b, rs, ldq), *B = prow(b, c0 * nb, ldq);
        double *C = block(c0) + (rs - c0 * nb) * ldl;
        // (late panels: the updates leave room on every CU for the latency-bound kernels of the look-ahead streams, as in potrf_split)
        const size_t left = n - std::min(n, (b + 1) * nb);
        ctx->gemm_thin_lds = left <= 6144 ? (size_t)70 * 1024 : (left <= 12288 ? (size_t)20 * 1024 : 0);
        if (nb % 128 == 0) {
          const int st = lgp_gemm_launch<double>(ctx, sc, false, true, p.r1 - rs, ncols, width(b), -1.0, A, ldq, B, ldq, 1.0, C, ldl, true, 3, (int)nb,
                                                 (int)(G - 1), (int)(rs - c0 * nb));
          ctx->gemm_thin_lds = 0;
          LGP_CHECK(st);
        } else {
          ctx->gemm_thin_lds = 0;
          // block widths that are not whole tiles: one launch per block column
          for (size_t c = c0; c < nblocks; c += G) {
            const size_t rc = std::max(p.r0, c * nb);
            if (rc >= p.r1) continue;
            LGP_CHECK(lgp_gemm_launch<double>(ctx, sc, false, true, p.r1 - rc, width(c), width(b), -1.0, prow(b, rc, ldq), ldq, prow(b, c * nb, ldq), ldq,
                                              1.0, block(c) + (rc - c * nb) * ldl, ldl, true, 2, 0, 0, (int)(rc - c * nb)));
          }
        }
      }
    }
    LGP_CHECK(lgp_tr_record(ctx, ev_bulk[b][p.i], sc));
    if (p.i + 1 == pieces[b].size()) LGP_CHECK(lgp_tr_record(ctx, ev_free_sc[b % nslots], sc));
    return LGP_OK;
  };

  // A failure in the middle must not leave the peers inside a collective: the broadcasts of the rest of the plan are still
  // issued (their contents no longer matter), every stream is drained, then the error is reported.
  int failed = LGP_OK;
  for (size_t t = 0; t < plan.size(); ++t) {
    const Piece &p = plan[t];
    if (chain_tail != 0 && p.i == 0 && p.b >= 1 && n - p.b * nb <= chain_tail) {
      // (everything of the earlier panels is enqueued: piece 1 of panel b - 1 sorts in front of the square of panel b)
      failed = u1(p.b);
      if (failed == LGP_OK) {
        bool enqueue = true;
        if (tr != nullptr) {
          long long sl, ol, sw, ow;
          lgp_tr_ptr(ctx, block(p.b), sl, ol); lgp_tr_ptr(ctx, dblk(p.b), sw, ow);
          enqueue = LGP_TRACED(ctx, LGP_TR_TAIL, sc, (long long)(n - p.b * nb), sl, ol, (long long)ldl, ow, (long long)nx, (long long)(p.b * nb));
        }
        if (enqueue) {
          lgp_tracer *const keep = ctx->tracer;        // (the chain driver's own launches are not part of this schedule)
          ctx->tracer = nullptr;
          failed = lgp_potrf_chain_tail_f64(ctx, n - p.b * nb, block(p.b), ldl, dblk(p.b), nx, p.b * nb);
          ctx->tracer = keep;
        }
      }
      break;
    }
    if (failed == LGP_OK && owner(p.b) == me) failed = lane(p);
    if (p.i == 0) continue;                                // (squares are not sent, see `lane`)
    // (without peers nothing travels; LGP_DIST_FORCE_COMM issues the -- empty -- broadcasts all the same, see `comm`)
    if (failed != LGP_OK) {                                // (the events of a broken schedule are not to be trusted: no more waits)
      bare_broadcasts(t);
      break;
    }
    failed = (G > 1 || force_comm) ? comm(p) : LGP_OK;
    if (failed != LGP_OK) {                                // (this piece's broadcast may or may not have been issued: nothing sane is left
      bare_broadcasts(t + 1);                              //  to do about that one; the rest of the plan is)
      break;
    }
    failed = bulk(p);
  }
  if (failed == LGP_OK) {
    int e = lgp_tr_record(ctx, ev_end1, s1);
    if (e == LGP_OK) e = lgp_tr_record(ctx, ev_end3, s3);
    if (e == LGP_OK) e = lgp_tr_record(ctx, ev_endx, sx);
    if (e == LGP_OK) e = lgp_tr_wait(ctx, sc, ev_end1);
    if (e == LGP_OK) e = lgp_tr_wait(ctx, sc, ev_end3);
    if (e == LGP_OK) e = lgp_tr_wait(ctx, sc, ev_endx);
    failed = e;
  }
  // L = L~ * s[:, None] (_decomp.py:392) on the matrix rows of every owned block column; then the packed images of the diagonal
  // blocks that the solves read (D, D', 1 / diag), from the finished factor
  if (failed == LGP_OK && dsrow != nullptr && nlocal > 0 && LGP_TRACED(ctx, LGP_TR_SCALE, sc, (long long)n, (long long)(nlocal * nb), 0, 0, (long long)ldl))
    failed = lgp_scale_rows_cols_f64(ctx, n, nlocal * nb, dLloc, ldl, dsrow, nullptr, 0);
  for (size_t b = me; b < nblocks && failed == LGP_OK; b += G) {
    bool enqueue = true;
    if (tr != nullptr) {
      long long sl, ol, sw, ow;
      lgp_tr_ptr(ctx, block(b), sl, ol); lgp_tr_ptr(ctx, dblk(b), sw, ow);
      enqueue = LGP_TRACED(ctx, LGP_TR_EXTRACT, sc, sl, ol, (long long)ldl, (long long)width(b), ow);
    }
    if (enqueue) failed = lgp_extract_dblk_public(ctx, sc, block(b), ldl, width(b), dblk(b));
  }
  if (dry) return failed;
  int info = 0;
  if (failed == LGP_OK && hipMemcpyAsync(&info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, sc) != hipSuccess) failed = LGP_EHIP;
  // drain everything before the events go away, on every path
  (void)hipStreamSynchronize(s1);
  (void)hipStreamSynchronize(s3);
  (void)hipStreamSynchronize(sx);
  (void)hipStreamSynchronize(sc);
  tracer_scope.dump(failed);
  if (stats) {
    auto total = [](const std::vector<hipEvent_t> &v) {
      double ms = 0;
      for (size_t k = 0; k + 1 < v.size(); k += 2) {
        float t = 0;
        if (hipEventElapsedTime(&t, v[k], v[k + 1]) == hipSuccess) ms += t;
      }
      return ms;
    };
    double *o = ctx->dist_stats;
    o[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();      // the call, wall clock
    o[1] = total(wait_pairs);           // the compute stream idle in front of a piece that had not arrived / was not solved yet
    o[2] = total(pack_pairs);
    o[3] = bytes_recv;                  // panel bytes this rank's updates read from other ranks
    o[4] = bytes_own;
    o[5] = (double)(wait_pairs.size() / 2);
    o[6] = (G >= 3 && ctx->bcast_algo != 1 && (ctx->bcast_algo == 2 || ctx->comm != nullptr)) ? 2 : (G >= 2 ? 1 : 0);   // 2: scatter + all-gather (pieces >= bcast_min), 1: the transport's broadcast
    o[7] = owner_sends_in_order ? 0 : 1;
  }
  if (failed != LGP_OK) return failed;
  if (hinfo) *hinfo = info;
  return lgp_info_result(ctx, info) < 0 ? LGP_EHIP : LGP_OK;
}

// dist.hip -- the 1-D block-column-cyclic Cholesky across the GPUs of one node, driven from the library: no host language
// between the panels (the reference has no distributed code).  One process per GPU; block column b
// (columns [b nb, (b+1) nb), all rows) lives on rank b % G.  Right-looking with one-panel look-ahead:
//
//     step b:   every rank has panel b (rows b nb .., nb columns) in panel slot b % 2
//               owner(b+1):  update block column b+1 with panel b, factor it, pack it           [compute stream]
//               all ranks:   broadcast panel b+1 from owner(b+1) into slot (b+1) % 2           [comm stream: RCCL / host transport]
//               all ranks:   update ALL their other owned block columns c > b with panel b     [compute stream, one staircase launch]
//
// so the (N - k) x nb panel message travels while the trailing updates (MFMA) run.  Panels are packed with a PADDED leading
// dimension (ldp = nb + 16 doubles by default): with ld = nb = 1024 every row of the A operand of the trailing update
// starts on the same HBM channel (8 KiB stride), which was measured at 62 instead of 68.7 TFLOP/s.
#include "common.h"

namespace {

struct Events {
  hipEvent_t ready[2] = {nullptr, nullptr}, recv[2] = {nullptr, nullptr}, freed[2] = {nullptr, nullptr};
  int create() {
    for (int s = 0; s < 2; ++s) {
      LGP_HIP(hipEventCreateWithFlags(&ready[s], hipEventDisableTiming));
      LGP_HIP(hipEventCreateWithFlags(&recv[s], hipEventDisableTiming));
      LGP_HIP(hipEventCreateWithFlags(&freed[s], hipEventDisableTiming));
    }
    return LGP_OK;
  }
  ~Events() {
    for (int s = 0; s < 2; ++s) {
      if (ready[s]) (void)hipEventDestroy(ready[s]);
      if (recv[s]) (void)hipEventDestroy(recv[s]);
      if (freed[s]) (void)hipEventDestroy(freed[s]);
    }
  }
};

// the context's entry points enqueue on ctx->stream; the broadcasts go on the auxiliary stream by swapping the two for the
// duration of the call (RAII: always swapped back)
struct OnCommStream {
  lgp_ctx *c;
  explicit OnCommStream(lgp_ctx *ctx) : c(ctx) { std::swap(c->stream, c->stream2); }
  ~OnCommStream() { std::swap(c->stream, c->stream2); }
};

}  // namespace

extern "C" int lgp_chol_factor_dist_f64(lgp_ctx *ctx, size_t n, size_t nb, size_t nx, double *dLloc, size_t ldl, double *ddblk,
                                        double *dpanels, size_t ldp, const double *dsrow, int *hinfo) {
  LGP_REQUIRE(ctx != nullptr && dLloc != nullptr && ddblk != nullptr && dpanels != nullptr, "null pointer");
  LGP_REQUIRE(n >= 1 && nb >= 1 && nb <= n + nb, "bad sizes");
  LGP_REQUIRE(ldp >= nb, "ldp < nb");
  const size_t G = (size_t)ctx->nranks, me = (size_t)ctx->rank;
  const size_t nblocks = (n + nb - 1) / nb;
  const size_t nlocal = (nblocks + G - 1 - me) / G;                   // blocks b with b % G == me
  LGP_REQUIRE(nlocal == 0 || ldl >= nlocal * nb, "ldl smaller than the local block columns");
  DeviceGuard dg(ctx->device);
  hipStream_t sc = ctx->stream, sm = ctx->stream2;                    // compute, communication
  Events ev;
  LGP_CHECK(ev.create());
  const size_t rows_all = n + nx;
  double *P[2] = {dpanels, dpanels + rows_all * ldp};
  const size_t dblk_per_block = lgp_chol_dblk_size(nb);
  LGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), sc));
  ctx->potf2_small = n <= 8192;

  auto owner = [&](size_t b) { return b % G; };
  auto width = [&](size_t b) { return (n - b * nb) < nb ? (n - b * nb) : nb; };
  auto rowsof = [&](size_t b) { return n - b * nb + nx; };            // panel rows: from the diagonal of block b down, plus carried rows
  auto block = [&](size_t b) { return dLloc + (b * nb) * ldl + (b / G) * nb; };   // local block column b from its diagonal down

  // the latency-bound panel factorisation and the packing run on the auxiliary (high priority) stream, in front of the
  // broadcast of the same panel: the owner's trailing update of the other block columns proceeds beside them on the
  // compute stream, as the look-ahead of the single-GPU driver does
  auto panel_factor = [&](size_t b, hipStream_t st) -> int {
    LGP_CHECK(lgp_panel_factor_impl<double>(ctx, st, rowsof(b), width(b), block(b), ldl, ddblk + (b / G) * dblk_per_block, b * nb));
    return lgp_extract_dblk_public(ctx, st, block(b), ldl, width(b), ddblk + (b / G) * dblk_per_block);
  };
  auto pack = [&](size_t b, int slot, hipStream_t st) -> int {
    LGP_HIP(hipMemcpy2DAsync(P[slot], ldp * sizeof(double), block(b), ldl * sizeof(double), width(b) * sizeof(double), rowsof(b),
                             hipMemcpyDeviceToDevice, st));
    return LGP_OK;
  };
  auto bcast = [&](size_t b, int slot) -> int {
    OnCommStream on(ctx);
    return lgp_bcast(ctx, P[slot], rowsof(b) * ldp * sizeof(double), (int)owner(b));
  };
  // block column c -= P_b[rows of c] P_b[block rows of c]'
  auto update_one = [&](size_t c, size_t b, int slot) -> int {
    const double *A = P[slot] + (c - b) * nb * ldp;
    return lgp_gemm_launch<double>(ctx, sc, false, true, rowsof(c), width(c), width(b), -1.0, A, ldp, A, ldp, 1.0, block(c), ldl, true, 2);
  };
  // every remaining owned block column (consecutive local blocks from c0 on) in one staircase launch
  auto update_rest = [&](size_t c0, size_t b, int slot) -> int {
    size_t ncols = 0;
    for (size_t c = c0; c < nblocks; c += G) ncols += width(c);
    if (nb % 128 != 0) {
      for (size_t c = c0; c < nblocks; c += G) LGP_CHECK(update_one(c, b, slot));
      return LGP_OK;
    }
    const double *A = P[slot] + (c0 - b) * nb * ldp;
    return lgp_gemm_launch<double>(ctx, sc, false, true, rowsof(c0), ncols, width(b), -1.0, A, ldp, A, ldp, 1.0, block(c0), ldl, true, 3, (int)nb,
                                   (int)(G - 1));
  };

  LGP_HIP(hipEventRecord(ev.ready[0], sc));                              // (everything the caller enqueued on the compute stream)
  LGP_HIP(hipStreamWaitEvent(sm, ev.ready[0], 0));
  if (owner(0) == me) {
    LGP_CHECK(panel_factor(0, sm));
    LGP_CHECK(pack(0, 0, sm));
  }
  LGP_CHECK(bcast(0, 0));
  LGP_HIP(hipEventRecord(ev.recv[0], sm));
  for (size_t b = 0; b < nblocks; ++b) {
    const int slot = (int)(b % 2), nslot = (int)((b + 1) % 2);
    LGP_HIP(hipStreamWaitEvent(sc, ev.recv[slot], 0));
    const size_t nxt = b + 1;
    const bool mine = nxt < nblocks && owner(nxt) == me;
    // compute stream: the look-ahead block column first, then every other owned block column in one staircase launch
    if (mine) {
      LGP_CHECK(update_one(nxt, b, slot));
      LGP_HIP(hipEventRecord(ev.ready[nslot], sc));
    }
    size_t c0 = b + 1;
    while (c0 < nblocks && (owner(c0) != me || c0 == nxt)) ++c0;
    if (c0 < nblocks) LGP_CHECK(update_rest(c0, b, slot));
    LGP_HIP(hipEventRecord(ev.freed[slot], sc));
    // auxiliary stream: factor + pack (owner) and broadcast panel b + 1.  (Enqueued after the trailing update so that a
    // transport that blocks the host -- the host-staged one of the tests -- does not hold the update back.)
    if (nxt < nblocks) {
      if (b >= 1) LGP_HIP(hipStreamWaitEvent(sm, ev.freed[nslot], 0));      // the slot's previous panel is no longer read
      if (mine) {
        LGP_HIP(hipStreamWaitEvent(sm, ev.ready[nslot], 0));
        LGP_CHECK(panel_factor(nxt, sm));
        LGP_CHECK(pack(nxt, nslot, sm));
      }
      LGP_CHECK(bcast(nxt, nslot));
      LGP_HIP(hipEventRecord(ev.recv[nslot], sm));
    }
  }
  if (dsrow != nullptr && nlocal > 0) {
    // L = L~ * s[:, None] (_decomp.py:392) on the matrix rows of every owned block column, then the packed diagonal blocks
    // again from the scaled factor
    LGP_CHECK(lgp_scale_rows_cols_f64(ctx, n, nlocal * nb, dLloc, ldl, dsrow, nullptr, 0));
    for (size_t b = me; b < nblocks; b += G)
      LGP_CHECK(lgp_extract_dblk_public(ctx, sc, block(b), ldl, width(b), ddblk + (b / G) * dblk_per_block));
  }
  int info = 0;
  LGP_HIP(hipMemcpyAsync(&info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, sc));
  LGP_HIP(hipStreamSynchronize(sm));
  LGP_HIP(hipStreamSynchronize(sc));
  if (hinfo) *hinfo = info;
  return lgp_info_result(ctx, info) < 0 ? LGP_EHIP : LGP_OK;
}

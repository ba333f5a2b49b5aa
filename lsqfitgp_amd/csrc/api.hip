// api.hip -- context, memory, timing, the fused posterior / log-density conveniences and the RCCL
// plumbing of liblsqfitgp_hip.so.  See include/lsqfitgp_hip.h for the contract of every entry point.
#include "common.h"
#include <cstdlib>
#include <vector>
#include <functional>
#include <algorithm>
#include <dlfcn.h>
#include <cmath>
#include <rccl/rccl.h>

// ---- error string ----------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void lgp_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

template <typename T>
int lgp_dot_strided_device(lgp_ctx *ctx, hipStream_t stream, size_t n, const T *x, size_t incx, const T *y, size_t incy, T *dout);
template <typename T>
int lgp_logdet_device(lgp_ctx *ctx, hipStream_t stream, size_t n, const T *L, size_t ldl, T *dout);
template <typename T>
int lgp_frob_device(lgp_ctx *ctx, hipStream_t stream, size_t rows, size_t cols, const T *A, size_t lda, const T *B, size_t ldb, T *dpart, T *dout);
template <typename T>
int lgp_set_diag_device(lgp_ctx *ctx, hipStream_t stream, size_t n, T *A, size_t lda, T value);

// ---- workspaces -----------------------------------------------------------------------------------------------------
static void ws_trim(lgp_ctx *c, size_t keep) {
  // idle blocks, largest first, until the idle total fits (hipFree waits for the device: nothing can still be using them)
  for (;;) {
    size_t idle = 0, big = (size_t)-1;
    for (size_t i = 0; i < c->ws_blocks.size(); ++i)
      if (!c->ws_blocks[i].busy) {
        idle += c->ws_blocks[i].bytes;
        if (big == (size_t)-1 || c->ws_blocks[i].bytes > c->ws_blocks[big].bytes) big = i;
      }
    if (idle <= keep || big == (size_t)-1) return;
    (void)hipFree(c->ws_blocks[big].ptr);
    if (c->ws_blocks[big].ev) (void)hipEventDestroy(c->ws_blocks[big].ev);
    c->ws_blocks.erase(c->ws_blocks.begin() + (long)big);
  }
}

// LGP_WS_ASYNC=1 (diagnosis only, tools/malloc_async_probe.hip and DESIGN.md): the stream-ordered allocator of the runtime instead
static bool ws_async() {
  static const bool on = getenv("LGP_WS_ASYNC") != nullptr && atoi(getenv("LGP_WS_ASYNC")) != 0;
  return on;
}

int lgp_ws_alloc(lgp_ctx *c, hipStream_t stream, size_t bytes, void **out) {
  *out = nullptr;
  if (bytes == 0) bytes = 256;
  bytes = (bytes + 255) / 256 * 256;
  if (ws_async()) {
    LGP_HIP(hipMallocAsync(out, bytes, stream));
    return LGP_OK;
  }
  size_t best = (size_t)-1;
  for (size_t i = 0; i < c->ws_blocks.size(); ++i) {
    const auto &b = c->ws_blocks[i];
    if (!b.busy && b.bytes >= bytes && b.bytes <= 2 * bytes + 4096 && (best == (size_t)-1 || b.bytes < c->ws_blocks[best].bytes)) best = i;
  }
  if (best == (size_t)-1) {
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {          // give back what is idle and try once more
      (void)hipGetLastError();
      ws_trim(c, 0);
      e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      lgp_set_error("hipMalloc of a %zu-byte workspace failed: %s", bytes, hipGetErrorString(e));
      return LGP_ENOMEM;
    }
    hipEvent_t ev = nullptr;
    LGP_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    c->ws_blocks.push_back({p, bytes, nullptr, ev, false});
    best = c->ws_blocks.size() - 1;
  }
  auto &b = c->ws_blocks[best];
  if (b.last != nullptr && b.last != stream) LGP_HIP(hipStreamWaitEvent(stream, b.ev, 0));
  b.busy = true;
  *out = b.ptr;
  return LGP_OK;
}

int lgp_ws_free(lgp_ctx *c, hipStream_t stream, void *ptr) {
  if (ptr == nullptr) return LGP_OK;
  if (ws_async()) {
    LGP_HIP(hipFreeAsync(ptr, stream));
    return LGP_OK;
  }
  for (auto &b : c->ws_blocks)
    if (b.ptr == ptr) {
      b.busy = false;
      b.last = stream;
      hipError_t e = hipEventRecord(b.ev, stream);
      ws_trim(c, c->ws_keep_bytes);
      if (e != hipSuccess) {
        lgp_set_error("hipEventRecord failed: %s", hipGetErrorString(e));
        return LGP_EHIP;
      }
      return LGP_OK;
    }
  lgp_set_error("lgp_ws_free: not a workspace block");
  return LGP_EINVAL;
}

extern "C" {

int lgp_version(void) { return 1; }
const char *lgp_last_error(void) { return g_err; }

int lgp_device_count(int *count) {
  LGP_REQUIRE(count != nullptr, "null pointer");
  *count = 0;
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    lgp_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return LGP_EHIP;
  }
  return LGP_OK;
}

int lgp_ctx_create(int device, lgp_ctx **out) {
  LGP_REQUIRE(out != nullptr, "null pointer");
  *out = nullptr;
  int count = 0;
  LGP_HIP(hipGetDeviceCount(&count));
  LGP_REQUIRE(device >= 0 && device < count, "no such HIP device (is a GPU visible?)");
  LGP_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  LGP_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    lgp_set_error("device %d is %s: this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    return LGP_EINVAL;
  }
  lgp_ctx *c = new lgp_ctx();
  // any failure below releases what was created so far (lgp_ctx_destroy accepts a partially built context)
  struct Guard {
    lgp_ctx *c;
    ~Guard() { if (c) lgp_ctx_destroy(c); }
  } guard{c};
  c->device = device;
  c->cus = prop.multiProcessorCount;
  if (const char *m = getenv("LGP_POTF2")) c->potf2_mode = strcmp(m, "valu") == 0 ? 1 : (strcmp(m, "mfma") == 0 ? 2 : (strcmp(m, "mfma1") == 0 ? 3 : 0));
  if (const char *m = getenv("LGP_CHAIN")) c->chain_mode = strcmp(m, "0") == 0 ? 1 : 0;
  if (const char *m = getenv("LGP_CHAIN_MAX_N")) c->chain_max_n = (size_t)strtoull(m, nullptr, 10);
  if (const char *m = getenv("LGP_CHAIN_HOG_MAX_N")) c->chain_hog_max_n = (size_t)strtoull(m, nullptr, 10);
  if (const char *m = getenv("LGP_CHAIN_WORKERS")) {
    const int wk = atoi(m);
    if (wk >= 1 && wk <= 256) c->chain_workers = wk;
  }
  if (const char *m = getenv("LGP_CHAIN_GROUPS")) {
    const int g = atoi(m);
    if (g >= 1 && g <= 4) c->chain_groups = g;
  }
  if (const char *m = getenv("LGP_CHAIN_TIMEOUT_MS")) {
    const long ms = atol(m);
    if (ms >= 1 && ms <= 600000) c->chain_timeout = (unsigned long long)ms * 100000ull;
  }
  if (const char *m = getenv("LGP_CHAIN_TEST_STALL")) c->chain_test_stall = atoi(m) != 0;
  if (const char *m = getenv("LGP_CHAIN_LDS_KB")) {
    const int kb = atoi(m);
    if (kb >= 0 && kb <= 140) c->chain_lds_kb = kb;
  }
  // stream2 carries the latency-bound panel factorisation of the look-ahead: highest priority, so that its
  // few workgroups are dispatched ahead of the queued trailing-update workgroups of the main stream
  int prio_lo = 0, prio_hi = 0;
  LGP_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
  c->prio_hi = prio_hi;
  LGP_HIP(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_lo));
  LGP_HIP(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_hi));
  LGP_HIP(hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, prio_hi));
  if (const char *m = getenv("LGP_RESERVE")) {
    int rc = 0, rf = 0;
    if (sscanf(m, "%d,%d", &rc, &rf) == 2 && rc >= 0 && rf >= 0 && rc + rf < c->cus) {
      c->reserve_chain = rc;
      c->reserve_far = rf;
    }
  }
  if (const char *m = getenv("LGP_RESERVE_MAX_N")) c->reserve_max_n = (size_t)strtoull(m, nullptr, 10);
  if (const char *m = getenv("LGP_BCAST_ALGO")) c->bcast_algo = strcmp(m, "plain") == 0 ? 1 : (strcmp(m, "scatter") == 0 ? 2 : 0);
  if (const char *m = getenv("LGP_BCAST_MIN_BYTES")) c->bcast_min = (size_t)strtoull(m, nullptr, 10);
  if (c->reserve_max_n > 0 && c->reserve_chain + c->reserve_far > 0 && c->cus > 64) {
    const int words = (c->cus + 31) / 32;
    std::vector<uint32_t> mg((size_t)words, 0u), mf((size_t)words, 0u);
    for (int i = 0; i < c->cus; ++i) {
      if (i >= c->reserve_chain + c->reserve_far) mg[(size_t)(i / 32)] |= 1u << (i % 32);
      if (i >= c->reserve_chain) mf[(size_t)(i / 32)] |= 1u << (i % 32);
    }
    LGP_HIP(hipExtStreamCreateWithCUMask(&c->stream_gm, (uint32_t)words, mg.data()));
    LGP_HIP(hipExtStreamCreateWithCUMask(&c->stream_fm, (uint32_t)words, mf.data()));
  }
  LGP_HIP(hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming));
  LGP_HIP(hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming));
  LGP_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_info), sizeof(int)));
  LGP_HIP(hipMemset(c->d_info, 0, sizeof(int)));
  c->scratch_bytes = 1 << 20;
  LGP_HIP(hipMalloc(&c->d_scratch, c->scratch_bytes));
  LGP_HIP(hipHostMalloc(&c->h_pinned, 4096, hipHostMallocDefault));
  if (const char *m = getenv("LGP_WS_KEEP_MB")) c->ws_keep_bytes = (size_t)strtoull(m, nullptr, 10) << 20;
  guard.c = nullptr;
  *out = c;
  return LGP_OK;
}

int lgp_ctx_destroy(lgp_ctx *c) {
  if (!c) return LGP_OK;
  DeviceGuard dg(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  for (auto &b : c->ws_blocks) b.busy = false;
  ws_trim(c, 0);
  lgp_comm_destroy(c);
  if (c->d_info) (void)hipFree(c->d_info);
  if (c->chain_trace) {
    if (const char *path = getenv("LGP_CHAIN_TRACE")) {
      std::vector<unsigned long long> tr(c->chain_trace_tasks * 4);
      std::vector<unsigned int> sc(c->chain_trace_tasks * 2);
      if (hipMemcpy(tr.data(), c->chain_trace, tr.size() * 8, hipMemcpyDeviceToHost) == hipSuccess &&
          hipMemcpy(sc.data(), c->chain_trace_sched, sc.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
        if (FILE *f = fopen(path, "w")) {
          fprintf(f, "# task type k j r c worker taken ready done (10 ns ticks)\n");
          for (size_t t = 0; t < c->chain_trace_tasks; ++t)
            fprintf(f, "%zu %u %u %u %u %u %llu %llu %llu %llu\n", t, sc[2 * t] & 255u, sc[2 * t] >> 24, (sc[2 * t] >> 8) & 255u, sc[2 * t + 1],
                    (sc[2 * t] >> 16) & 255u, tr[4 * t], tr[4 * t + 1], tr[4 * t + 2], tr[4 * t + 3]);
          fclose(f);
        }
      }
    }
    (void)hipFree(c->chain_trace);
  }
  if (c->chain_state) (void)hipFree(c->chain_state);
  for (const auto &e : c->chain_sched) (void)hipFree(e.tasks);
  if (c->d_scratch) (void)hipFree(c->d_scratch);
  if (c->h_pinned) (void)hipHostFree(c->h_pinned);
  if (c->h_small) (void)hipHostFree(c->h_small);
  if (c->ev_a) (void)hipEventDestroy(c->ev_a);
  if (c->ev_b) (void)hipEventDestroy(c->ev_b);
  for (hipEvent_t e : c->prof_events) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->panel_events) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->dist_events) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->dist_timing_events) (void)hipEventDestroy(e);
  if (c->stream3) { (void)hipStreamSynchronize(c->stream3); (void)hipStreamDestroy(c->stream3); }
  for (hipStream_t s : c->far_streams)
    if (s != c->stream3) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
  if (c->stream_comm) { (void)hipStreamSynchronize(c->stream_comm); (void)hipStreamDestroy(c->stream_comm); }
  if (c->stream_gm) { (void)hipStreamSynchronize(c->stream_gm); (void)hipStreamDestroy(c->stream_gm); }
  if (c->stream_fm) { (void)hipStreamSynchronize(c->stream_fm); (void)hipStreamDestroy(c->stream_fm); }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  delete c;
  return LGP_OK;
}

int lgp_sync(lgp_ctx *c) {
  LGP_REQUIRE(c != nullptr, "null context");
  DeviceGuard dg(c->device);
  LGP_HIP(hipStreamSynchronize(c->stream));
  LGP_HIP(hipStreamSynchronize(c->stream2));
  LGP_HIP(hipStreamSynchronize(c->stream3));
  for (hipStream_t s : c->far_streams) LGP_HIP(hipStreamSynchronize(s));
  if (c->stream_comm) LGP_HIP(hipStreamSynchronize(c->stream_comm));
  if (c->stream_gm) LGP_HIP(hipStreamSynchronize(c->stream_gm));
  if (c->stream_fm) LGP_HIP(hipStreamSynchronize(c->stream_fm));
  return LGP_OK;
}

int lgp_device_info(lgp_ctx *c, char *name, int namelen, int *cus, int *clock_khz, size_t *mem_bytes) {
  LGP_REQUIRE(c != nullptr, "null context");
  hipDeviceProp_t prop;
  LGP_HIP(hipGetDeviceProperties(&prop, c->device));
  if (name && namelen > 0) snprintf(name, (size_t)namelen, "%s (%s)", prop.name, prop.gcnArchName);
  if (cus) *cus = prop.multiProcessorCount;
  if (clock_khz) *clock_khz = prop.clockRate;
  if (mem_bytes) *mem_bytes = prop.totalGlobalMem;
  return LGP_OK;
}

int lgp_malloc(lgp_ctx *c, size_t bytes, void **dptr) {
  LGP_REQUIRE(c != nullptr && dptr != nullptr, "null pointer");
  *dptr = nullptr;
  if (bytes == 0) bytes = 16;
  DeviceGuard dg(c->device);
  hipError_t e = hipMalloc(dptr, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    lgp_set_error("hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    return LGP_ENOMEM;
  }
  return LGP_OK;
}

int lgp_free(lgp_ctx *c, void *dptr) {
  LGP_REQUIRE(c != nullptr, "null context");
  if (!dptr) return LGP_OK;
  DeviceGuard dg(c->device);
  LGP_HIP(hipStreamSynchronize(c->stream));
  LGP_HIP(hipStreamSynchronize(c->stream2));
  LGP_HIP(hipFree(dptr));
  return LGP_OK;
}

int lgp_memset(lgp_ctx *c, void *dptr, int byte, size_t bytes) {
  LGP_REQUIRE(c != nullptr && (dptr != nullptr || bytes == 0), "null pointer");
  DeviceGuard dg(c->device);
  if (bytes) LGP_HIP(hipMemsetAsync(dptr, byte, bytes, c->stream));
  return LGP_OK;
}

int lgp_upload(lgp_ctx *c, void *ddst, const void *hsrc, size_t bytes) {
  LGP_REQUIRE(c != nullptr && ((ddst != nullptr && hsrc != nullptr) || bytes == 0), "null pointer");
  DeviceGuard dg(c->device);
  if (bytes) {
    LGP_HIP(hipMemcpyAsync(ddst, hsrc, bytes, hipMemcpyHostToDevice, c->stream));
    LGP_HIP(hipStreamSynchronize(c->stream));
  }
  return LGP_OK;
}

int lgp_download(lgp_ctx *c, void *hdst, const void *dsrc, size_t bytes) {
  LGP_REQUIRE(c != nullptr && ((hdst != nullptr && dsrc != nullptr) || bytes == 0), "null pointer");
  DeviceGuard dg(c->device);
  if (bytes) {
    LGP_HIP(hipMemcpyAsync(hdst, dsrc, bytes, hipMemcpyDeviceToHost, c->stream));
    LGP_HIP(hipStreamSynchronize(c->stream));
  }
  return LGP_OK;
}

int lgp_copy2d(lgp_ctx *c, void *ddst, size_t ldd, const void *dsrc, size_t lds, size_t rows, size_t cols, size_t elemsize) {
  LGP_REQUIRE(c != nullptr, "null context");
  if (rows == 0 || cols == 0) return LGP_OK;
  LGP_REQUIRE(ddst != nullptr && dsrc != nullptr, "null pointer");
  LGP_REQUIRE(elemsize == 4 || elemsize == 8, "element size must be 4 or 8");
  LGP_REQUIRE(ldd >= cols && lds >= cols, "leading dimension too small");
  DeviceGuard dg(c->device);
  LGP_HIP(hipMemcpy2DAsync(ddst, ldd * elemsize, dsrc, lds * elemsize, cols * elemsize, rows, hipMemcpyDeviceToDevice, c->stream));
  return LGP_OK;
}

int lgp_event_create(lgp_ctx *c, void **event) {
  LGP_REQUIRE(c != nullptr && event != nullptr, "null pointer");
  DeviceGuard dg(c->device);
  hipEvent_t e;
  LGP_HIP(hipEventCreate(&e));
  *event = e;
  return LGP_OK;
}
int lgp_event_destroy(lgp_ctx *c, void *event) {
  LGP_REQUIRE(c != nullptr, "null context");
  if (event) LGP_HIP(hipEventDestroy(static_cast<hipEvent_t>(event)));
  return LGP_OK;
}
int lgp_event_record(lgp_ctx *c, void *event) {
  LGP_REQUIRE(c != nullptr && event != nullptr, "null pointer");
  DeviceGuard dg(c->device);
  LGP_HIP(hipEventRecord(static_cast<hipEvent_t>(event), c->stream));
  return LGP_OK;
}
int lgp_select_stream(lgp_ctx *c, int which) {
  LGP_REQUIRE(c != nullptr, "null context");
  LGP_REQUIRE(which == 0 || which == 1, "stream index must be 0 or 1");
  if ((which == 1) != (c->swapped != 0)) {
    hipStream_t t = c->stream;
    c->stream = c->stream2;
    c->stream2 = t;
    c->swapped = which;
  }
  return LGP_OK;
}
int lgp_stream_wait_event(lgp_ctx *c, void *event) {
  LGP_REQUIRE(c != nullptr && event != nullptr, "null pointer");
  DeviceGuard dg(c->device);
  LGP_HIP(hipStreamWaitEvent(c->stream, static_cast<hipEvent_t>(event), 0));
  return LGP_OK;
}
int lgp_event_elapsed_ms(lgp_ctx *c, void *start, void *stop, float *ms) {
  LGP_REQUIRE(c != nullptr && start != nullptr && stop != nullptr && ms != nullptr, "null pointer");
  DeviceGuard dg(c->device);
  LGP_HIP(hipEventSynchronize(static_cast<hipEvent_t>(stop)));
  LGP_HIP(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
  return LGP_OK;
}

}  // extern "C"

// ---- live timing of the trailing-update launches ----------------------------------------------------
int lgp_prof_start(lgp_ctx *c, hipStream_t stream, double flops) {
  if (!c->prof_on) return LGP_OK;
  if (c->prof_used + 2 > c->prof_events.size()) {
    hipEvent_t a, b;
    LGP_HIP(hipEventCreate(&a));
    LGP_HIP(hipEventCreate(&b));
    c->prof_events.push_back(a);
    c->prof_events.push_back(b);
  }
  c->prof_flops.push_back(flops);
  LGP_HIP(hipEventRecord(c->prof_events[c->prof_used], stream));
  return LGP_OK;
}
int lgp_prof_stop(lgp_ctx *c, hipStream_t stream) {
  if (!c->prof_on) return LGP_OK;
  LGP_HIP(hipEventRecord(c->prof_events[c->prof_used + 1], stream));
  c->prof_used += 2;
  return LGP_OK;
}

extern "C" int lgp_profile_begin(lgp_ctx *c) {
  LGP_REQUIRE(c != nullptr, "null context");
  c->prof_on = 1;
  c->prof_used = 0;
  c->prof_flops.clear();
  return LGP_OK;
}

extern "C" int lgp_profile_end(lgp_ctx *c, size_t *launches, double *total_ms, double *total_flops, double *max_flops, double *max_ms) {
  LGP_REQUIRE(c != nullptr, "null context");
  DeviceGuard dg(c->device);
  c->prof_on = 0;
  LGP_HIP(hipStreamSynchronize(c->stream));
  LGP_HIP(hipStreamSynchronize(c->stream2));
  double ms_sum = 0, fl_sum = 0, fl_max = 0, ms_max = 0;
  const size_t n = c->prof_used / 2;
  for (size_t i = 0; i < n; ++i) {
    float ms = 0;
    LGP_HIP(hipEventElapsedTime(&ms, c->prof_events[2 * i], c->prof_events[2 * i + 1]));
    ms_sum += ms;
    fl_sum += c->prof_flops[i];
    if (c->prof_flops[i] > fl_max) { fl_max = c->prof_flops[i]; ms_max = ms; }
  }
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms_sum;
  if (total_flops) *total_flops = fl_sum;
  if (max_flops) *max_flops = fl_max;
  if (max_ms) *max_ms = ms_max;
  c->prof_used = 0;
  c->prof_flops.clear();
  return LGP_OK;
}

// ---- fused posterior / log density --------------------------------------------------------------

template <typename T>
static int posterior_impl(lgp_ctx *c, size_t n, size_t m, const T *L, size_t ldl, const T *dblk, T *W, size_t ldw, const T *Kxsxs,
                          size_t ldk, T *mean, T *cov, size_t ldc, T *hmlnd) {
  LGP_REQUIRE(c != nullptr, "null context");
  LGP_REQUIRE(n >= 1, "empty training set");
  LGP_REQUIRE(L != nullptr && dblk != nullptr && W != nullptr, "null pointer");
  LGP_REQUIRE(ldw >= m + 1, "W must have m+1 columns");
  DeviceGuard dg(c->device);
  hipStream_t st = c->stream;
  // W <- L^-1 [Kxxs | y]   (the reference solves L^-1 Kxxs twice, _decomp.py:408 and :419; once is enough)
  LGP_CHECK(lgp_trsm_impl<T>(c, st, n, L, ldl, dblk, 0, W, ldw, m + 1));
  if (hmlnd) {
    // minus log density of y for free: column m of W is L^-1 y already           _decomp.py:484-488
    T *d = reinterpret_cast<T *>(c->d_scratch);
    LGP_CHECK(lgp_logdet_device<T>(c, st, n, L, ldl, d));
    LGP_CHECK(lgp_dot_strided_device<T>(c, st, n, W + m, ldw, W + m, ldw, d + 1));
    LGP_HIP(hipMemcpyAsync(c->h_pinned, d, 2 * sizeof(T), hipMemcpyDeviceToHost, st));
  }
  if (m == 0 && !hmlnd) return LGP_OK;
  // mean = (L^-1 Kxxs)^T (L^-1 y)                      _decomp.py:409
  if (mean && m > 0) LGP_CHECK(lgp_gemm_launch<T>(c, st, true, false, m, 1, n, T(1), W, ldw, W + m, ldw, T(0), mean, 1, false));
  // cov = Kxsxs - (L^-1 Kxxs)^T (L^-1 Kxxs)            _decomp.py:420, _compute.py:260
  if (cov && m > 0) {
    LGP_REQUIRE(Kxsxs != nullptr && ldk >= m && ldc >= m, "bad covariance arguments");
    if (Kxsxs != cov) LGP_HIP(hipMemcpy2DAsync(cov, ldc * sizeof(T), Kxsxs, ldk * sizeof(T), m * sizeof(T), m, hipMemcpyDeviceToDevice, st));
    LGP_CHECK(lgp_gemm_launch<T>(c, st, true, false, m, m, n, T(-1), W, ldw, W, ldw, T(1), cov, ldc, false));
  }
  if (hmlnd) {
    LGP_HIP(hipStreamSynchronize(st));
    const T *h = reinterpret_cast<const T *>(c->h_pinned);
    *hmlnd = T(0.5) * (T(n) * T(std::log(2 * M_PI)) + T(2) * h[0] + h[1]);
  }
  return LGP_OK;
}

// posterior from right-hand sides that were carried through the factorisation (lgp_chol_factor_rhs): Wt is
// (m+1) x n, rows 0..m-1 = (L^-1 Kxxs)^T, row m = (L^-1 y)^T
template <typename T>
static int posterior_solved_impl(lgp_ctx *c, size_t n, size_t m, const T *L, size_t ldl, const T *Wt, size_t ldw, const T *Kxsxs,
                                 size_t ldk, T *mean, T *cov, size_t ldc, T *hmlnd) {
  LGP_REQUIRE(c != nullptr, "null context");
  LGP_REQUIRE(n >= 1, "empty training set");
  LGP_REQUIRE(L != nullptr && Wt != nullptr, "null pointer");
  LGP_REQUIRE(ldw >= n && ldl >= n, "leading dimension too small");
  DeviceGuard dg(c->device);
  hipStream_t st = c->stream;
  const T *yrow = Wt + m * ldw;
  if (hmlnd) {
    T *d = reinterpret_cast<T *>(c->d_scratch);
    LGP_CHECK(lgp_logdet_device<T>(c, st, n, L, ldl, d));
    LGP_CHECK(lgp_dot_strided_device<T>(c, st, n, yrow, 1, yrow, 1, d + 1));
    LGP_HIP(hipMemcpyAsync(c->h_pinned, d, 2 * sizeof(T), hipMemcpyDeviceToHost, st));
  }
  // mean = (L^-1 Kxxs)^T (L^-1 y)                      _decomp.py:409
  if (mean && m > 0) LGP_CHECK(lgp_gemm_launch<T>(c, st, false, true, m, 1, n, T(1), Wt, ldw, yrow, ldw, T(0), mean, 1, false));
  // cov = Kxsxs - (L^-1 Kxxs)^T (L^-1 Kxxs)            _decomp.py:420, _compute.py:260
  if (cov && m > 0) {
    LGP_REQUIRE(Kxsxs != nullptr && ldk >= m && ldc >= m, "bad covariance arguments");
    if (Kxsxs != cov) LGP_HIP(hipMemcpy2DAsync(cov, ldc * sizeof(T), Kxsxs, ldk * sizeof(T), m * sizeof(T), m, hipMemcpyDeviceToDevice, st));
    LGP_CHECK(lgp_gemm_launch<T>(c, st, false, true, m, m, n, T(-1), Wt, ldw, Wt, ldw, T(1), cov, ldc, false));
  }
  if (hmlnd) {
    LGP_HIP(hipStreamSynchronize(st));
    const T *h = reinterpret_cast<const T *>(c->h_pinned);
    *hmlnd = T(0.5) * (T(n) * T(std::log(2 * M_PI)) + T(2) * h[0] + h[1]);
  }
  return LGP_OK;
}

template <typename T>
static int mlnd_impl(lgp_ctx *c, size_t n, const T *L, size_t ldl, const T *dblk, T *r, T *hvalue) {
  LGP_REQUIRE(c != nullptr && L != nullptr && dblk != nullptr && r != nullptr && hvalue != nullptr, "null pointer");
  LGP_REQUIRE(n >= 1, "empty matrix");
  DeviceGuard dg(c->device);
  hipStream_t st = c->stream;
  LGP_CHECK(lgp_trsm_impl<T>(c, st, n, L, ldl, dblk, 0, r, 1, 1));          // invLr, _decomp.py:467
  T *d = reinterpret_cast<T *>(c->d_scratch);
  LGP_CHECK(lgp_logdet_device<T>(c, st, n, L, ldl, d));                        // sum(log(diag(L))), :486
  LGP_CHECK(lgp_dot_strided_device<T>(c, st, n, r, 1, r, 1, d + 1));          // invLr @ invLr, :487
  T h[2];
  LGP_HIP(hipMemcpyAsync(h, d, 2 * sizeof(T), hipMemcpyDeviceToHost, st));
  LGP_HIP(hipStreamSynchronize(st));
  // :484-488   1/2 * (len(L) * log(2 pi) + 2 * sum(log(diag(L))) + invLr @ invLr)
  *hvalue = T(0.5) * (T(n) * T(std::log(2 * M_PI)) + T(2) * h[0] + h[1]);
  return LGP_OK;
}

// minus_log_normal_density with forward-mode gradient and Fisher matrix (_decomp.py:461-561) on the device.  Like the
// reference it forms K^-1 = L^-T L^-1 explicitly (:470-472); the Fisher term tr(K^-1 dK_k K^-1 dK_q) is taken as
// <K^-1 dK_k K^-1, dK_q> (two MFMA gemms per parameter + Frobenius products) instead of the reference's two-sided
// triangular solves of every dK (:547-553): the same number, gemm-bound instead of latency-bound.
template <typename T>
static int mlnd_grad_impl(lgp_ctx *c, size_t n, const T *L, size_t ldl, const T *dblk, T *r, size_t p, const T *const *hdK, size_t lddk,
                          const T *ddr, size_t lddr, T *hvalue, T *hgrad, T *hfisher) {
  LGP_REQUIRE(c != nullptr && L != nullptr && dblk != nullptr && r != nullptr, "null pointer");
  LGP_REQUIRE(n >= 1 && p <= 64, "bad sizes (at most 64 parameters)");
  LGP_REQUIRE(p == 0 || hdK != nullptr || ddr != nullptr, "no derivative given");
  LGP_REQUIRE(hdK == nullptr || lddk >= n, "lddk < n");
  LGP_REQUIRE(ddr == nullptr || lddr >= p, "lddr < p");
  DeviceGuard dg(c->device);
  hipStream_t st = c->stream;
  const bool want_grad = hgrad != nullptr && p > 0, want_fisher = hfisher != nullptr && p > 0;
  // scalars (device): [0] sum log L_ii, [1] |L^-1 r|^2, [2 + k] <K^-1, dK_k>, [2 + p + k] z' dK_k z, [2 + 2p + k] z' dr_k,
  // [2 + 3p + k p + q] <K^-1 dK_k K^-1, dK_q>, [2 + 3p + p^2 + k p + q] (L^-1 dr)'(L^-1 dr); partial sums after them
  const size_t nscal = 2 + 3 * p + 2 * p * p;
  LGP_REQUIRE((nscal + 1024) * sizeof(T) <= c->scratch_bytes, "scratch too small");
  T *ds = reinterpret_cast<T *>(c->d_scratch), *dpart = ds + nscal;
  LGP_HIP(hipMemsetAsync(ds, 0, nscal * sizeof(T), st));
  LGP_CHECK(lgp_trsm_impl<T>(c, st, n, L, ldl, dblk, 0, r, 1, 1));            // y = L^-1 r, :467
  LGP_CHECK(lgp_logdet_device<T>(c, st, n, L, ldl, ds));
  LGP_CHECK(lgp_dot_strided_device<T>(c, st, n, r, 1, r, 1, ds + 1));
  T *ws = nullptr;
  WsGuard wsg(c, st, nullptr);
  if (want_grad || want_fisher) {
    const bool needK = hdK != nullptr;
    const size_t nmat = needK ? (want_fisher ? 3 : 2) : 0;
    LGP_CHECK(lgp_ws_alloc(c, st, (nmat * n * n + 2 * n + n * (p ? p : 1)) * sizeof(T), reinterpret_cast<void **>(&ws)));
    wsg.p = ws;
    T *X = ws, *IK = ws + (needK ? n * n : 0), *S = ws + (needK ? 2 * n * n : 0);
    T *z = ws + nmat * n * n, *w = z + n, *V = w + n;
    LGP_HIP(hipMemcpyAsync(z, r, n * sizeof(T), hipMemcpyDeviceToDevice, st));
    LGP_CHECK(lgp_trsm_impl<T>(c, st, n, L, ldl, dblk, 1, z, 1, 1));          // z = K^-1 r, :469
    if (needK) {
      // X = L^-1 (forward solve of the identity), K^-1 = X' X                  :470-472
      LGP_HIP(hipMemsetAsync(X, 0, n * n * sizeof(T), st));
      LGP_CHECK(lgp_set_diag_device<T>(c, st, n, X, n, T(1)));
      LGP_CHECK(lgp_trsm_impl<T>(c, st, n, L, ldl, dblk, 0, X, n, n));
      LGP_CHECK(lgp_gemm_launch<T>(c, st, true, false, n, n, n, T(1), X, n, X, n, T(0), IK, n, false));
      for (size_t k = 0; k < p; ++k) {
        LGP_REQUIRE(hdK[k] != nullptr, "null derivative matrix");
        LGP_CHECK(lgp_frob_device<T>(c, st, n, n, IK, n, hdK[k], lddk, dpart, ds + 2 + k));                  // tr(K^-1 dK_k), :526
        LGP_CHECK(lgp_gemm_launch<T>(c, st, false, false, n, 1, n, T(1), hdK[k], lddk, z, 1, T(0), w, 1, false));
        LGP_CHECK(lgp_dot_strided_device<T>(c, st, n, z, 1, w, 1, ds + 2 + p + k));                          // z' dK_k z, :527
      }
      if (want_fisher) {
        for (size_t k = 0; k < p; ++k) {
          LGP_CHECK(lgp_gemm_launch<T>(c, st, false, false, n, n, n, T(1), hdK[k], lddk, IK, n, T(0), X, n, false));   // dK_k K^-1
          LGP_CHECK(lgp_gemm_launch<T>(c, st, false, false, n, n, n, T(1), IK, n, X, n, T(0), S, n, false));          // K^-1 dK_k K^-1
          for (size_t q = k; q < p; ++q)
            LGP_CHECK(lgp_frob_device<T>(c, st, n, n, S, n, hdK[q], lddk, dpart, ds + 2 + 3 * p + k * p + q));
        }
      }
    }
    if (ddr != nullptr) {
      for (size_t k = 0; k < p; ++k) LGP_CHECK(lgp_dot_strided_device<T>(c, st, n, z, 1, ddr + k, lddr, ds + 2 + 2 * p + k));   // z' dr_k, :532
      if (want_fisher) {
        LGP_HIP(hipMemcpy2DAsync(V, p * sizeof(T), ddr, lddr * sizeof(T), p * sizeof(T), n, hipMemcpyDeviceToDevice, st));
        LGP_CHECK(lgp_trsm_impl<T>(c, st, n, L, ldl, dblk, 0, V, p, p));                                     // L^-1 dr, :556
        LGP_CHECK(lgp_gemm_launch<T>(c, st, true, false, p, p, n, T(1), V, p, V, p, T(0), ds + 2 + 3 * p + p * p, p, false));
      }
    }
  }
  std::vector<T> h(nscal);
  LGP_HIP(hipMemcpyAsync(h.data(), ds, nscal * sizeof(T), hipMemcpyDeviceToHost, st));
  LGP_HIP(hipStreamSynchronize(st));
  if (hvalue) *hvalue = T(0.5) * (T(n) * T(std::log(2 * M_PI)) + T(2) * h[0] + h[1]);                       // :484-488
  if (want_grad)
    for (size_t k = 0; k < p; ++k) {
      T gk = T(0);
      if (hdK) gk += T(0.5) * (h[2 + k] - h[2 + p + k]);                                                     // :528
      if (ddr) gk += h[2 + 2 * p + k];                                                                        // :533
      hgrad[k] = gk;
    }
  if (want_fisher)
    for (size_t k = 0; k < p; ++k)
      for (size_t q = 0; q < p; ++q) {
        T f = T(0);
        if (hdK) f += T(0.5) * (k <= q ? h[2 + 3 * p + k * p + q] : h[2 + 3 * p + q * p + k]);               // :554
        if (ddr) f += h[2 + 3 * p + p * p + k * p + q];                                                       // :558
        hfisher[k * p + q] = f;
      }
  return LGP_OK;
}

extern "C" {
int lgp_mlnd_grad_f64(lgp_ctx *c, size_t n, const double *L, size_t ldl, const double *dblk, double *r, size_t p, const double *const *hdK,
                      size_t lddk, const double *ddr, size_t lddr, double *hvalue, double *hgrad, double *hfisher) {
  return mlnd_grad_impl<double>(c, n, L, ldl, dblk, r, p, hdK, lddk, ddr, lddr, hvalue, hgrad, hfisher);
}
int lgp_mlnd_grad_f32(lgp_ctx *c, size_t n, const float *L, size_t ldl, const float *dblk, float *r, size_t p, const float *const *hdK,
                      size_t lddk, const float *ddr, size_t lddr, float *hvalue, float *hgrad, float *hfisher) {
  return mlnd_grad_impl<float>(c, n, L, ldl, dblk, r, p, hdK, lddk, ddr, lddr, hvalue, hgrad, hfisher);
}
int lgp_posterior_f64(lgp_ctx *c, size_t n, size_t m, const double *L, size_t ldl, const double *dblk, double *W, size_t ldw,
                      const double *K, size_t ldk, double *mean, double *cov, size_t ldc, double *hmlnd) {
  return posterior_impl<double>(c, n, m, L, ldl, dblk, W, ldw, K, ldk, mean, cov, ldc, hmlnd);
}
int lgp_posterior_f32(lgp_ctx *c, size_t n, size_t m, const float *L, size_t ldl, const float *dblk, float *W, size_t ldw,
                      const float *K, size_t ldk, float *mean, float *cov, size_t ldc, float *hmlnd) {
  return posterior_impl<float>(c, n, m, L, ldl, dblk, W, ldw, K, ldk, mean, cov, ldc, hmlnd);
}
int lgp_posterior_solved_f64(lgp_ctx *c, size_t n, size_t m, const double *L, size_t ldl, const double *Wt, size_t ldw,
                             const double *K, size_t ldk, double *mean, double *cov, size_t ldc, double *hmlnd) {
  return posterior_solved_impl<double>(c, n, m, L, ldl, Wt, ldw, K, ldk, mean, cov, ldc, hmlnd);
}
int lgp_posterior_solved_f32(lgp_ctx *c, size_t n, size_t m, const float *L, size_t ldl, const float *Wt, size_t ldw,
                             const float *K, size_t ldk, float *mean, float *cov, size_t ldc, float *hmlnd) {
  return posterior_solved_impl<float>(c, n, m, L, ldl, Wt, ldw, K, ldk, mean, cov, ldc, hmlnd);
}
// The plain question of GP.predfromdata -- one isotropic kernel, n data points, m prediction points, diagonal noise -- from host
// buffers to host results in one call: two uploads, the cross blocks straight into the rows carried below the factor, the fused
// Gram build + regularisation + factorisation, the posterior, ONE download.  The same kernels in the same order as the separate
// entry points (lgp_gram_isotropic, lgp_gram_chol_factor_rhs, lgp_posterior_solved): what it saves is host round trips, which are
// all that a problem of a thousand points costs (docs/userguide/optim.rst:88-159).
int lgp_predict_isotropic_f64(lgp_ctx *c, int kernel, int p, int npar, const double *hpar, int d, const double *hpts, size_t n, size_t m,
                              int has_loc, double loc, int has_scale, double scale, int nops, const int *hops_code, const double *hops_val,
                              const double *hy, const double *hnoise, double *dP, double *dL, size_t ldl, size_t nx, double *ddblk, double *dwork,
                              double *hout, double *heps_out) {
  LGP_REQUIRE(c != nullptr && hpts != nullptr && hy != nullptr && dP != nullptr && dL != nullptr && ddblk != nullptr && dwork != nullptr && hout != nullptr,
              "null pointer");
  LGP_REQUIRE(n >= 1 && m >= 1 && d >= 1, "empty problem");
  LGP_REQUIRE(nx >= m + 1 && ldl >= n, "the carried rows: nx >= m + 1, ldl >= n");
  DeviceGuard dg(c->device);
  hipStream_t st = c->stream;
  if (lgp_predict_small_fits(kernel, d, n, m)) {
    // a small problem: one launch of one workgroup does all of it, inputs and results through mapped host memory (gram.hip)
    int info = 0;
    double *dnz = dwork + m * m + m + 2;
    LGP_CHECK(lgp_predict_small_f64(c, kernel, p, npar, hpar, d, hpts, n, m, has_loc, loc, has_scale, scale, nops, hops_code, hops_val, hy, hnoise, dP, dL, ldl,
                                    nx, ddblk, dnz, hout, heps_out, &info));
    if (info != 0) return info;
    double *h = hout + m * m + m;                               // _decomp.py:484-488
    h[0] = 0.5 * ((double)n * std::log(2 * M_PI) + 2.0 * h[0] + h[1]);
    return LGP_OK;
  }
  // the uploads below read the caller's pageable host buffers asynchronously: no path out of this function may leave them in flight
  struct Drain {
    hipStream_t s;
    ~Drain() { (void)hipStreamSynchronize(s); }
  } drain{st};
  const size_t np = n + m;
  double *Wt = dL + n * ldl;                                    // (m + 1) x n: Kxxs^T, then y
  double *dcov = dwork, *dmean = dwork + m * m, *dsc = dmean + m, *dnoise = dsc + 2;
  LGP_HIP(hipMemcpyAsync(dP, hpts, (size_t)d * np * sizeof(double), hipMemcpyHostToDevice, st));
  LGP_HIP(hipMemcpyAsync(Wt + m * ldl, hy, n * sizeof(double), hipMemcpyHostToDevice, st));
  if (hnoise) LGP_HIP(hipMemcpyAsync(dnoise, hnoise, n * sizeof(double), hipMemcpyHostToDevice, st));
  if (nx > m + 1) LGP_HIP(hipMemset2DAsync(Wt + (m + 1) * ldl, ldl * sizeof(double), 0, n * sizeof(double), nx - m - 1, st));
  const double *dx = dP, *dxs = dP + n;                         // coordinate f of point i at dP[f * np + i]
  LGP_CHECK(lgp_gram_isotropic_f64(c, kernel, p, npar, hpar, d, dxs, np, m, dx, np, n, has_loc, loc, has_scale, scale, nops, hops_code, hops_val, 0, Wt, ldl));
  LGP_CHECK(lgp_gram_isotropic_f64(c, kernel, p, npar, hpar, d, dxs, np, m, dxs, np, m, has_loc, loc, has_scale, scale, nops, hops_code, hops_val, 1, dcov, m));
  // ONE synchronisation per prediction: the factorisation leaves its read-back (eps, the pivot verdict) and the diagonal-block
  // images to this function -- the posterior needs neither, its solve rode through the factorisation
  int fst;
  {
    struct Defer {
      lgp_ctx *c;
      explicit Defer(lgp_ctx *ctx) : c(ctx) { c->defer_finish = 1; }
      ~Defer() { c->defer_finish = 0; }
    } defer(c);
    fst = lgp_gram_chol_factor_rhs_f64(c, kernel, p, npar, hpar, d, dx, np, n, has_loc, loc, has_scale, scale, nops, hops_code, hops_val, 0.0,
                                       hnoise ? dnoise : nullptr, 0.0, 0.0, LGP_EPS_AUTO_REL | LGP_FACTOR_LOWER_ONLY, dL, ldl, nx, ddblk, nullptr);
  }
  if (fst != 0) return fst;                                     // an error (< 0)
  LGP_CHECK(posterior_solved_impl<double>(c, n, m, dL, ldl, Wt, ldl, dcov, m, dmean, dcov, m, nullptr));
  LGP_CHECK(lgp_logdet_device<double>(c, st, n, dL, ldl, dsc));
  LGP_CHECK(lgp_dot_strided_device<double>(c, st, n, Wt + m * ldl, 1, Wt + m * ldl, 1, dsc + 1));
  LGP_HIP(hipMemcpyAsync(hout, dwork, (m * m + m + 2) * sizeof(double), hipMemcpyDeviceToHost, st));
  LGP_HIP(hipStreamSynchronize(st));
  const double *hp = static_cast<const double *>(c->h_pinned);
  int info = 0;
  memcpy(&info, hp + 64, sizeof(int));
  if (heps_out) *heps_out = hp[33];
  if (const int verdict = lgp_info_result(c, info); verdict != 0) return verdict;      // a stall (< 0) or the first bad pivot (> 0)
  LGP_CHECK(lgp_extract_dblk_public(c, st, dL, ldl, n, ddblk));  // (for the solves of later questions: enqueued, not waited for)
  double *h = hout + m * m + m;                                 // _decomp.py:484-488
  h[0] = 0.5 * ((double)n * std::log(2 * M_PI) + 2.0 * h[0] + h[1]);
  return LGP_OK;
}
int lgp_mlnd_value_f64(lgp_ctx *c, size_t n, const double *L, size_t ldl, const double *dblk, double *r, double *h) {
  return mlnd_impl<double>(c, n, L, ldl, dblk, r, h);
}
int lgp_mlnd_value_f32(lgp_ctx *c, size_t n, const float *L, size_t ldl, const float *dblk, float *r, float *h) {
  return mlnd_impl<float>(c, n, L, ldl, dblk, r, h);
}
}

// ---- RCCL (loaded lazily: single-GPU use never touches librccl) -----------------------------------
namespace {
struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int rccl_load() {
  if (g_rccl.handle) return LGP_OK;
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    lgp_set_error("cannot load librccl.so: %s", dlerror());
    return LGP_ECOMM;
  }
#define LGP_SYM(field, name)                                              \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
  if (!g_rccl.field) {                                                    \
    lgp_set_error("librccl.so lacks %s", name);                          \
    dlclose(h);                                                           \
    return LGP_ECOMM;                                                     \
  }
  LGP_SYM(GetUniqueId, "ncclGetUniqueId")
  LGP_SYM(CommInitRank, "ncclCommInitRank")
  LGP_SYM(CommDestroy, "ncclCommDestroy")
  LGP_SYM(CommCount, "ncclCommCount")
  LGP_SYM(CommUserRank, "ncclCommUserRank")
  LGP_SYM(Broadcast, "ncclBroadcast")
  LGP_SYM(AllReduce, "ncclAllReduce")
  LGP_SYM(Send, "ncclSend")
  LGP_SYM(Recv, "ncclRecv")
  LGP_SYM(GroupStart, "ncclGroupStart")
  LGP_SYM(GroupEnd, "ncclGroupEnd")
  LGP_SYM(GetErrorString, "ncclGetErrorString")
#undef LGP_SYM
  g_rccl.handle = h;
  return LGP_OK;
}

#define LGP_NCCL(expr)                                                                       \
  do {                                                                                       \
    ncclResult_t _r = (expr);                                                                \
    if (_r != ncclSuccess) {                                                                 \
      lgp_set_error("%s failed: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?"); \
      return LGP_ECOMM;                                                                      \
    }                                                                                        \
  } while (0)

struct lgp_xfer { int src, dst; size_t off, len; };
}  // namespace

extern "C" {

int lgp_comm_unique_id(void *hid128) {
  LGP_REQUIRE(hid128 != nullptr, "null pointer");
  static_assert(sizeof(ncclUniqueId) <= LGP_COMM_ID_BYTES, "unique id does not fit");
  LGP_CHECK(rccl_load());
  ncclUniqueId id;
  LGP_NCCL(g_rccl.GetUniqueId(&id));
  memset(hid128, 0, LGP_COMM_ID_BYTES);
  memcpy(hid128, &id, sizeof(id));
  return LGP_OK;
}

int lgp_comm_init(lgp_ctx *c, int rank, int nranks, const void *hid128) {
  LGP_REQUIRE(c != nullptr && hid128 != nullptr, "null pointer");
  LGP_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank");
  LGP_REQUIRE(c->comm == nullptr, "communicator already initialised");
  LGP_REQUIRE(c->hostfn == nullptr, "a host-staged transport is active on this context (lgp_comm_destroy it first)");
  LGP_CHECK(rccl_load());
  DeviceGuard dg(c->device);
  ncclUniqueId id;
  memcpy(&id, hid128, sizeof(id));
  ncclComm_t comm;
  LGP_NCCL(g_rccl.CommInitRank(&comm, nranks, id, rank));
  c->comm = comm;
  c->rank = rank;
  c->nranks = nranks;
  return LGP_OK;
}

int lgp_comm_destroy(lgp_ctx *c) {
  LGP_REQUIRE(c != nullptr, "null context");
  if (c->comm && g_rccl.handle) {
    DeviceGuard dg(c->device);
    (void)g_rccl.CommDestroy(static_cast<ncclComm_t>(c->comm));
  }
  c->comm = nullptr;
  c->hostfn = nullptr;      // a destroyed host transport must not keep intercepting lgp_bcast / lgp_allreduce_*
  c->hostuser = nullptr;
  c->rank = 0;
  c->nranks = 1;
  return LGP_OK;
}

int lgp_comm_info(lgp_ctx *c, int *rank, int *nranks, int *transport) {
  LGP_REQUIRE(c != nullptr, "null context");
  int r = c->rank, n = c->nranks, t = LGP_TRANSPORT_NONE;
  if (c->comm) {          // what the communicator itself says, not what it was asked for
    DeviceGuard dg(c->device);
    LGP_NCCL(g_rccl.CommCount(static_cast<ncclComm_t>(c->comm), &n));
    LGP_NCCL(g_rccl.CommUserRank(static_cast<ncclComm_t>(c->comm), &r));
    t = LGP_TRANSPORT_RCCL;
  } else if (c->hostfn) {
    t = LGP_TRANSPORT_HOST;
  }
  if (rank) *rank = r;
  if (nranks) *nranks = n;
  if (transport) *transport = t;
  return LGP_OK;
}

// host-staged transport: device -> host copy on the current stream, the caller's collective, host -> device copy
static int host_staged(lgp_ctx *c, void *dbuf, size_t bytes, const std::function<int(char *)> &body) {
  DeviceGuard dg(c->device);
  void *h = malloc(bytes);
  LGP_REQUIRE(h != nullptr, "out of host memory for the staged collective");
  hipError_t e = hipMemcpyAsync(h, dbuf, bytes, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  int rc = LGP_OK;
  if (e != hipSuccess) {
    lgp_set_error("staged collective: %s", hipGetErrorString(e));
    rc = LGP_EHIP;
  } else if ((rc = body(static_cast<char *>(h))) == LGP_OK) {
    e = hipMemcpyAsync(dbuf, h, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
      lgp_set_error("staged collective: %s", hipGetErrorString(e));
      rc = LGP_EHIP;
    }
  }
  free(h);
  return rc;
}

static int host_collective(lgp_ctx *c, int op, void *dbuf, size_t count, size_t bytes, int root) {
  return host_staged(c, dbuf, bytes, [&](char *h) {
    if (c->hostfn(c->hostuser, op, h, count, root) == 0) return (int)LGP_OK;
    lgp_set_error("host collective callback failed (op %d)", op);
    return (int)LGP_ECOMM;
  });
}

// ---- broadcast as scatter + all-gather.  The GPUs of a node are joined pairwise (xGMI: 7 links per GPU, no switch): a ring or
// tree broadcast moves the whole buffer over ONE link of every GPU on the way, S / B_link, while six links idle.  Here the buffer
// is cut into G - 1 chunks, one per peer of the root: the root sends chunk p to peer p (G - 1 links, S / (G - 1) each), then
// every peer sends its chunk to every other peer (each link of each peer again carries S / (G - 1)):  2 S / ((G - 1) B_link).
// The same table of transfers drives RCCL (grouped ncclSend / ncclRecv) and the host-staged transport (op 5 of the callback).
static int lgp_bcast_xfers(size_t bytes, int nranks, int root, std::vector<lgp_xfer> &scatter, std::vector<lgp_xfer> &gather) {
  scatter.clear();
  gather.clear();
  if (nranks < 2 || bytes == 0) return 0;
  const size_t P = (size_t)nranks - 1;
  size_t chunk = (bytes + P - 1) / P;
  chunk = (chunk + 511) / 512 * 512;                       // (whole 512-byte runs: the rows of a panel slot start on them)
  int used = 0;
  for (size_t p = 0; p < P; ++p) {
    const size_t off = p * chunk;
    if (off >= bytes) break;
    const size_t len = std::min(chunk, bytes - off);
    const int holder = (root + 1 + (int)p) % nranks;
    scatter.push_back({root, holder, off, len});
    ++used;
  }
  for (int p = 0; p < used; ++p)
    for (int q = 0; q < (int)P; ++q) {                     // (peers beyond the last chunk hold nothing but still receive everything)
      const int dst = (root + 1 + q) % nranks;
      if (dst != scatter[p].dst) gather.push_back({scatter[p].dst, dst, scatter[p].off, scatter[p].len});
    }
  return used;
}

int lgp_comm_init_host(lgp_ctx *c, int rank, int nranks, int (*fn)(void *, int, void *, size_t, int), void *user) {
  LGP_REQUIRE(c != nullptr && fn != nullptr, "null pointer");
  LGP_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank");
  LGP_REQUIRE(c->comm == nullptr, "an RCCL communicator is already initialised");
  LGP_REQUIRE(c->hostfn == nullptr, "a host-staged transport is already active on this context");
  c->hostfn = fn;
  c->hostuser = user;
  c->rank = rank;
  c->nranks = nranks;
  return LGP_OK;
}

static bool bcast_scatters(const lgp_ctx *c, size_t bytes) {
  if (c->nranks < 3 || c->bcast_algo == 1) return false;
  if (c->bcast_algo == 2) return true;
  return c->comm != nullptr && bytes >= c->bcast_min;      // auto: a host transport is only asked for op 5 when told so
}

int lgp_comm_set_bcast(lgp_ctx *c, int algo, size_t min_bytes) {
  LGP_REQUIRE(c != nullptr, "null context");
  LGP_REQUIRE(algo >= 0 && algo <= 2, "algo: 0 auto, 1 plain, 2 scatter + all-gather");
  c->bcast_algo = algo;
  c->bcast_min = min_bytes;
  return LGP_OK;
}

long lgp_bcast_plan(size_t bytes, int nranks, int root, unsigned long long *xfers, size_t capacity, size_t *nscatter) {
  if (nranks < 1 || root < 0 || root >= nranks) {
    lgp_set_error("invalid argument: bad root");
    return LGP_EINVAL;
  }
  std::vector<lgp_xfer> sc, ga;
  lgp_bcast_xfers(bytes, nranks, root, sc, ga);
  if (nscatter) *nscatter = sc.size();
  size_t t = 0;
  for (const auto *v : {&sc, &ga})
    for (const lgp_xfer &x : *v) {
      if (xfers != nullptr && t < capacity) {
        xfers[4 * t] = (unsigned long long)x.src; xfers[4 * t + 1] = (unsigned long long)x.dst; xfers[4 * t + 2] = x.off; xfers[4 * t + 3] = x.len;
      }
      ++t;
    }
  return (long)t;
}

int lgp_bcast(lgp_ctx *c, void *dbuf, size_t bytes, int root) {
  LGP_REQUIRE(c != nullptr && (dbuf != nullptr || bytes == 0), "null pointer");
  LGP_REQUIRE(root >= 0 && root < c->nranks, "bad root");
  const bool scatter = bytes != 0 && bcast_scatters(c, bytes);
  std::vector<lgp_xfer> phase[2];
  if (scatter) lgp_bcast_xfers(bytes, c->nranks, root, phase[0], phase[1]);
  if (bytes && c->hostfn) {
    if (!scatter) return host_collective(c, 0, dbuf, bytes, bytes, root);
    // every rank walks the whole table (the callback decides who moves what): op 5, root = src | dst << 16
    return host_staged(c, dbuf, bytes, [&](char *h) {
      for (const auto &ph : phase)
        for (const lgp_xfer &x : ph)
          if (c->hostfn(c->hostuser, 5, h + x.off, x.len, x.src | (x.dst << 16)) != 0) {
            lgp_set_error("host collective callback failed (op 5: %d -> %d)", x.src, x.dst);
            return (int)LGP_ECOMM;
          }
      return (int)LGP_OK;
    });
  }
  if (bytes == 0 || (c->comm == nullptr && c->nranks == 1)) return LGP_OK;   // no communicator: single process
  LGP_REQUIRE(c->comm != nullptr, "communicator not initialised");
  DeviceGuard dg(c->device);
  ncclComm_t comm = static_cast<ncclComm_t>(c->comm);
  if (!scatter) {
    LGP_NCCL(g_rccl.Broadcast(dbuf, dbuf, bytes, ncclUint8, root, comm, c->stream));
    return LGP_OK;
  }
  // two groups on the one stream: the second starts when this rank's chunk has arrived.  A group must be closed on every path
  char *base = static_cast<char *>(dbuf);
  for (const auto &ph : phase) {
    LGP_NCCL(g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    for (const lgp_xfer &x : ph) {
      if (r != ncclSuccess) break;
      if (x.src == c->rank) r = g_rccl.Send(base + x.off, x.len, ncclUint8, x.dst, comm, c->stream);
      else if (x.dst == c->rank) r = g_rccl.Recv(base + x.off, x.len, ncclUint8, x.src, comm, c->stream);
    }
    const ncclResult_t rend = g_rccl.GroupEnd();
    LGP_NCCL(r);
    LGP_NCCL(rend);
  }
  return LGP_OK;
}

int lgp_allreduce_sum_f64(lgp_ctx *c, double *dbuf, size_t count) {
  LGP_REQUIRE(c != nullptr && (dbuf != nullptr || count == 0), "null pointer");
  if (count && c->hostfn) return host_collective(c, 1, dbuf, count, count * sizeof(double), 0);
  if (count == 0 || (c->comm == nullptr && c->nranks == 1)) return LGP_OK;
  LGP_REQUIRE(c->comm != nullptr, "communicator not initialised");
  DeviceGuard dg(c->device);
  LGP_NCCL(g_rccl.AllReduce(dbuf, dbuf, count, ncclDouble, ncclSum, static_cast<ncclComm_t>(c->comm), c->stream));
  return LGP_OK;
}

int lgp_allreduce_max_f64(lgp_ctx *c, double *dbuf, size_t count) {
  LGP_REQUIRE(c != nullptr && (dbuf != nullptr || count == 0), "null pointer");
  if (count && c->hostfn) return host_collective(c, 2, dbuf, count, count * sizeof(double), 0);
  if (count == 0 || (c->comm == nullptr && c->nranks == 1)) return LGP_OK;
  LGP_REQUIRE(c->comm != nullptr, "communicator not initialised");
  DeviceGuard dg(c->device);
  LGP_NCCL(g_rccl.AllReduce(dbuf, dbuf, count, ncclDouble, ncclMax, static_cast<ncclComm_t>(c->comm), c->stream));
  return LGP_OK;
}

}  // extern "C"

// chaint.hip -- the chain kernel of round 3: the diagonal squares of all panels factored by resident worker workgroups as a task graph
// (P: factor a diagonal block, S: substitution of 64 rows, G: update of a 64 x 64 tile).  Round 4 built a replacement whose first
// workgroup keeps the diagonal block on chip; measured side by side on one box it was 9 % SLOWER up to N = 4096 (1.28 vs 1.17 ms
// at N = 2048, 2.57 vs 2.35 at 4096, 6.43 vs 6.23 at 8192; equal from 16384 on).  It was retired in round 5: sources, host test
// and trace tools under docs/history/walker.  Driven by potrf_chain (potrf.hip) through the flag words of common.h (LGP_CH_*).
#include "potf2_dev.h"
#define P4_QUAD_ARG (LGP_P4_QUAD != 0)      // the workers have their CUs to themselves: the four pivots of a register together (potf2_dev.h)
#include <algorithm>
#include <cstdlib>

namespace {
// ------------------------------------------------------------------------------------------------
// The dependent chain of the WHOLE factorisation as one launch: resident worker workgroups that own their CUs.
//
// What bounds the factorisation below N ~ 16384 is the chain  potf2 -> rows of the next diagonal block solved -> that block
// updated -> potf2.  As separate launches on a high-priority stream its kernels share CUs, matrix pipes and LDS with the
// trailing update: the kernel trace of N = 8192 (tools/timeline.py) shows potf2 at 48 us while the GPU is otherwise idle and
// at 130-170 us beside a trailing update, the substitution at 18 vs 44-104 us.  CU-masked streams keep the trailing update
// off some CUs but slow every masked launch by 13-30 % (measured, LGP_RESERVE).  So the chain gets CUs of its own the only
// way left: W workgroups are launched once, before anything else, each with enough LDS that no trailing-update workgroup
// fits beside it, and stay for the whole factorisation.  They execute the diagonal SQUARES of all panels as a task graph on
// 64-row units r and 64-column units c of panel k:
//     P(k, j)        factor diagonal block j (the four-wave kernel above), write its fragment image
//     S(k, r, j)     X(r, j) <- A(r, j) L_jj^-T for the 64 rows of unit r below block j (subst_right_wave)
//     G(k, r, c, j)  A(r, c) -= X(r, j) X(c, j)^T, a 64 x 64 tile with K = 128, c a unit right of block j, r >= c
// with  S(r, j) <- P(j), G(r, 2j | 2j+1, i < j);  G(r, c, j) <- S(r, j), S(c, j), G(r, c, j - 1);  P(j) <- G(2j | 2j+1, 2j | 2j+1, j - 1),
// kept as flags and per-tile step counters in device memory.  Workers take tasks from a list in topological order
// (lgp_chaint_schedule: every block's chain first, the other rows of the square lagging behind) with an atomic counter, wait
// for the task's inputs (agent-scope acquire), run it, publish it (release).  A task is only ever held by a running
// workgroup and only waits for tasks earlier in the list, which have all been taken: the earliest unfinished task can always
// run -- no deadlock, whatever the dispatch order.
// Everything else flows around the chain on ordinary streams (potrf.hip, potrf_chain), tied to it by memory flags that the
// command processor polls or writes (hipStreamWaitValue32 / hipStreamWriteValue32: 1.7 us from a flag store of a running
// kernel to the first wave of the waiting stream's next kernel, tools/waitvalue_probe.hip):
//     ready[k][j]  (kernel -> stream)  block j factored and the square's rows solved against it: the far-row stream may run
//                                      its substitution and K = 128 update of the rows below the square
//     u1[k]        (stream -> kernel)  the square of panel k has received the updates of all earlier panels
// Every wait in the kernel is bounded: after CH_TIMEOUT a worker raises the abort flag, releases every ready[] word so that
// no stream is left waiting, everybody leaves, and the host reports LGP_INFO_STALLED instead of hanging.
constexpr unsigned CH_TYPE_P = 0, CH_TYPE_S = 1, CH_TYPE_G = 2;
constexpr unsigned long long CH_TIMEOUT = 1000000000ull;      // wall_clock64 ticks (100 MHz): 10 s per wait (env LGP_CHAIN_TIMEOUT_MS)

struct ChainArgs {
  double *L;              // the matrix (lower triangle of K/s/s' on entry, the factor on exit)
  long ldl;
  int n;                  // a multiple of 128
  double *dblk;           // diagonal-block workspaces, one per 128 columns
  int *info;
  const uint2 *tasks;     // x: type | j << 8 | c << 16 | k << 24, y: r;  the chain list first, then the bulk list
  int nchain, ntasks;     // tasks [0, nchain): the dependent chain, taken by the first `chainw` workers; [nchain, ntasks): the rest
  int chainw;
  unsigned *state;        // LGP_CH_* layout (common.h), zeroed by the driver
  unsigned long long *trace;   // optional (LGP_CHAIN_TRACE): per task  worker, taken, inputs ready, done  (wall_clock64)
  unsigned long long timeout;  // wall_clock64 ticks a worker waits for one input before it abandons the launch
  long row0;                   // global index of the matrix's first row (pivot indices of the info word)
  int samexcd;                 // chain workers all on XCD 0 (experiment, env LGP_CHAIN_SAMEXCD)
};

__device__ __forceinline__ unsigned ch_peek(unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// thread 0 of the workgroup: spin until *p >= need; false = the launch is being abandoned
__device__ __forceinline__ bool ch_wait(unsigned *p, unsigned need, unsigned *state, unsigned long long timeout) {
  unsigned spins = 0;
  unsigned long long t0 = 0;
  while (ch_peek(p) < need) {
    __builtin_amdgcn_s_sleep(2);
    if ((++spins & 63u) == 0) {
      if (ch_peek(state + LGP_CH_ABORT) != 0) return false;
      const unsigned long long now = wall_clock64();
      if (t0 == 0) t0 = now;
      if (now - t0 > timeout) {
        __hip_atomic_store(state + LGP_CH_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return false;
      }
    }
  }
  return true;
}

// G(r, c, j): the 64 x 64 tile at (64 r, 64 c) of the panel -= X(r, j) X(c, j)^T.  Four waves, 32 x 32 each; both operands are
// "row fragments" (lane (n, g) holds row n, k = g of a 16 x 4 slab) read straight from global memory two k at a time (any
// k order serves as long as both operands use the same one); the C tile starts in the accumulators, negated.
__device__ __forceinline__ void ch_update_tile(double *P, long ldp, int j, int r, int c) {
  typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
  typedef unsigned int u2_t __attribute__((ext_vector_type(2)));
  typedef double d2_t __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wy = w >> 1, wx = w & 1;
  const bool diag = r == c;
  if (diag && wy < wx) return;                           // above the diagonal
  const long Ra = 64L * r + 32 * wy, Rb = 64L * c + 32 * wx;
  constexpr int OOB = 0x7ffffff0;
  const __amdgpu_buffer_rsrc_t abuf = __builtin_amdgcn_make_buffer_rsrc(P + Ra * ldp + 128 * j, 0, 0x7fffff00, 0x00020000);
  const __amdgpu_buffer_rsrc_t bbuf = __builtin_amdgcn_make_buffer_rsrc(P + Rb * ldp + 128 * j, 0, 0x7fffff00, 0x00020000);
  const __amdgpu_buffer_rsrc_t cbuf = __builtin_amdgcn_make_buffer_rsrc(P + Ra * ldp + Rb, 0, 0x7fffff00, 0x00020000);
  const int ldb8 = (int)(ldp * 8);
  const int vab = n * ldb8 + g * 16;                     // operand row n of a 16-row slab, k pair g
  const int vc = g * ldb8 + n * 8;                       // accumulator register q of a tile: row 4 q + g, column n
  acc_t acc[2][2];
  const bool skip01 = diag && wy == wx;                  // tile (0, 1) of a wave on the diagonal lies above it
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc[ti][tj][q] = -__builtin_bit_cast(double, (u2_t)__builtin_amdgcn_raw_buffer_load_b64(cbuf, vc, (16 * ti + 4 * q) * ldb8 + tj * 128, 0));
  auto lda2 = [&](const __amdgpu_buffer_rsrc_t &buf, int soff) -> d2_t {
    return __builtin_bit_cast(d2_t, (u4_t)__builtin_amdgcn_raw_buffer_load_b128(buf, vab, soff, 0));
  };
  d2_t a[2][2][4], b[2][2][4];                            // [buffer][tile][u]
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      a[0][t][u] = lda2(abuf, t * 16 * ldb8 + u * 64);
      b[0][t][u] = lda2(bbuf, t * 16 * ldb8 + u * 64);
    }
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const int cur = qq & 1, nxt = cur ^ 1;
    if (qq < 3) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          a[nxt][t][u] = lda2(abuf, t * 16 * ldb8 + ((qq + 1) * 4 + u) * 64);
          b[nxt][t][u] = lda2(bbuf, t * 16 * ldb8 + ((qq + 1) * 4 + u) * 64);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
          for (int tj = 0; tj < 2; ++tj) {
            if (skip01 && ti == 0 && tj == 1) continue;
            acc[ti][tj] = MmaD::run(a[cur][ti][u][h], b[cur][tj][u][h], acc[ti][tj]);
          }
  }
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      if (skip01 && ti == 0 && tj == 1) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool keep = !(skip01 && ti == tj) || (4 * q + g >= n);      // diagonal 16 x 16 tile: row >= column
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, -acc[ti][tj][q]), cbuf, keep ? vc : OOB, (16 * ti + 4 * q) * ldb8 + tj * 128, 0);
      }
    }
}

__global__ void __launch_bounds__(256) chain_kernel(ChainArgs p) {
  __shared__ double rowbuf[P4_ROWBUF];
  __shared__ double diagbuf[4 * 64];
  __shared__ int s_task;
  __builtin_amdgcn_s_setprio(LGP_CHAIN_PRIO);
  unsigned *const st = p.state;
  const int npanels = (p.n + LGP_NB - 1) / LGP_NB;
  // the workers of the dependent chain: the first `chainw` workgroups, or (samexcd) the first chainw of those that the hardware
  // places on XCD 0 (workgroup i runs on XCD i mod 8), so that what one of them writes the next finds in the same L2
  bool on_chain = p.samexcd ? ((blockIdx.x & 7u) == 0u && (int)(blockIdx.x >> 3) < p.chainw) : (int)blockIdx.x < p.chainw;          // (thread 0's copy is the one that counts)
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = -1;
      if (on_chain) {
        t = (int)atomicAdd(st + LGP_CH_NEXT, 1u);
        if (t >= p.nchain) {
          on_chain = false;                              // the chain is through: help with the rest
          t = -1;
        }
      }
      if (!on_chain) {
        t = p.nchain + (int)atomicAdd(st + LGP_CH_NEXT_BULK, 1u);
        if (t >= p.ntasks) t = -1;
      }
      s_task = t;
    }
    __syncthreads();
    const int t = s_task;
    if (t < 0) break;
    const uint2 tk = p.tasks[t];
    if (p.trace && threadIdx.x == 0) {
      p.trace[4 * t] = blockIdx.x;
      p.trace[4 * t + 1] = wall_clock64();
    }
    const unsigned type = __builtin_amdgcn_readfirstlane(tk.x & 255u);
    const int j = __builtin_amdgcn_readfirstlane((int)((tk.x >> 8) & 255u)), c = __builtin_amdgcn_readfirstlane((int)((tk.x >> 16) & 255u));
    const int k = __builtin_amdgcn_readfirstlane((int)(tk.x >> 24)), r = __builtin_amdgcn_readfirstlane((int)tk.y);
    unsigned *const pst = st + LGP_CH_HEADER + LGP_CH_PANEL * k;
    unsigned *const flagP = pst + LGP_CH_FLAGP, *const flagS = pst + LGP_CH_FLAGS, *const cnt = pst + LGP_CH_CNT;
    // ---- inputs
    if (threadIdx.x == 0) {
      bool ok = true;
      if (type == CH_TYPE_P) {
        if (j == 0) {
          if (k > 0) ok = ch_wait(pst + LGP_CH_U1, 1, st, p.timeout);
        } else {
          ok = ch_wait(cnt + 16 * (2 * j) + 2 * j, j, st, p.timeout) && ch_wait(cnt + 16 * (2 * j + 1) + 2 * j, j, st, p.timeout) && ch_wait(cnt + 16 * (2 * j + 1) + 2 * j + 1, j, st, p.timeout);
        }
      } else if (type == CH_TYPE_S) {
        ok = ch_wait(flagP + j, 1, st, p.timeout);
        if (j > 0) ok = ok && ch_wait(cnt + 16 * r + 2 * j, j, st, p.timeout) && ch_wait(cnt + 16 * r + 2 * j + 1, j, st, p.timeout);
      } else {
        ok = ch_wait(flagS + 8 * r + j, 1, st, p.timeout) && ch_wait(flagS + 8 * c + j, 1, st, p.timeout);
        if (j > 0) ok = ok && ch_wait(cnt + 16 * r + c, j, st, p.timeout);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      s_task = ok ? 0 : -1;
      if (p.trace) p.trace[4 * t + 2] = wall_clock64();
    }
    __syncthreads();
    if (s_task < 0) break;                               // abandoned: every worker leaves
    // ---- the task
    double *const P = p.L + ((long)LGP_NB * k) * (p.ldl + 1);              // top left of panel k's square
    double *const pdblk = p.dblk + (size_t)(8 * k) * LGP_DBLK;
    const int kb = (p.n - LGP_NB * k) < LGP_NB ? (p.n - LGP_NB * k) : LGP_NB;
    if (type == CH_TYPE_P) {
      double *const A = P + (128L * j) * p.ldl + 128 * j, *const dblk = pdblk + (size_t)j * LGP_DBLK;
      const long lda = p.ldl, row0 = p.row0 + (long)LGP_NB * k + 128 * j;
      constexpr int nb = 128;
      int *const info = p.info;
      // the block is published as soon as its fragment image is out: the substitutions read nothing else
#define P4_FRAGMENTS_OUT                                                                                              \
      __syncthreads();                                                                                                \
      if (threadIdx.x == 0) {                                                                                         \
        __hip_atomic_store(flagP + j, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);                                \
        __hip_atomic_fetch_add(pst + LGP_CH_READY + j, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);              \
        if (p.trace) p.trace[4 * t + 3] = wall_clock64();                                                             \
      }
#include "potf2m4_block.inc"
#undef P4_FRAGMENTS_OUT
    } else if (type == CH_TYPE_S) {
      subst_right_wave<double>(pdblk + (size_t)j * LGP_DBLK, 128, P + 128 * j, p.ldl, (long)kb, 64L * r + 16 * (threadIdx.x >> 6));
    } else {
      ch_update_tile(P, p.ldl, j, r, c);
    }
    // ---- publish: the workgroup's stores, then the flag; ready[k][j] counts P(k, j) and the S(k, ., j) for the far-row stream
    __syncthreads();
    if (threadIdx.x == 0 && type != CH_TYPE_P) {         // (P published itself above)
      unsigned *f = type == CH_TYPE_S ? flagS + 8 * r + j : cnt + 16 * r + c;
      __hip_atomic_store(f, type == CH_TYPE_G ? (unsigned)(j + 1) : 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (type == CH_TYPE_S) __hip_atomic_fetch_add(pst + LGP_CH_READY + j, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      if (p.trace) p.trace[4 * t + 3] = wall_clock64();
    }
  }
  if (threadIdx.x == 0 && ch_peek(st + LGP_CH_ABORT) != 0) {
    // nobody may be left waiting for this launch: release every stream-side wait, then report
    // (ready[k][j] and flagP[k][j] are the two families of words the far-row streams poll, potrf.hip potrf_chain; the
    // u2 / u3 / grp words are written by the streams themselves once these waits are through)
    for (int k = 0; k < npanels; ++k)
      for (int j = 0; j < 8; ++j) {
        __hip_atomic_store(st + LGP_CH_HEADER + LGP_CH_PANEL * k + LGP_CH_READY + j, 0x7fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(st + LGP_CH_HEADER + LGP_CH_PANEL * k + LGP_CH_FLAGP + j, 0x7fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    atomicExch(p.info, LGP_INFO_STALLED);
  }
}

// self-test of the two mechanisms the chain kernel lives on: a running kernel sees a value written by a stream
// (hipStreamWriteValue32) and a stream waiting on a value (hipStreamWaitValue32) sees a store of a running kernel.  Bounded:
// after 20 ms the kernel gives up, but still stores the word the stream waits for, so that nothing is left waiting.
__global__ void chain_selftest_kernel(unsigned *words) {       // [0] stream -> kernel, [1] kernel -> stream, [2] verdict
  const unsigned long long t0 = wall_clock64();
  bool seen = false;
  while (!(seen = ch_peek(words) >= 1u) && wall_clock64() - t0 < 2000000ull) __builtin_amdgcn_s_sleep(4);
  words[2] = seen ? 1u : 0u;
  __hip_atomic_store(words + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void chain_selftest_mark(unsigned *words) { words[3] = 1u; }

}  // namespace

// ---- the schedule (host).  Inside a panel: step by step (block j), P, then the substitutions (nearest rows first), then the
// tile updates column by column (the columns of the next diagonal block first) -- every task after the tasks it waits for.
// The tasks ON the dependent chain (P(j), the two row units of block j + 1 solved, its three tiles updated) form a list of
// their own with workers of their own: in one list they queued behind ~100 updates of step j that nothing was waiting for
// (trace of N = 2048: the next potf2 started 18-66 us late).  The rest is taken in step order: everything in it is ready
// soon after P(j), so a worker that takes the next task rarely holds it idle (an order that let far rows lag behind, so as to
// favour near ones, made workers sit on tasks of later steps while earlier, ready ones waited: 30-40 us stalls of the chain).
constexpr unsigned CH_ON_CHAIN = 0x80u;     // bit of the type byte in the published schedule: the task is on the chain list

// out: one topological order of all tasks (what lgp_chaint_schedule publishes); chain / bulk: the two lists the workers take
// from, each a subsequence of `out`
static void chain_build(size_t n, std::vector<uint2> &out, std::vector<uint2> *chain = nullptr, std::vector<uint2> *bulk = nullptr) {
  struct Item { long key; unsigned r, c, w0; bool on_chain; };
  out.clear();
  const int npanels = (int)((n + LGP_NB - 1) / LGP_NB);
  for (int k = 0; k < npanels; ++k) {
    const int kb = (int)((n - (size_t)LGP_NB * k) < (size_t)LGP_NB ? (n - (size_t)LGP_NB * k) : (size_t)LGP_NB), nq = kb / 128, R64 = 2 * nq;
    std::vector<Item> items;
    for (int j = 0; j < nq; ++j) {
      items.push_back({(1024L * j) * 32, 0u, 0u, CH_TYPE_P | (unsigned)j << 8 | (unsigned)k << 24, true});
      for (int r = 2 * j + 2; r < R64; ++r) {
        const int d = r / 2 - j - 1;
        const bool look = d == 0;                        // the rows of the next diagonal block
        const long base = (1024L * j) * 32;
        items.push_back({base + 1, (unsigned)r, 0u, CH_TYPE_S | (unsigned)j << 8 | (unsigned)k << 24, look});
        for (int c = 2 * j + 2; c <= r; ++c)
          items.push_back({base + 2 + (c - 2 * j - 2), (unsigned)r, (unsigned)c, CH_TYPE_G | (unsigned)j << 8 | (unsigned)c << 16 | (unsigned)k << 24,
                           look && c / 2 == j + 1});
      }
    }
    std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) {
      if (a.key != b.key) return a.key < b.key;
      if (a.r != b.r) return a.r < b.r;
      return a.c < b.c;
    });
    for (const Item &it : items) {
      out.push_back(make_uint2(it.w0 | (it.on_chain ? CH_ON_CHAIN : 0u), it.r));
      if (it.on_chain && chain) chain->push_back(make_uint2(it.w0, it.r));
      if (!it.on_chain && bulk) bulk->push_back(make_uint2(it.w0, it.r));
    }
  }
}

long lgp_chaint_schedule(size_t n, unsigned int *tasks, size_t capacity) {
  if (n == 0 || n % 128 != 0 || n > (size_t)LGP_NB * 255) {
    lgp_set_error("invalid argument: the chain schedule needs n a multiple of 128, at most %d", LGP_NB * 255);
    return LGP_EINVAL;
  }
  std::vector<uint2> sched;
  chain_build(n, sched);
  if (tasks != nullptr) {
    const size_t m = sched.size() < capacity ? sched.size() : capacity;
    for (size_t i = 0; i < m; ++i) {
      tasks[2 * i] = sched[i].x;
      tasks[2 * i + 1] = sched[i].y;
    }
  }
  return (long)sched.size();
}

// Kernels that wait for kernels of other streams need an environment that really runs streams side by side and implements
// the stream memory operations.  Under a counter-collecting profiler, for one, kernels are serialised: the chain kernel would
// wait until its time-out.  Checked once per context with a 20 ms bounded handshake; on failure the launch-per-operation driver
// is used instead.
static bool chain_selftest_pair(hipStream_t kernel_stream, hipStream_t other, unsigned *words) {
  bool ok = hipMemset(words, 0, 4 * sizeof(unsigned)) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
  if (!ok) return false;
  hipLaunchKernelGGL(chain_selftest_kernel, dim3(1), dim3(1), 0, kernel_stream, words);
  ok = hipGetLastError() == hipSuccess;
  // (whatever happens from here on, the kernel ends by itself and releases the stream wait)
  ok = ok && hipStreamWriteValue32(other, words, 1, 0) == hipSuccess;
  ok = ok && hipStreamWaitValue32(other, words + 1, 1, hipStreamWaitValueGte, 0xffffffffu) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(chain_selftest_mark, dim3(1), dim3(1), 0, other, words);
    ok = hipGetLastError() == hipSuccess;
  }
  (void)hipStreamSynchronize(kernel_stream);
  (void)hipStreamSynchronize(other);
  unsigned h[4] = {0, 0, 0, 0};
  return ok && hipMemcpy(h, words, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess && h[2] == 1u && h[3] == 1u;
}

// The chain kernel runs on stream2 and trades flag words with EVERY other stream of the driver (potrf_chain: the caller's
// stream for u2 / u3, the far-row streams for ready / flagP / u1): each of them is paired with stream2 in turn.  A stream that
// the runtime mapped onto stream2's hardware queue (more streams in the process than queues) fails its pairing -- its packets
// sit behind the resident kernel -- and the context falls back to the launch-per-operation driver instead of stalling.
static bool chain_selftest(lgp_ctx *ctx) {
  if (ctx->far_streams.empty()) ctx->far_streams.push_back(ctx->stream3);        // (the context's third stream is the first of them)
  while (ctx->far_streams.size() < (size_t)ctx->chain_groups) {
    hipStream_t s;
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, ctx->prio_hi) != hipSuccess) return false;
    ctx->far_streams.push_back(s);
  }
  unsigned *words = nullptr;
  if (hipMalloc(reinterpret_cast<void **>(&words), 4 * sizeof(unsigned)) != hipSuccess) return false;
  bool ok = chain_selftest_pair(ctx->stream2, ctx->stream, words);
  for (size_t g = 0; ok && g < ctx->far_streams.size(); ++g) ok = chain_selftest_pair(ctx->stream2, ctx->far_streams[g], words);
  (void)hipFree(words);
  (void)hipGetLastError();
  return ok;
}

bool lgp_chaint_usable(lgp_ctx *ctx, size_t n, size_t ldl) {
  if (ctx->chain_mode == 0 && n >= 256 && n % 128 == 0 && ctx->stream2 != nullptr && ctx->stream3 != nullptr) {
    ctx->chain_mode = chain_selftest(ctx) ? 2 : 1;      // 2: checked and available, 1: off
  }
  return ctx->chain_mode == 2 && n >= 256 && n % 128 == 0 && n <= (size_t)LGP_NB * 255 && ldl < (1u << 20);
}

size_t lgp_chaint_state_bytes(size_t n) {
  return ((size_t)LGP_CH_HEADER + (size_t)LGP_CH_PANEL * ((n + LGP_NB - 1) / LGP_NB)) * sizeof(unsigned);
}

int lgp_chaint_state(lgp_ctx *ctx, size_t n, unsigned **out) {
  const size_t need = lgp_chaint_state_bytes(n);
  if (ctx->chain_state_bytes < need) {
    // (the previous buffer, if any, is no longer in use: every factorisation joins its streams into ctx->stream, and the
    // caller's stream is synchronised by the blocking allocation)
    if (ctx->chain_state) LGP_HIP(hipFree(ctx->chain_state));
    ctx->chain_state = nullptr;
    ctx->chain_state_bytes = 0;
    LGP_HIP(hipMalloc(&ctx->chain_state, need));
    ctx->chain_state_bytes = need;
  }
  *out = static_cast<unsigned *>(ctx->chain_state);
  return LGP_OK;
}

int lgp_chaint_launch(lgp_ctx *ctx, hipStream_t stream, size_t n, double *L, size_t ldl, double *dblk, size_t row0) {
  LGP_REQUIRE(lgp_chaint_usable(ctx, n, ldl), "matrix not covered by the chain kernel");
  LGP_REQUIRE(ctx->chain_state != nullptr && ctx->chain_state_bytes >= lgp_chaint_state_bytes(n), "chain state not allocated");
  const uint2 *dtasks = nullptr;
  int ntasks = 0, nchain = 0;
  for (const auto &e : ctx->chain_sched)
    if (e.key == (unsigned long long)n) {
      dtasks = static_cast<const uint2 *>(e.tasks);
      ntasks = e.ntasks;
      nchain = e.nchain;
    }
  if (dtasks == nullptr) {
    std::vector<uint2> all, sched, bulk;
    chain_build(n, all, &sched, &bulk);
    nchain = (int)sched.size();
    sched.insert(sched.end(), bulk.begin(), bulk.end());
    void *d = nullptr;
    LGP_HIP(hipMalloc(&d, sched.size() * sizeof(uint2)));
    // (blocking copy from pageable memory; once per matrix size and context)
    LGP_HIP(hipMemcpy(d, sched.data(), sched.size() * sizeof(uint2), hipMemcpyHostToDevice));
    ctx->chain_sched.push_back({(unsigned long long)n, d, (int)sched.size(), nchain});
    dtasks = static_cast<const uint2 *>(d);
    ntasks = (int)sched.size();
  }
  ChainArgs a;
  a.L = L; a.ldl = (long)ldl; a.n = (int)n; a.dblk = dblk; a.info = ctx->d_info;
  a.tasks = dtasks; a.ntasks = ntasks; a.nchain = nchain; a.state = static_cast<unsigned *>(ctx->chain_state);
  a.trace = nullptr;
  a.row0 = (long)row0;
  if (getenv("LGP_CHAIN_TRACE") != nullptr) {           // debugging aid: the last launch's task times, written out by lgp_ctx_destroy
    if (ctx->chain_trace != nullptr && ctx->chain_trace_tasks < (size_t)ntasks) {
      LGP_HIP(hipFree(ctx->chain_trace));
      ctx->chain_trace = nullptr;
    }
    if (ctx->chain_trace == nullptr) LGP_HIP(hipMalloc(&ctx->chain_trace, (size_t)ntasks * 4 * sizeof(unsigned long long)));
    ctx->chain_trace_tasks = (size_t)ntasks;
    ctx->chain_trace_sched = dtasks;
    LGP_HIP(hipMemsetAsync(ctx->chain_trace, 0, (size_t)ntasks * 4 * sizeof(unsigned long long), stream));
    a.trace = static_cast<unsigned long long *>(ctx->chain_trace);
  }
  // at least one worker beside the chain's own: tasks of the chain list wait for tasks of the bulk list (S(4, 1) for the tiles
  // of step 0), which only a worker that is not holding a chain task can run
  const int want = ctx->chain_workers < 2 ? 2 : ctx->chain_workers;
  const int workers = ntasks < want ? (ntasks < 2 ? 2 : ntasks) : want;
  a.timeout = ctx->chain_timeout;
  a.chainw = workers >= 4 ? 3 : 1;
  static const bool samexcd = getenv("LGP_CHAIN_SAMEXCD") != nullptr;
  a.samexcd = (samexcd && workers >= 17 && a.chainw == 3) ? 1 : 0;      // the chain offers three tasks at a time at most: S(2j+2), S(2j+3) / the three tiles of the next block
  // dynamic LDS that is never touched: with it no trailing-update workgroup (64 KB of LDS) fits on a worker's CU
  const size_t hog = n <= ctx->chain_hog_max_n ? (size_t)ctx->chain_lds_kb * 1024 : 0;
  if (!ctx->chain_attr_set) {      // per context (= per device; a context is used by one host thread at a time)
    LGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    ctx->chain_attr_set = true;
  }
  hipLaunchKernelGGL(chain_kernel, dim3((unsigned)workers), dim3(256), hog, stream, a);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

// ---- the chain kernel of the library (round 4 had an alternative whose first workgroup kept the diagonal block on chip: slower,
// kept under docs/history/walker)
extern "C" long lgp_chain_schedule(size_t n, unsigned int *tasks, size_t capacity) { return lgp_chaint_schedule(n, tasks, capacity); }
bool lgp_chain_usable(lgp_ctx *ctx, size_t n, size_t ldl) { return lgp_chaint_usable(ctx, n, ldl); }
size_t lgp_chain_state_bytes(size_t n) { return lgp_chaint_state_bytes(n); }
int lgp_chain_state(lgp_ctx *ctx, size_t n, unsigned **out) { return lgp_chaint_state(ctx, n, out); }
int lgp_chain_launch(lgp_ctx *ctx, hipStream_t stream, size_t n, double *L, size_t ldl, double *dblk, size_t row0) {
  return lgp_chaint_launch(ctx, stream, n, L, ldl, dblk, row0);
}

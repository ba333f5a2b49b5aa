// Probe: can a stream wait for a flag that a RUNNING kernel sets (hipStreamWaitValue32), and can a stream release a
// spinning kernel (hipStreamWriteValue32)?  Which memory works (hipMalloc / signal memory), and what does a hop cost?
//   hipcc --offload-arch=gfx950 -O2 tools/waitvalue_probe.hip -o tools/waitvalue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d: %s\n", hipGetErrorString(e_), __LINE__, #x); return 1; } } while (0)

// waits (bounded) for in[i] >= 1, then sets out[i] = 1, for i = 0 .. n-1; stamps[i] = wall clock when out[i] was set
__global__ void pingpong(unsigned *in, unsigned *out, unsigned long long *stamps, int n, int sys_scope) {
  for (int i = 0; i < n; ++i) {
    const unsigned long long t0 = wall_clock64();
    if (in != nullptr)
      while (__hip_atomic_load(in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < 1u) {
        __builtin_amdgcn_s_sleep(4);
        if (wall_clock64() - t0 > 200000000ull) { stamps[i] = ~0ull; return; }   // 2 s
      }
    else {
      while (wall_clock64() - t0 < 2000ull) {}        // 20 us of "work" per step
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    stamps[i] = wall_clock64();
    if (sys_scope) __hip_atomic_store(out + i, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    else __hip_atomic_store(out + i, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ void stamp(unsigned long long *s, int i) { s[i] = wall_clock64(); }

int main() {
  const int n = 8;
  for (int mem = 0; mem < 2; ++mem)
    for (int sys = 0; sys < 2; ++sys) {
      unsigned *a, *b;
      if (mem == 0) { CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); }
      else { CK(hipExtMallocWithFlags((void **)&a, n * 4, hipMallocSignalMemory)); CK(hipExtMallocWithFlags((void **)&b, n * 4, hipMallocSignalMemory)); }
      unsigned long long *s1, *s2;
      CK(hipMalloc(&s1, n * 8)); CK(hipMalloc(&s2, n * 8));
      CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(s1, 0, n * 8)); CK(hipMemset(s2, 0, n * 8));
      hipStream_t A, B;
      CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
      // test 1: a running kernel on A sets b[i] every 20 us; stream B waits for each b[i] and runs a stamp kernel
      pingpong<<<1, 1, 0, A>>>(nullptr, b, s1, n, sys);
      for (int i = 0; i < n; ++i) {
        hipError_t e = hipStreamWaitValue32(B, b + i, 1, hipStreamWaitValueGte, 0xffffffffu);
        if (e != hipSuccess) { printf("mem %d sys %d: hipStreamWaitValue32 -> %s\n", mem, sys, hipGetErrorString(e)); break; }
        stamp<<<1, 1, 0, B>>>(s2, i);
      }
      auto t0 = std::chrono::steady_clock::now();
      bool done = false;
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 5.0) {
        if (hipStreamQuery(A) == hipSuccess && hipStreamQuery(B) == hipSuccess) { done = true; break; }
      }
      if (!done) { printf("mem %d sys %d: test 1 did not finish in 5 s (stream wait never released)\n", mem, sys); return 2; }
      unsigned long long h1[n], h2[n];
      CK(hipMemcpy(h1, s1, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2, s2, n * 8, hipMemcpyDeviceToHost));
      printf("mem %s, %s-scope stores: kernel flag -> waiting stream's next kernel runs after (us):", mem ? "signal" : "hipMalloc", sys ? "system" : "agent");
      for (int i = 0; i < n; ++i) printf(" %.1f", (double)(long long)(h2[i] - h1[i]) / 100.0);
      printf("\n");
      // test 2: a kernel on A spins on a[i]; stream B writes a[i] after a stamp kernel
      CK(hipMemset(s1, 0, n * 8)); CK(hipMemset(s2, 0, n * 8)); CK(hipMemset(b, 0, n * 4));
      pingpong<<<1, 1, 0, A>>>(a, b, s1, n, sys);
      bool ok = true;
      for (int i = 0; i < n; ++i) {
        stamp<<<1, 1, 0, B>>>(s2, i);
        hipError_t e = hipStreamWriteValue32(B, a + i, 1, 0);
        if (e != hipSuccess) { printf("hipStreamWriteValue32 -> %s\n", hipGetErrorString(e)); ok = false; break; }
      }
      if (ok) {
        t0 = std::chrono::steady_clock::now(); done = false;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 5.0) {
          if (hipStreamQuery(A) == hipSuccess && hipStreamQuery(B) == hipSuccess) { done = true; break; }
        }
        if (!done) { printf("test 2 did not finish in 5 s\n"); return 3; }
        CK(hipMemcpy(h1, s1, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2, s2, n * 8, hipMemcpyDeviceToHost));
        printf("   stream write -> spinning kernel sees it after (us since the stamp kernel before the write):");
        for (int i = 0; i < n; ++i) printf(" %.1f", h1[i] == ~0ull ? -1.0 : (double)(long long)(h1[i] - h2[i]) / 100.0);
        printf("\n");
      }
      CK(hipStreamDestroy(A)); CK(hipStreamDestroy(B));
      CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(s1)); CK(hipFree(s2));
    }
  return 0;
}

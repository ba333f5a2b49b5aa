// clock_probe: at what shader clock does a LONE latency-bound wave run, and does load on the other CUs change it?
// One wave runs a chain of dependent fp64 FMAs and reads both the shader-clock counter (s_memtime) and the constant 100 MHz
// counter (s_memrealtime); alone, then beside a kernel that keeps every other CU busy with MFMAs.
//   hipcc --offload-arch=gfx950 -O2 -o tools/clock_probe tools/clock_probe.hip && tools/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void chain(double *out, long long *stamps, int n) {
  double a = out[0], b = 1.0000001, c = 1e-9;
  unsigned long long r0, r1;
  asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
  const long long c0 = clock64();
  for (int i = 0; i < n; ++i) a = __builtin_fma(a, b, c);
  const long long c1 = clock64();
  asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
  out[0] = a;
  stamps[0] = c1 - c0;
  stamps[1] = (long long)(r1 - r0);
}

typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) busy(double *sink, int iters) {
  v4d acc = {0, 0, 0, 0};
  const double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
  for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  if (acc[0] == 12345.678) sink[0] = acc[1];
}

int main() {
  double *d; long long *st; double *sink;
  CK(hipMalloc(&d, 8)); CK(hipMalloc(&st, 16)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(d, 0, 8));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const int n = 2000000;
  for (int mode = 0; mode < 4; ++mode) {
    // mode 0, 1: alone (twice);  2: beside 248 busy workgroups;  3: beside 1024
    if (mode >= 2) hipLaunchKernelGGL(busy, dim3(mode == 2 ? 248 : 1024), dim3(256), 0, s2, sink, 3000000);
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, d, st, n);
    CK(hipStreamSynchronize(s1));
    long long h[2];
    CK(hipMemcpy(h, st, 16, hipMemcpyDeviceToHost));
    CK(hipDeviceSynchronize());
    const double us = h[1] / 100.0;
    printf("mode %d: %d dependent FMAs: %lld shader cycles, %.1f us  ->  %.2f cycles per FMA, %.0f MHz, %.2f ns per FMA\n", mode, n, h[0], us,
           (double)h[0] / n, h[0] / us, us * 1e3 / n);
  }
  return 0;
}

// Probe: cost per iteration of the pivot loop of the matrix-core potf2 (one diagonal tile), and of the same loop with
// parts removed, to find what a lone in-order wave really pays.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} }while(0)

__device__ __forceinline__ double bperm(int addr, double v) {
  const long long bits = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_ds_bpermute(addr, (int)bits), hi = __builtin_amdgcn_ds_bpermute(addr, (int)(bits >> 32));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double rdlane(double v, int lane) {
  const long long bits = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)bits, lane), hi = __builtin_amdgcn_readlane((int)(bits >> 32), lane);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double rcp_chain(double d) {
  const double y = __builtin_amdgcn_rcp(d);
  const double e = __builtin_fma(-d, y, 1.0);
  const double e2 = __builtin_fma(e, e, e);
  return __builtin_fma(y, e2, y);
}

template <int MODE, int NT>
__global__ void probe(unsigned long long *out, double *sink, int iters) {
  const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
  double w[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) w[i] = 1.0 + 0.001 * lane + i;
  double dg = 1.0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const int g0 = it & 3, kk = g0;
    double d;
    if (MODE & 1) d = rdlane(w[0], g0 * 16 + kk); else d = 1.0 + 1e-9 * it;
    const int arow = (g0 * 16 + n) * 4, acol = (g0 * 16 + g) * 4;
    double cf, rn[NT];
    if (MODE & 2) {
      cf = bperm(acol, w[0]);
#pragma unroll
      for (int i = 0; i < NT; ++i) rn[i] = bperm(arow, w[i]);
    } else {
      cf = w[0] * 0.5;
#pragma unroll
      for (int i = 0; i < NT; ++i) rn[i] = w[i] * 0.25;
    }
    double rc;
    if (MODE & 4) rc = rcp_chain(d); else rc = d * 1e-3;
    const double cfm = (g > g0) ? -cf : 0.0;
    w[0] = __builtin_fma(cfm * rn[0], rc, w[0]) + 1e-30;
    const double cfs = cfm * rc;
#pragma unroll
    for (int i = 1; i < NT; ++i) w[i] = __builtin_fma(cfs, rn[i], w[i]);
    dg = (g == g0) ? d : dg;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[0] = t1 - t0;
  double s = dg;
#pragma unroll
  for (int i = 0; i < NT; ++i) s += w[i];
  sink[lane] = s;
}

template <int MODE, int NT>
int run(const char *name, unsigned long long *d, double *s) {
  const int iters = 4000;
  for (int rep = 0; rep < 2; ++rep) { probe<MODE, NT><<<1, 64>>>(d, s, iters); CK(hipDeviceSynchronize()); }
  unsigned long long h;
  CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
  printf("%-60s %8.1f cycles / pivot\n", name, (double)h / iters);
  return 0;
}

int main() {
  unsigned long long *d; double *s;
  CK(hipMalloc(&d, 256)); CK(hipMalloc(&s, 64 * 8));
  run<7, 1>("1 tile: readlane + bpermute + rcp chain (as the kernel)", d, s);
  run<6, 1>("1 tile: no readlane", d, s);
  run<5, 1>("1 tile: no bpermute", d, s);
  run<3, 1>("1 tile: no rcp chain", d, s);
  run<0, 1>("1 tile: arithmetic only", d, s);
  run<7, 8>("8 tiles: as the kernel", d, s);
  run<5, 8>("8 tiles: no bpermute", d, s);
  run<7, 4>("4 tiles: as the kernel", d, s);
  return 0;
}

// Probe: semantics and dependent-chain latency of the lane-crossing VALU operations of gfx950 that can replace ds_bpermute in
// the pivot loops: v_permlane16_swap / v_permlane32_swap (row broadcast of a 16-lane row to all four rows in two steps) and
// v_mov_dpp row_newbcast with a row mask (lane N of the own row, a different N per row).
//   hipcc --offload-arch=gfx950 -O3 tools/permlane_probe.hip -o tools/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int G0> __device__ __forceinline__ unsigned rowbcast(unsigned x) {
  const u2 a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  const unsigned h = (G0 & 1) ? a.y : a.x;
  const u2 b = __builtin_amdgcn_permlane32_swap(h, h, false, false);
  return (G0 & 2) ? b.y : b.x;
}
__global__ void sem(unsigned *out) {
  const unsigned x = 100 * (threadIdx.x >> 4) + (threadIdx.x & 15);       // 100 row + lane-in-row
  out[0 * 64 + threadIdx.x] = rowbcast<0>(x);
  out[1 * 64 + threadIdx.x] = rowbcast<1>(x);
  out[2 * 64 + threadIdx.x] = rowbcast<2>(x);
  out[3 * 64 + threadIdx.x] = rowbcast<3>(x);
  unsigned y = 7777;
  y = __builtin_amdgcn_update_dpp(y, x, 0x150 + 4, 0x1, 0xf, false);      // row 0 <- its lane 4
  y = __builtin_amdgcn_update_dpp(y, x, 0x150 + 5, 0x2, 0xf, false);      // row 1 <- its lane 5
  y = __builtin_amdgcn_update_dpp(y, x, 0x150 + 6, 0x4, 0xf, false);
  out[4 * 64 + threadIdx.x] = y;                                          // row 3 keeps 7777
}
__global__ void lat(unsigned long long *out, int iters) {
  unsigned x = threadIdx.x;
  double d = 1.0 + threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) x = rowbcast<1>(x) + 1;                  // two swaps + add, dependent
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) x = __builtin_amdgcn_update_dpp(x, x, 0x150 + 3, 0xf, 0xf, false) + 1;
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) x = __builtin_amdgcn_ds_bpermute(((x & 63) * 4), x) + 1;
  unsigned long long t3 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {                                        // the pivot chain: row broadcast -> dpp -> mul -> fma
    const long long bits = __builtin_bit_cast(long long, d);
    const unsigned lo = rowbcast<2>((unsigned)bits), hi = rowbcast<2>((unsigned)(bits >> 32));
    const unsigned clo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + 7, 0xf, 0xf, false), chi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + 7, 0xf, 0xf, false);
    const double y = __builtin_bit_cast(double, ((long long)hi << 32) | lo), c = __builtin_bit_cast(double, ((long long)chi << 32) | clo);
    d = __builtin_fma(c * y, 1e-30, d);
  }
  unsigned long long t4 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; out[4] = x + (unsigned long long)d; }
}
int main() {
  unsigned *d; CK(hipMalloc(&d, 5 * 64 * 4));
  sem<<<1, 64>>>(d); CK(hipDeviceSynchronize());
  unsigned h[5 * 64]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  for (int t = 0; t < 5; ++t) {
    printf(t < 4 ? "rowbcast<%d>: " : "dpp newbcast rows 0..2 (lane 4, 5, 6 of the own row), row 3 untouched: ", t);
    for (int r = 0; r < 4; ++r) printf("[%u %u .. %u] ", h[t * 64 + 16 * r], h[t * 64 + 16 * r + 1], h[t * 64 + 16 * r + 15]);
    printf("\n");
  }
  unsigned long long *dt; CK(hipMalloc(&dt, 64));
  const int iters = 20000;
  lat<<<1, 64>>>(dt, iters); CK(hipDeviceSynchronize());
  unsigned long long ht[5]; CK(hipMemcpy(ht, dt, 40, hipMemcpyDeviceToHost));
  printf("dependent cycles per iteration (s_memtime ticks at 100 MHz x 24 if the counter is the 100 MHz one: raw %llu %llu %llu %llu over %d iterations)\n", ht[0], ht[1], ht[2], ht[3], iters);
  printf("  rowbcast + add %.1f | dpp newbcast + add %.1f | ds_bpermute + add %.1f | f64: 2 x rowbcast + 2 x dpp + mul + fma %.1f  (raw ticks / iteration)\n",
         (double)ht[0] / iters, (double)ht[1] / iters, (double)ht[2] / iters, (double)ht[3] / iters);
  return 0;
}

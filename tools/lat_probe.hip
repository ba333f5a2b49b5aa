// Probe: dependent-chain latencies (cycles per operation, one wave alone on a CU) of the instructions the matrix-core
// potf2 chains: fp64 fma, v_rcp_f64, v_rsq_f64, ds_bpermute_b32 (round trip), v_readlane -> VALU, fp64 MFMA dependent.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} }while(0)
typedef double acc_t __attribute__((ext_vector_type(4)));

__global__ void probe(unsigned long long *out, double *sink, int iters, double seed) {
  const int lane = threadIdx.x;
  double a = seed + lane * 1e-9, b = 1.0000001, c = 1e-9;
  unsigned long long t0, t1;
  int k = 0;
  // 1. fp64 fma dependent
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) { a = __builtin_fma(a, b, c); a = __builtin_fma(a, b, c); a = __builtin_fma(a, b, c); a = __builtin_fma(a, b, c); }
  asm volatile("" :: "v"(a));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[k] = t1 - t0; ++k;
  // 2. rcp dependent
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) { a = __builtin_amdgcn_rcp(a); a = __builtin_amdgcn_rcp(a); a = __builtin_amdgcn_rcp(a); a = __builtin_amdgcn_rcp(a); }
  asm volatile("" :: "v"(a));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[k] = t1 - t0; ++k;
  // 3. rsq dependent
  a = fabs(a) + 1.0;
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) { a = __builtin_amdgcn_rsq(a); a = __builtin_amdgcn_rsq(a); a = __builtin_amdgcn_rsq(a); a = __builtin_amdgcn_rsq(a); }
  asm volatile("" :: "v"(a));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[k] = t1 - t0; ++k;
  // 4. bpermute round trip (one dword), dependent through the data
  int v = lane, addr = ((lane + 17) & 63) * 4;
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) { v = __builtin_amdgcn_ds_bpermute(addr, v); v = __builtin_amdgcn_ds_bpermute(addr, v); v = __builtin_amdgcn_ds_bpermute(addr, v); v = __builtin_amdgcn_ds_bpermute(addr, v); }
  asm volatile("" :: "v"(v));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[k] = t1 - t0; ++k;
  // 5. readlane -> VALU -> readlane (fp64 via two readlanes + one fma)
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      long long bits = __builtin_bit_cast(long long, a);
      int lo = __builtin_amdgcn_readlane((int)bits, 5), hi = __builtin_amdgcn_readlane((int)(bits >> 32), 5);
      double s = __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
      a = __builtin_fma(s, b, a);
    }
  }
  asm volatile("" :: "v"(a));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[k] = t1 - t0; ++k;
  // 6. fp64 MFMA dependent accumulator chain
  acc_t acc = {a, a, a, a};
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0); }
  asm volatile("" :: "v"(acc));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[k] = t1 - t0; ++k;
  // 7. MFMA result -> VALU fma -> MFMA operand (A from the accumulator): the slab hand-over
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[1] * b, c, acc, 0, 0, 0); }
  }
  asm volatile("" :: "v"(acc));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[k] = t1 - t0; ++k;
  // 8. 16 independent bpermutes then one wait (throughput)
  int w[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) w[u] = lane + u;
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) w[u] = __builtin_amdgcn_ds_bpermute(addr, w[u]);
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) asm volatile("" :: "v"(w[u]));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[k] = t1 - t0; ++k;
  // 9. 16 independent fp64 fma (issue rate)
  double f[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) f[u] = a + u;
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) f[u] = __builtin_fma(f[u], b, c);
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) asm volatile("" :: "v"(f[u]));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[k] = t1 - t0; ++k;
  sink[lane] = a + v + acc[0] + w[3] + f[5];
}

int main() {
  unsigned long long *d; double *s;
  CK(hipMalloc(&d, 256)); CK(hipMalloc(&s, 64 * 8));
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) { probe<<<1, 64>>>(d, s, iters, 1.5); CK(hipDeviceSynchronize()); }
  unsigned long long h[16];
  CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  const char *names[] = {"fp64 fma dependent", "v_rcp_f64 dependent", "v_rsq_f64 dependent", "ds_bpermute_b32 round trip", "readlane x2 + fma dependent",
                         "fp64 mfma dependent (acc chain)", "mfma -> v_mul -> mfma A operand", "ds_bpermute_b32 x16 independent (per op)", "fp64 fma x16 independent (per op)"};
  for (int k = 0; k < 9; ++k) {
    double per = (double)h[k] / (iters * (k >= 7 ? 16.0 : 4.0));
    printf("%-45s %8.1f cycles\n", names[k], per);
  }
  return 0;
}

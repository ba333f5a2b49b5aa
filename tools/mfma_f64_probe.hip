// Probe: v_mfma_f64_16x16x4_f64 operand/result lane maps and issue rate on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__global__ void layout_probe(const double* A, const double* B, double* C) {
  // A: 16x4 row-major, B: 4x16 row-major, C: 16x16 row-major; one wave.
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[l * 4 + r] = c[r];  // raw dump: lane-major
}

template <int NACC>
__global__ void __launch_bounds__(256) rate_probe(double* out, int iters, double seed) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) fma_probe(double* out, int iters, double seed) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = i;
  double a = seed + threadIdx.x * 1e-9, b = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(a, acc[i], b);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s arch %s CUs %d clock %d kHz mem %.1f GiB\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate, p.totalGlobalMem / 1073741824.0);
  // ---- layout
  std::vector<double> A(64), B(64), C(256), R(256, 0.0);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 0.5 + k * 7;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 3 + k * 11 + j * j * 0.25;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; R[i * 16 + j] = s; }
  double *dA, *dB, *dC; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dC, 2048));
  CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
  layout_probe<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize());
  CK(hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost));
  // hypothesis 1: col = l&15, row = (l>>4) + 4*r ; hypothesis 2: row = 4*(l>>4) + r
  int bad1 = 0, bad2 = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    double v = C[l * 4 + r];
    if (v != R[((l >> 4) + 4 * r) * 16 + (l & 15)]) bad1++;
    if (v != R[(4 * (l >> 4) + r) * 16 + (l & 15)]) bad2++;
  }
  printf("layout: row=(l>>4)+4r mismatches %d ; row=4(l>>4)+r mismatches %d\n", bad1, bad2);
  // ---- rate
  int nblk = p.multiProcessorCount * 4; int iters = 4096;
  double* dout; CK(hipMalloc(&dout, nblk * 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](auto kern, int nacc, const char* name, int blocks) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dout, iters, 1.0 + rep);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double flops = (double)blocks * 4 * iters * nacc * 2048.0;
      printf("%s blocks=%d nacc=%d: %.3f ms  %.2f TFLOP/s\n", name, blocks, nacc, ms, flops / ms * 1e-9);
    }
  };
  run(rate_probe<1>, 1, "mfma_f64 dep-chain", p.multiProcessorCount);
  run(rate_probe<4>, 4, "mfma_f64", p.multiProcessorCount);
  run(rate_probe<16>, 16, "mfma_f64", p.multiProcessorCount);
  run(rate_probe<16>, 16, "mfma_f64 2wave/simd", p.multiProcessorCount * 2);
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(fma_probe, dim3(nblk), dim3(256), 0, 0, dout, iters, 1.0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double flops = (double)nblk * 256 * iters * 16 * 2.0;
    printf("v_fma_f64 blocks=%d: %.3f ms %.2f TFLOP/s\n", nblk, ms, flops / ms * 1e-9);
  }
  return 0;
}

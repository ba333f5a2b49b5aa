// Probe: time potf2_kernel / subst kernels alone (1 workgroup) and back-to-back, and read the shader clock.
#include "../lsqfitgp_amd/csrc/potrf.hip"
#include "../lsqfitgp_amd/csrc/potf2m.hip"
#include <vector>
#include <cmath>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
void lgp_set_error(const char *fmt, ...) {}
int lgp_prof_start(lgp_ctx *, hipStream_t, double) { return 0; }
size_t lgp_gemm_t64_trailing_max() { return 10000; }
int lgp_prof_stop(lgp_ctx *, hipStream_t) { return 0; }
template <typename T> int lgp_gemm_launch(lgp_ctx *, hipStream_t, bool, bool, size_t, size_t, size_t, T, const T *, size_t, const T *, size_t, T, T *, size_t, bool, int, int, int) { return 0; }
template int lgp_gemm_launch<double>(lgp_ctx *, hipStream_t, bool, bool, size_t, size_t, size_t, double, const double *, size_t, const double *, size_t, double, double *, size_t, bool, int, int, int);
template int lgp_gemm_launch<float>(lgp_ctx *, hipStream_t, bool, bool, size_t, size_t, size_t, float, const float *, size_t, const float *, size_t, float, float *, size_t, bool, int, int, int);

__global__ void clock_probe(unsigned long long *out, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  double a = threadIdx.x, b = 1e-9;
  for (int i = 0; i < iters; ++i) a = __builtin_fma(a, 1.0000001, b);
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)a; }
}

int main() {
  const int n = 128;
  std::vector<double> A(n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = std::exp(-0.5 * (i - j) * (i - j) / 400.0) + (i == j ? 1e-3 : 0);
  double *dA, *dW; int *dinfo; unsigned long long *dclk;
  CK(hipMalloc(&dA, n * n * 8)); CK(hipMalloc(&dW, LGP_DBLK * 8)); CK(hipMalloc(&dinfo, 4)); CK(hipMemset(dinfo, 0, 4)); CK(hipMalloc(&dclk, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // clock when a single wave is running
  for (int rep = 0; rep < 3; ++rep) {
    clock_probe<<<1, 64>>>(dclk, 200000); CK(hipDeviceSynchronize());
    unsigned long long h[3]; CK(hipMemcpy(h, dclk, 24, hipMemcpyDeviceToHost));
    printf("single-wave clock: %.0f MHz (%llu cycles)\n", (double)h[0] / h[1] * 100.0, h[0]);
  }
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0));
    potf2_kernel<double><<<1, 256>>>(dA, n, n, dW, dinfo, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("potf2 alone: %.1f us\n", ms * 1e3);
  }
  // the matrix-core kernel: same block, compared with the VALU kernel's result, timed alone and back-to-back
  {
    std::vector<double> L0(n * n), L1(n * n);
    CK(hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice));
    potf2_kernel<double><<<1, 256>>>(dA, n, n, dW, dinfo, 0);
    CK(hipMemcpy(L0.data(), dA, n * n * 8, hipMemcpyDeviceToHost));
    std::vector<double> F0(LGP_NFRAG * 64), F1(LGP_NFRAG * 64);
    pack_frag_kernel<double><<<(LGP_NFRAG + 3) / 4, 256>>>(dW);
    CK(hipMemcpy(F0.data(), dW + LGP_FRAG0, F0.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemset(dW, 0, LGP_DBLK * 8));
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice));
      CK(hipEventRecord(e0));
      potf2_mfma_kernel<<<1, 64>>>(dA, n, n, dW, dinfo, 0);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("potf2_mfma alone: %.1f us\n", ms * 1e3);
    }
    CK(hipMemcpy(L1.data(), dA, n * n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(F1.data(), dW + LGP_FRAG0, F1.size() * 8, hipMemcpyDeviceToHost));
    double dl = 0, df = 0, nl = 0, nf = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { dl = fmax(dl, fabs(L0[i * n + j] - L1[i * n + j])); nl = fmax(nl, fabs(L0[i * n + j])); }
    for (size_t i = 0; i < F0.size(); ++i) { df = fmax(df, fabs(F0[i] - F1[i])); nf = fmax(nf, fabs(F0[i])); }
    printf("potf2_mfma vs potf2: max |dL| = %.3e (max |L| %.3e), max |dF| = %.3e (max |F| %.3e)\n", dl, nl, df, nf);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 50; ++i) potf2_mfma_kernel<<<1, 64>>>(dA, n, n, dW, dinfo, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("potf2_mfma x50 back-to-back: %.1f us each\n", ms * 1e3 / 50);
#ifdef LGP_POTF2_TRACE
    unsigned long long st[16];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(lgp_potf2_trace), sizeof(st)));
    printf("potf2_mfma phases (cycles): ");
    for (int k = 1; k <= 9; ++k) printf("%llu ", st[k] - st[k - 1]);
    printf("| hot-loop passes: %llu %llu %llu", st[11] - st[9], st[12] - st[11], st[13] - st[12]);
    printf("\n");
#endif
  }
  // the four-wave matrix-core kernel
  {
    std::vector<double> L0(n * n), L1(n * n), F0(LGP_NFRAG * 64), F1(LGP_NFRAG * 64);
    CK(hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice));
    potf2_kernel<double><<<1, 256>>>(dA, n, n, dW, dinfo, 0);
    pack_frag_kernel<double><<<(LGP_NFRAG + 3) / 4, 256>>>(dW);
    CK(hipMemcpy(L0.data(), dA, n * n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(F0.data(), dW + LGP_FRAG0, F0.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemset(dW, 0, LGP_DBLK * 8));
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice));
      CK(hipEventRecord(e0));
      potf2_mfma4_kernel<<<1, 256>>>(dA, n, n, dW, dinfo, 0);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("potf2_mfma4 alone: %.1f us\n", ms * 1e3);
    }
    CK(hipMemcpy(L1.data(), dA, n * n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(F1.data(), dW + LGP_FRAG0, F1.size() * 8, hipMemcpyDeviceToHost));
    double dl = 0, df = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) dl = fmax(dl, fabs(L0[i * n + j] - L1[i * n + j]));
    for (size_t i = 0; i < F0.size(); ++i) df = fmax(df, fabs(F0[i] - F1[i]));
    printf("potf2_mfma4 vs potf2: max |dL| = %.3e, max |dF| = %.3e\n", dl, df);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 50; ++i) potf2_mfma4_kernel<<<1, 256>>>(dA, n, n, dW, dinfo, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("potf2_mfma4 x50 back-to-back: %.1f us each\n", ms * 1e3 / 50);
#ifdef LGP_POTF2_TRACE
    {
      CK(hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice));
      potf2_mfma4_kernel<<<1, 256>>>(dA, n, n, dW, dinfo, 0);
      CK(hipDeviceSynchronize());
      unsigned long long st[64];
      CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(lgp_p4_trace), sizeof(st)));
      printf("potf2_mfma4 wave 0 (cycles): loads %llu |", st[1] - st[0]);
      for (int s2 = 0; s2 < 8; ++s2)
        printf(" step %d: publish %llu pivots %llu barrier %llu trailing %llu |", s2, st[2 + 4 * s2] - (s2 ? st[1 + 4 * s2] : st[1]), st[3 + 4 * s2] - st[2 + 4 * s2],
               st[4 + 4 * s2] - st[3 + 4 * s2], st[5 + 4 * s2] - st[4 + 4 * s2]);
      printf(" stores %llu; total %llu\n", st[34] - st[33], st[34] - st[0]);
    }
#endif
  }
  // 50 back-to-back
  CK(hipEventRecord(e0));
  for (int i = 0; i < 50; ++i) potf2_kernel<double><<<1, 256>>>(dA, n, n, dW, dinfo, 0);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("potf2 x50 back-to-back: %.1f us each\n", ms * 1e3 / 50);
  // subst right on 64, 4096, 16384 rows
  for (long rows : {64L, 4096L, 16384L}) {
    double *dX; CK(hipMalloc(&dX, rows * 128 * 8)); CK(hipMemset(dX, 0, rows * 128 * 8));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      subst_right_mfma_kernel<double><<<(rows + SUBSTM_ROWS - 1) / SUBSTM_ROWS, 256>>>(dW, 128, dX, 128, rows);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("subst_right rows=%ld: %.1f us\n", rows, ms * 1e3);
    }
#ifdef LGP_SUBST_TRACE
    { unsigned long long st[16]; CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(lgp_subst_trace), sizeof(st)));
      printf("subst phases (cycles): start->"); for (int k = 1; k <= 9; ++k) printf("%llu ", st[k] - st[k - 1]); printf("\n"); }
#endif
    CK(hipFree(dX));
  }
  return 0;
}

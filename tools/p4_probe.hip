// Probe: the four-wave diagonal-block kernel (potf2m.hip) alone -- result against a host Cholesky, time alone and back to back, and
// (-DLGP_POTF2_TRACE) wave 0's cycles per phase of every block step.  Build both pivot variants and run them on ONE box:
//   for q in 1 0; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -DLGP_P4_QUAD=$q -DLGP_POTF2_TRACE tools/p4_probe.hip -o tools/p4_probe_q$q; done
#include "../lsqfitgp_amd/csrc/potf2m.hip"
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
void lgp_set_error(const char *fmt, ...) {}

int main(int argc, char **argv) {
  const int n = 128;
  const double jitter = argc > 1 ? atof(argv[1]) : 1e-3;
  std::vector<double> A(n * n), R(n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = std::exp(-0.5 * (i - j) * (i - j) / 400.0) + (i == j ? jitter : 0);
  R = A;                                                        // host reference, long double accumulation
  for (int j = 0; j < n; ++j) {
    long double d = R[j * n + j];
    for (int k = 0; k < j; ++k) d -= (long double)R[j * n + k] * R[j * n + k];
    R[j * n + j] = (double)sqrtl(d);
    for (int i = j + 1; i < n; ++i) {
      long double s = R[i * n + j];
      for (int k = 0; k < j; ++k) s -= (long double)R[i * n + k] * R[j * n + k];
      R[i * n + j] = (double)(s / R[j * n + j]);
    }
  }
  double *dA, *dW; int *dinfo;
  CK(hipMalloc(&dA, n * n * 8)); CK(hipMalloc(&dW, LGP_DBLK * 8)); CK(hipMalloc(&dinfo, 4)); CK(hipMemset(dinfo, 0, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<double> L(n * n);
  float ms;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0));
    potf2_mfma4_kernel<(LGP_P4_QUAD != 0)><<<1, 256>>>(dA, n, n, dW, dinfo, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("potf2_mfma4 alone: %.1f us\n", ms * 1e3);
  }
  CK(hipMemcpy(L.data(), dA, n * n * 8, hipMemcpyDeviceToHost));
  int info = 0; CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
  double dl = 0, nl = 0;
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { dl = fmax(dl, fabs(L[i * n + j] - R[i * n + j])); nl = fmax(nl, fabs(R[i * n + j])); }
  // the residual |L L' - A| is what a factorisation owes; |L - R| grows with the condition number
  double res = 0;
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
    long double s = 0;
    for (int k = 0; k <= j; ++k) s += (long double)L[i * n + k] * L[j * n + k];
    res = fmax(res, fabs((double)(s - A[i * n + j])));
  }
  printf("LGP_P4_QUAD=%d jitter %.1e info %d: max |L - ref| = %.3e (max |L| %.3e), max |L L' - A| = %.3e\n", LGP_P4_QUAD, jitter, info, dl, nl, res);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 50; ++i) potf2_mfma4_kernel<(LGP_P4_QUAD != 0)><<<1, 256>>>(dA, n, n, dW, dinfo, 0);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("potf2_mfma4 x50 back-to-back: %.1f us each\n", ms * 1e3 / 50);
#ifdef LGP_POTF2_TRACE
  CK(hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice));
  potf2_mfma4_kernel<(LGP_P4_QUAD != 0)><<<1, 256>>>(dA, n, n, dW, dinfo, 0);
  CK(hipDeviceSynchronize());
  unsigned long long st[64];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(lgp_p4_trace), sizeof(st)));
  printf("wave 0 (cycles): loads %llu |", st[1] - st[0]);
  unsigned long long piv = 0, trail = 0, bar = 0, pub = 0;
  for (int s2 = 0; s2 < 8; ++s2) {
    printf(" step %d: publish %llu pivots %llu barrier %llu trailing %llu |", s2, st[2 + 4 * s2] - (s2 ? st[1 + 4 * s2] : st[1]), st[3 + 4 * s2] - st[2 + 4 * s2],
           st[4 + 4 * s2] - st[3 + 4 * s2], st[5 + 4 * s2] - st[4 + 4 * s2]);
    pub += st[2 + 4 * s2] - (s2 ? st[1 + 4 * s2] : st[1]); piv += st[3 + 4 * s2] - st[2 + 4 * s2]; bar += st[4 + 4 * s2] - st[3 + 4 * s2]; trail += st[5 + 4 * s2] - st[4 + 4 * s2];
  }
  printf(" stores %llu; total %llu\nsums: publish %llu pivots %llu (%.0f per pivot) barriers %llu trailing %llu\n", st[34] - st[33], st[34] - st[0], pub, piv, piv / 128.0, bar, trail);
#endif
  return 0;
}

// Probe: duration of the chain kernels (potf2, pack, subst_right variants) alone and while the trailing-update
// gemm fills the GPU on another stream.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I lsqfitgp_amd/csrc tools/coexec_probe.hip -o tools/coexec_probe
#include "../lsqfitgp_amd/csrc/gemm.hip"
#include "../lsqfitgp_amd/csrc/potrf.hip"
#include <vector>
#include <cmath>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
void lgp_set_error(const char *fmt, ...) {}
int lgp_prof_start(lgp_ctx *, hipStream_t, double) { return 0; }
int lgp_prof_stop(lgp_ctx *, hipStream_t) { return 0; }
extern "C" const char *lgp_last_error(void) { return ""; }

int main(int argc, char **argv) {
  const int n = 128;
  const size_t m = argc > 1 ? atol(argv[1]) : 16384;
  const long xrows = 16384;
  std::vector<double> A(n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = std::exp(-0.5 * (i - j) * (i - j) / 400.0) + (i == j ? 1e-3 : 0);
  double *dA, *dW; int *dinfo;
  CK(hipMalloc(&dA, n * n * 8)); CK(hipMalloc(&dW, (size_t)LGP_DBLK * 8)); CK(hipMalloc(&dinfo, 4)); CK(hipMemset(dinfo, 0, 4));
  double *P, *C, *X;
  const size_t ldp = 1024 + 16;   // padded panel rows
  CK(hipMalloc(&P, m * ldp * 8)); CK(hipMalloc(&C, m * m * 8)); CK(hipMalloc(&X, xrows * 128 * 8));
  CK(hipMemset(P, 0, m * ldp * 8)); CK(hipMemset(C, 0, m * m * 8)); CK(hipMemset(X, 0, xrows * 128 * 8));
  lgp_ctx ctx; ctx.cus = 256;
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t e0, e1, g0, g1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&g0)); CK(hipEventCreate(&g1));
  float ms;
  auto timed = [&](const char *name, auto launch) {
    CK(hipEventRecord(e0, sb)); launch(); CK(hipEventRecord(e1, sb)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("  %s %.1f us", name, ms * 1e3);
  };
  for (int with_gemm = 0; with_gemm < 2; ++with_gemm) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice));
      CK(hipDeviceSynchronize());
      if (with_gemm) {
        CK(hipEventRecord(g0, sa));
        lgp_gemm_launch<double>(&ctx, sa, false, true, m, m, 1024, -1.0, P, ldp, P, ldp, 1.0, C, m, true, 1);
        lgp_gemm_launch<double>(&ctx, sa, false, true, m, m, 1024, -1.0, P, ldp, P, ldp, 1.0, C, m, true, 1);
        CK(hipEventRecord(g1, sa));
        hipStreamQuery(sa);
        for (volatile int spin = 0; spin < 3000000; ++spin) {}
      }
      printf("with_gemm=%d", with_gemm);
      timed("potf2", [&] { potf2_kernel<double><<<1, 256, 0, sb>>>(dA, n, n, dW, dinfo, 0); });
      timed("pack", [&] { pack_frag_kernel<double><<<(LGP_NFRAG + 3) / 4, 256, 0, sb>>>(dW); });
      timed("subst_right_mfma", [&] { subst_right_mfma_kernel<double><<<xrows / SUBSTM_ROWS, 256, 0, sb>>>(dW, 128, X, 128, xrows); });
      CK(hipDeviceSynchronize());
      if (with_gemm) { CK(hipEventElapsedTime(&ms, g0, g1)); printf("  | 2 gemms %.3f ms", ms); }
      printf("\n");
    }
  }
  return 0;
}

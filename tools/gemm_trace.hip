// Per-iteration s_memtime trace of the trailing-update gemm (wave 0 of the first 8 workgroups).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLGP_GEMM_TRACE -I lsqfitgp_amd/csrc tools/gemm_trace.hip -o tools/gemm_trace
//   tools/gemm_trace <m> <k> <ld>
#include "../lsqfitgp_amd/csrc/gemm.hip"
#include <cstdio>
#include <vector>
thread_local char lgp_err_buf[512];
void lgp_set_error(const char *fmt, ...) {}
extern "C" const char *lgp_last_error(void) { return ""; }
int main(int argc, char **argv) {
  const size_t m = argc > 1 ? atol(argv[1]) : 4096, k = argc > 2 ? atol(argv[2]) : 1024, ld = argc > 3 ? atol(argv[3]) : k;
  double *A, *C;
  unsigned long long *tr;
  hipMalloc(&A, m * ld * 8); hipMalloc(&C, m * m * 8); hipMalloc(&tr, 8 * 1024 * 8);
  std::vector<double> h(m * ld);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipMemset(C, 0, m * m * 8);
  GemmArgs<double> g;
  g.A = A; g.B = A; g.C = C; g.lda = g.ldb = (long)ld; g.ldc = (long)m; g.m = g.n = (int)m; g.k = (int)k;
  g.alpha = -1.0; g.beta = 1.0; g.lower = 1; g.tm = g.tn = (int)(m / 128); g.nsm = g.nsn = 0; g.nbulk = 0;
  g.alignedA = g.alignedB = 1; g.nsplit = 1; g.kchunk = 0; g.Cpart = nullptr; g.trace = tr; g.noload = argc > 4 ? atoi(argv[4]) : 0;
  if (g.tm >= 16) { g.nsm = g.nsn = (g.tm + 7) / 8; size_t ns = 0; for (int J = 0; J < g.nsm; ++J) ns += g.nsm - J; g.nbulk = (int)(ns / 8 * 8); }
  size_t grid = g.nsm ? 0 : (size_t)g.tm * g.tn;
  if (g.nsm) { size_t ns = 0; for (int J = 0; J < g.nsm; ++J) ns += g.nsm - J; grid = ns * 64; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(tr, 0, 8 * 1024 * 8);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((gemm_kernel<double, false, true, 128, 256, 1>), dim3((unsigned)grid), dim3(256), 0, 0, g);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("launch %d: %.3f ms (%.1f TFLOP/s with the stamps in)\n", rep, ms, (double)m * (m + 128) * k / ms / 1e9);
  }
  std::vector<unsigned long long> t(8 * 1024);
  hipMemcpy(t.data(), tr, t.size() * 8, hipMemcpyDeviceToHost);
  const int nk = (int)(k / 16);
  for (int b = 0; b < 8; b += 1) {
    const unsigned long long *q = t.data() + b * 1024;
    if (!q[1]) continue;
    double compute = 0, store = 0, barrier = 0, top = 0;
    for (int i = 0; i < nk; ++i) {
      compute += (double)(q[5 + 4 * i] - q[4 + 4 * i]);
      store += (double)(q[6 + 4 * i] - q[5 + 4 * i]);
      barrier += (double)(q[7 + 4 * i] - q[6 + 4 * i]);
      if (i + 1 < nk) top += (double)(q[4 + 4 * (i + 1)] - q[7 + 4 * i]);
    }
    // s_memtime ticks at 100 MHz: 10 ns per tick
    printf("wg %d: prologue %.2f us, loop %.2f us (loads+mfma %.2f, wait+store %.2f, barrier %.2f), per-iter %.3f us; iter samples:",
           b, (q[1] - q[0]) * 0.01, (q[2] - q[1]) * 0.01, compute * 0.01, store * 0.01, barrier * 0.01, (q[2] - q[1]) * 0.01 / nk);
    for (int i = 8; i < 14; ++i) printf(" [%llu %llu %llu]", q[5 + 4 * i] - q[4 + 4 * i], q[6 + 4 * i] - q[5 + 4 * i], q[7 + 4 * i] - q[6 + 4 * i]);
    printf("\n");
  }
  return 0;
}

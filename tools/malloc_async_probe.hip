// hipMallocAsync probe (round 4): is a stream-ordered pool block that is freed and allocated again by the next call
// safe to use the way lgp_mlnd_grad did in round 2 (everything on ONE stream: allocate, fill with several kernels, a nested
// allocate/free pair for a split-K workspace, reduce into a persistent scratch word, copy the word to pageable host memory,
// hipFreeAsync, synchronise)?  Variants, in this order (the pool's release threshold stays raised once it is):
//   0 the pattern   1 without the memset   2 without the nested pair   3 without either   4 the block also touched from a second
//   stream between events   5 = 0 with the release threshold raised (what lgp_ctx_create did)   6 = 4 with it raised
//   hipcc --offload-arch=gfx950 -O2 -o tools/malloc_async_probe tools/malloc_async_probe.hip && tools/malloc_async_probe [n] [calls]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void fill_lower(double *X, size_t n, double v) {       // X = v on and below the diagonal (the rest stays what the memset left)
  const size_t i = blockIdx.y, j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j < n && j <= i) X[i * n + j] = v;
}
__global__ void zero_kernel(double *X, size_t count) {
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < count; t += (size_t)gridDim.x * 256) X[t] = 0.0;
}
__global__ void scale_into(const double *X, double *Y, size_t count, double f) {
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < count; t += (size_t)gridDim.x * 256) Y[t] = f * X[t];
}
__global__ void partial_sums(const double *Y, size_t count, double *part) {
  double acc = 0;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < count; t += (size_t)gridDim.x * 256) acc += Y[t];
  __shared__ double sh[256];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
__global__ void final_sum(const double *part, int np, double *out) {
  double acc = 0;
  for (int t = 0; t < np; ++t) acc += part[t];
  *out = acc;
}

int main(int argc, char **argv) {
  const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4096;
  const int calls = argc > 2 ? atoi(argv[2]) : 12;
  int bad_total = 0;
  for (int variant = 0; variant < 7; ++variant) {
    const bool do_memset = variant != 1 && variant != 3, do_nested = variant != 2 && variant != 3, two_streams = variant == 4 || variant == 6;
    hipStream_t st, s2;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e1, e2;
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    if (variant >= 5) {
      hipMemPool_t pool;
      CK(hipDeviceGetDefaultMemPool(&pool, 0));
      uint64_t thr = ~0ull;
      CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
    }
    double *scratch;
    CK(hipMalloc(&scratch, 4096));
    int bad = 0;
    for (int call = 0; call < calls; ++call) {
      const double v = 1.0 + call;
      double *ws = nullptr;
      CK(hipMallocAsync(reinterpret_cast<void **>(&ws), (2 * n * n + 1024) * sizeof(double), st));
      double *X = ws, *Y = ws + n * n, *part = ws + 2 * n * n;
      if (do_memset) CK(hipMemsetAsync(X, 0, n * n * sizeof(double), st));
      else hipLaunchKernelGGL(zero_kernel, dim3(1024), dim3(256), 0, st, X, n * n);
      hipLaunchKernelGGL(fill_lower, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, st, X, n, v);
      if (do_nested) {
        double *nested = nullptr;                                 // (the split-K workspace of a product in between)
        CK(hipMallocAsync(reinterpret_cast<void **>(&nested), n * 64 * sizeof(double), st));
        hipLaunchKernelGGL(scale_into, dim3(64), dim3(256), 0, st, X, nested, n * 64, 3.0);
        CK(hipFreeAsync(nested, st));
      }
      hipStream_t sy = st;
      if (two_streams) { CK(hipEventRecord(e1, st)); CK(hipStreamWaitEvent(s2, e1, 0)); sy = s2; }
      hipLaunchKernelGGL(scale_into, dim3(1024), dim3(256), 0, sy, X, Y, n * n, 2.0);
      if (two_streams) { CK(hipEventRecord(e2, s2)); CK(hipStreamWaitEvent(st, e2, 0)); }
      hipLaunchKernelGGL(partial_sums, dim3(1024), dim3(256), 0, st, Y, n * n, part);
      hipLaunchKernelGGL(final_sum, dim3(1), dim3(1), 0, st, part, 1024, scratch);
      std::vector<double> h(1);
      CK(hipMemcpyAsync(h.data(), scratch, sizeof(double), hipMemcpyDeviceToHost, st));
      CK(hipFreeAsync(ws, st));
      CK(hipStreamSynchronize(st));
      const double want = 2.0 * v * (double)n * (double)(n + 1) / 2.0;
      if (h[0] != want) { ++bad; if (bad <= 2) printf("variant %d call %d: got %.17g want %.17g (block %p)\n", variant, call, h[0], want, (void *)ws); }
    }
    printf("variant %d: %d of %d calls wrong\n", variant, bad, calls);
    bad_total += bad;
    CK(hipFree(scratch));
    CK(hipStreamDestroy(st)); CK(hipStreamDestroy(s2));
  }
  return bad_total ? 1 : 0;
}

// elementwise.hip -- HBM-bound helpers of the hot path: block assembly (jnp.block,
// src/lsqfitgp/_GP/_elements.py:642-649), `Kxx + ycov` (_GP/_compute.py:85), `Kxsxs - Q` (:260), finite /
// symmetry checks (_elements.py:618-622, _compute.py:369-381), column sums of squares (einsum 'ji,ji->i',
// _linalg/_decomp.py:427), sum log diag and dot (:486-487).
#include "common.h"

namespace {

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

template <typename T>
__global__ void __launch_bounds__(256) axpby_kernel(size_t rows, size_t cols, T alpha, const T *A, size_t lda, T beta, const T *B,
                                                    size_t ldb, T *C, size_t ldc) {
  const size_t nbx = (cols + 255) / 256;
  for (size_t blk = blockIdx.x; blk < rows * nbx; blk += gridDim.x) {
    const size_t i = blk / nbx, j = (blk % nbx) * 256 + threadIdx.x;
    if (j < cols) {
      T v = T(0);
      if (A) v = alpha * A[i * lda + j];
      if (B) v += beta * B[i * ldb + j];
      C[i * ldc + j] = v;
    }
  }
}

// C = alpha A o B + beta C0 (Hadamard product; C0 may be null or C itself): the product of two kernels (_Kernel/_alg.py:58-82),
// and, with C0 = C, the product rule of its tangents
template <typename T>
__global__ void __launch_bounds__(256) mul_kernel(size_t rows, size_t cols, T alpha, const T *A, size_t lda, const T *B, size_t ldb, T beta,
                                                  const T *C0, size_t ldc0, T *C, size_t ldc) {
  const size_t nbx = (cols + 255) / 256;
  for (size_t blk = blockIdx.x; blk < rows * nbx; blk += gridDim.x) {
    const size_t i = blk / nbx, j = (blk % nbx) * 256 + threadIdx.x;
    if (j < cols) {
      T v = alpha * (A[i * lda + j] * B[i * ldb + j]);
      if (C0) v += beta * C0[i * ldc0 + j];
      C[i * ldc + j] = v;
    }
  }
}

// A <- a f(A) + b entry by entry, f chosen by `op` (0: asin): the outer function of kernels that are a function of another
// kernel's value (NNKernel, _kernels/_basic.py:122-154: 2 / pi asin(...))
template <typename T>
__global__ void __launch_bounds__(256) unary_kernel(size_t rows, size_t cols, T *A, size_t lda, int op, T a, T b) {
  const size_t nbx = (cols + 255) / 256;
  for (size_t blk = blockIdx.x; blk < rows * nbx; blk += gridDim.x) {
    const size_t i = blk / nbx, j = (blk % nbx) * 256 + threadIdx.x;
    if (j < cols) {
      T v = A[i * lda + j];
      if (op == 0) v = asin(v);
      A[i * lda + j] = a * v + b;
    }
  }
}

template <typename T>
__global__ void add_diag_kernel(size_t n, T *A, size_t lda, const T *vec, T alpha) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) A[i * lda + i] += vec ? vec[i] * alpha : alpha;
}

template <typename T>
__global__ void __launch_bounds__(256) add_scalar_kernel(size_t rows, size_t cols, T *A, size_t lda, T c) {
  const size_t nbx = (cols + 255) / 256;
  for (size_t blk = blockIdx.x; blk < rows * nbx; blk += gridDim.x) {
    const size_t i = blk / nbx, j = (blk % nbx) * 256 + threadIdx.x;
    if (j < cols) A[i * lda + j] += c;
  }
}

// dst = (TD) src, rows x cols with leading dimensions (mixed-precision block assembly)
template <typename TD, typename TS>
__global__ void __launch_bounds__(256) convert_kernel(size_t rows, size_t cols, TD *dst, size_t ldd, const TS *src, size_t lds) {
  const size_t nbx = (cols + 255) / 256;
  for (size_t blk = blockIdx.x; blk < rows * nbx; blk += gridDim.x) {
    const size_t i = blk / nbx, j = (blk % nbx) * 256 + threadIdx.x;
    if (j < cols) dst[i * ldd + j] = (TD)src[i * lds + j];
  }
}

// 32x32 LDS-tiled transpose: dst (cols x rows) = src^T
template <typename T>
__global__ void __launch_bounds__(256) transpose_kernel(T *dst, size_t ldd, const T *src, size_t lds, size_t rows, size_t cols) {
  __shared__ T tile[32][33];
  const size_t ntc = (cols + 31) / 32;
  const size_t tr = blockIdx.x / ntc, tc = blockIdx.x % ntc;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int k = 0; k < 32; k += 8) {
    const size_t r = tr * 32 + ty + k, c = tc * 32 + tx;
    if (r < rows && c < cols) tile[ty + k][tx] = src[r * lds + c];
  }
  __syncthreads();
  for (int k = 0; k < 32; k += 8) {
    const size_t c = tc * 32 + ty + k, r = tr * 32 + tx;
    if (r < rows && c < cols) dst[c * ldd + r] = tile[tx][ty + k];
  }
}

// flags[0] |= any non-finite; flags[1] |= any |a_ij - a_ji| > atol + rtol*|a_ji|  (jnp.allclose(A, A.T))
template <typename T>
__global__ void __launch_bounds__(256) check_kernel(size_t rows, size_t cols, const T *A, size_t lda, int check_sym, T rtol, T atol,
                                                    int *flags) {
  const size_t nbx = (cols + 255) / 256;
  int nonfinite = 0, asym = 0;
  for (size_t blk = blockIdx.x; blk < rows * nbx; blk += gridDim.x) {
    const size_t i = blk / nbx, j = (blk % nbx) * 256 + threadIdx.x;
    if (j < cols) {
      const T v = A[i * lda + j];
      if (!(fabs(v) <= (sizeof(T) == 8 ? T(1.7976931348623157e308) : T(3.4028234663852886e38)))) nonfinite = 1;
      if (check_sym && j < i) {
        const T u = A[j * lda + i];
        if (!(fabs(v - u) <= atol + rtol * fabs(u)) && !(v == u)) asym = 1;      // (equal infinities are close, as for numpy.allclose)
      }
    }
  }
  if (__any(nonfinite) && (threadIdx.x & 63) == 0) atomicOr(&flags[0], 1);
  if (__any(asym) && (threadIdx.x & 63) == 0) atomicOr(&flags[1], 1);
}

// flag |= any |a_ij - b_ij| > atol + rtol*|b_ij|  (not jnp.allclose(A, B))
template <typename T>
__global__ void __launch_bounds__(256) allclose_kernel(size_t rows, size_t cols, const T *A, size_t lda, const T *B, size_t ldb, T rtol,
                                                       T atol, int *flag) {
  const size_t nbx = (cols + 255) / 256;
  int bad = 0;
  for (size_t blk = blockIdx.x; blk < rows * nbx; blk += gridDim.x) {
    const size_t i = blk / nbx, j = (blk % nbx) * 256 + threadIdx.x;
    if (j < cols) {
      const T a = A[i * lda + j], b = B[i * ldb + j];
      if (!(fabs(a - b) <= atol + rtol * fabs(b)) && !(a == b)) bad = 1;
    }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// out[j] = sum_i A[i][j]^2.  One workgroup per (256-column strip, row chunk); partial sums by atomicAdd.
template <typename T>
__global__ void __launch_bounds__(256) colsumsq_kernel(size_t rows, size_t cols, const T *A, size_t lda, T *out, size_t rows_per_blk) {
  const size_t nbx = (cols + 255) / 256;
  const size_t strip = blockIdx.x % nbx, chunk = blockIdx.x / nbx;
  const size_t j = strip * 256 + threadIdx.x;
  if (j >= cols) return;
  const size_t r0 = chunk * rows_per_blk;
  const size_t r1 = (r0 + rows_per_blk) < rows ? (r0 + rows_per_blk) : rows;
  T acc = T(0);
  for (size_t i = r0; i < r1; ++i) {
    const T v = A[i * lda + j];
    acc += v * v;
  }
  atomicAdd(&out[j], acc);
}

// out[j] = sum_i |A[i][j]| (Gershgorin sums of a block column of a symmetric matrix)
template <typename T>
__global__ void __launch_bounds__(256) colabssum_kernel(size_t rows, size_t cols, const T *A, size_t lda, T *out, size_t rows_per_blk) {
  const size_t nbx = (cols + 255) / 256;
  const size_t strip = blockIdx.x % nbx, chunk = blockIdx.x / nbx;
  const size_t j = strip * 256 + threadIdx.x;
  if (j >= cols) return;
  const size_t r0 = chunk * rows_per_blk;
  const size_t r1 = (r0 + rows_per_blk) < rows ? (r0 + rows_per_blk) : rows;
  T acc = T(0);
  for (size_t i = r0; i < r1; ++i) acc += fabs(A[i * lda + j]);
  out[chunk * cols + j] = acc;          // partial of this row chunk; summed in chunk order by colabssum_reduce_kernel
}

// out[j] = sum over chunks, in chunk order: the Gershgorin bound (hence the jitter) is reproducible bit for bit
template <typename T>
__global__ void __launch_bounds__(256) colabssum_reduce_kernel(size_t cols, size_t chunks, const T *part, T *out) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= cols) return;
  T acc = T(0);
  for (size_t ch = 0; ch < chunks; ++ch) acc += part[ch * cols + j];
  out[j] = acc;
}

// single-workgroup reductions (n up to ~1e5: negligible), deterministic order
template <typename T>
__global__ void __launch_bounds__(1024) logdet_kernel(size_t n, const T *L, size_t ldl, T *out) {
  __shared__ T sh[16];
  T acc = T(0);
  for (size_t i = threadIdx.x; i < n; i += 1024) acc += log(L[i * ldl + i]);
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    T t = T(0);
    for (int w = 0; w < 16; ++w) t += sh[w];
    out[0] = t;
  }
}

template <typename T>
__global__ void __launch_bounds__(1024) dot_kernel(size_t n, const T *x, size_t incx, const T *y, size_t incy, T *out) {
  __shared__ T sh[16];
  T acc = T(0);
  for (size_t i = threadIdx.x; i < n; i += 1024) acc += x[i * incx] * y[i * incy];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    T t = T(0);
    for (int w = 0; w < 16; ++w) t += sh[w];
    out[0] = t;
  }
}

// Frobenius inner product sum_ij A_ij B_ij in two deterministic stages: per-workgroup partial sums over row chunks, then one
// workgroup over the partials (tr(K^-1 dK) = <K^-1, dK> and tr(K^-1 dK K^-1 dK') = <K^-1 dK K^-1, dK'>, _decomp.py:526,553)
template <typename T>
__global__ void __launch_bounds__(256) frob_partial_kernel(size_t rows, size_t cols, const T *A, size_t lda, const T *B, size_t ldb, T *part,
                                                           size_t rows_per_blk) {
  __shared__ T sh[4];
  const size_t r0 = (size_t)blockIdx.x * rows_per_blk;
  const size_t r1 = (r0 + rows_per_blk) < rows ? (r0 + rows_per_blk) : rows;
  T acc = T(0);
  for (size_t i = r0; i < r1; ++i)
    for (size_t j = threadIdx.x; j < cols; j += 256) acc += A[i * lda + j] * B[i * ldb + j];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

template <typename T>
__global__ void __launch_bounds__(256) sum_partials_kernel(size_t count, const T *part, T *out) {
  __shared__ T sh[4];
  T acc = T(0);
  for (size_t i = threadIdx.x; i < count; i += 256) acc += part[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

inline unsigned grid_for(size_t rows, size_t cols) {
  const size_t nb = rows * ((cols + 255) / 256);
  return (unsigned)(nb < 1048576 ? (nb ? nb : 1) : 1048576);
}


template <typename T>
int axpby_impl(lgp_ctx *ctx, size_t rows, size_t cols, T alpha, const T *A, size_t lda, T beta, const T *B, size_t ldb, T *C, size_t ldc) {
  LGP_REQUIRE(ctx != nullptr, "null context");
  if (rows == 0 || cols == 0) return LGP_OK;
  LGP_REQUIRE(C != nullptr && ldc >= cols, "bad output");
  LGP_REQUIRE((A == nullptr || lda >= cols) && (B == nullptr || ldb >= cols), "leading dimension too small");
  DeviceGuard dg(ctx->device);
  hipLaunchKernelGGL((axpby_kernel<T>), dim3(grid_for(rows, cols)), dim3(256), 0, ctx->stream, rows, cols, alpha, A, lda, beta, B, ldb, C, ldc);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

template <typename T>
int mul_impl(lgp_ctx *ctx, size_t rows, size_t cols, T alpha, const T *A, size_t lda, const T *B, size_t ldb, T beta, const T *C0, size_t ldc0,
             T *C, size_t ldc) {
  LGP_REQUIRE(ctx != nullptr, "null context");
  if (rows == 0 || cols == 0) return LGP_OK;
  LGP_REQUIRE(A != nullptr && B != nullptr && C != nullptr && lda >= cols && ldb >= cols && ldc >= cols, "bad operands");
  LGP_REQUIRE(C0 == nullptr || ldc0 >= cols, "leading dimension too small");
  DeviceGuard dg(ctx->device);
  hipLaunchKernelGGL((mul_kernel<T>), dim3(grid_for(rows, cols)), dim3(256), 0, ctx->stream, rows, cols, alpha, A, lda, B, ldb, beta, C0, ldc0, C, ldc);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

template <typename T>
int unary_impl(lgp_ctx *ctx, size_t rows, size_t cols, T *A, size_t lda, int op, T a, T b) {
  LGP_REQUIRE(ctx != nullptr, "null context");
  if (rows == 0 || cols == 0) return LGP_OK;
  LGP_REQUIRE(A != nullptr && lda >= cols, "bad matrix");
  LGP_REQUIRE(op == 0, "unknown function");
  DeviceGuard dg(ctx->device);
  hipLaunchKernelGGL((unary_kernel<T>), dim3(grid_for(rows, cols)), dim3(256), 0, ctx->stream, rows, cols, A, lda, op, a, b);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

template <typename T>
int add_diag_impl(lgp_ctx *ctx, size_t n, T *A, size_t lda, const T *vec, T alpha) {
  LGP_REQUIRE(ctx != nullptr, "null context");
  if (n == 0) return LGP_OK;
  LGP_REQUIRE(A != nullptr && lda >= n, "bad matrix");
  DeviceGuard dg(ctx->device);
  hipLaunchKernelGGL((add_diag_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, A, lda, vec, alpha);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

template <typename T>
int add_scalar_impl(lgp_ctx *ctx, size_t rows, size_t cols, T *A, size_t lda, T c) {
  LGP_REQUIRE(ctx != nullptr, "null context");
  if (rows == 0 || cols == 0) return LGP_OK;
  LGP_REQUIRE(A != nullptr && lda >= cols, "bad matrix");
  DeviceGuard dg(ctx->device);
  hipLaunchKernelGGL((add_scalar_kernel<T>), dim3(grid_for(rows, cols)), dim3(256), 0, ctx->stream, rows, cols, A, lda, c);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

template <typename TD, typename TS>
int convert_impl(lgp_ctx *ctx, size_t rows, size_t cols, TD *dst, size_t ldd, const TS *src, size_t lds) {
  LGP_REQUIRE(ctx != nullptr, "null context");
  if (rows == 0 || cols == 0) return LGP_OK;
  LGP_REQUIRE(dst != nullptr && src != nullptr && ldd >= cols && lds >= cols, "bad matrix");
  DeviceGuard dg(ctx->device);
  hipLaunchKernelGGL((convert_kernel<TD, TS>), dim3(grid_for(rows, cols)), dim3(256), 0, ctx->stream, rows, cols, dst, ldd, src, lds);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

template <typename T>
int transpose_impl(lgp_ctx *ctx, T *dst, size_t ldd, const T *src, size_t lds, size_t rows, size_t cols) {
  LGP_REQUIRE(ctx != nullptr, "null context");
  if (rows == 0 || cols == 0) return LGP_OK;
  LGP_REQUIRE(dst != nullptr && src != nullptr && ldd >= rows && lds >= cols, "bad matrix");
  const size_t nt = ((rows + 31) / 32) * ((cols + 31) / 32);
  LGP_REQUIRE(nt < (1ull << 31), "matrix too large");
  DeviceGuard dg(ctx->device);
  hipLaunchKernelGGL((transpose_kernel<T>), dim3((unsigned)nt), dim3(256), 0, ctx->stream, dst, ldd, src, lds, rows, cols);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

template <typename T>
int check_impl(lgp_ctx *ctx, size_t rows, size_t cols, const T *A, size_t lda, int check_sym, T rtol, T atol, int *hflags) {
  LGP_REQUIRE(ctx != nullptr && hflags != nullptr, "null pointer");
  hflags[0] = hflags[1] = 0;
  if (rows == 0 || cols == 0) return LGP_OK;
  LGP_REQUIRE(A != nullptr && lda >= cols, "bad matrix");
  LGP_REQUIRE(!check_sym || rows == cols, "symmetry check needs a square matrix");
  DeviceGuard dg(ctx->device);
  int *dflags = reinterpret_cast<int *>(ctx->d_scratch);
  LGP_HIP(hipMemsetAsync(dflags, 0, 2 * sizeof(int), ctx->stream));
  hipLaunchKernelGGL((check_kernel<T>), dim3(grid_for(rows, cols)), dim3(256), 0, ctx->stream, rows, cols, A, lda, check_sym, rtol, atol, dflags);
  LGP_LAUNCH_CHECK();
  LGP_HIP(hipMemcpyAsync(hflags, dflags, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  LGP_HIP(hipStreamSynchronize(ctx->stream));
  return LGP_OK;
}

// the same check without the read-back: the two flags are OR-ed into caller-owned device words (zeroed by the caller), so
// that a caller with many blocks to check reads them all at once
template <typename T>
int check_dev_impl(lgp_ctx *ctx, size_t rows, size_t cols, const T *A, size_t lda, int check_sym, T rtol, T atol, int *dflags) {
  LGP_REQUIRE(ctx != nullptr && dflags != nullptr, "null pointer");
  if (rows == 0 || cols == 0) return LGP_OK;
  LGP_REQUIRE(A != nullptr && lda >= cols, "bad matrix");
  LGP_REQUIRE(!check_sym || rows == cols, "symmetry check needs a square matrix");
  DeviceGuard dg(ctx->device);
  hipLaunchKernelGGL((check_kernel<T>), dim3(grid_for(rows, cols)), dim3(256), 0, ctx->stream, rows, cols, A, lda, check_sym, rtol, atol, dflags);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

template <typename T>
int allclose_impl(lgp_ctx *ctx, size_t rows, size_t cols, const T *A, size_t lda, const T *B, size_t ldb, T rtol, T atol, int *hflag) {
  LGP_REQUIRE(ctx != nullptr && hflag != nullptr, "null pointer");
  *hflag = 1;
  if (rows == 0 || cols == 0) return LGP_OK;
  LGP_REQUIRE(A != nullptr && B != nullptr && lda >= cols && ldb >= cols, "bad matrix");
  DeviceGuard dg(ctx->device);
  int *dflag = reinterpret_cast<int *>(ctx->d_scratch);
  LGP_HIP(hipMemsetAsync(dflag, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL((allclose_kernel<T>), dim3(grid_for(rows, cols)), dim3(256), 0, ctx->stream, rows, cols, A, lda, B, ldb, rtol, atol, dflag);
  LGP_LAUNCH_CHECK();
  int bad = 0;
  LGP_HIP(hipMemcpyAsync(&bad, dflag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  LGP_HIP(hipStreamSynchronize(ctx->stream));
  *hflag = bad ? 0 : 1;
  return LGP_OK;
}

template <typename T>
int logdet_impl(lgp_ctx *ctx, size_t n, const T *L, size_t ldl, T *hout) {
  LGP_REQUIRE(ctx != nullptr && hout != nullptr && L != nullptr && ldl >= n, "bad argument");
  DeviceGuard dg(ctx->device);
  T *d = reinterpret_cast<T *>(ctx->d_scratch);
  hipLaunchKernelGGL((logdet_kernel<T>), dim3(1), dim3(1024), 0, ctx->stream, n, L, ldl, d);
  LGP_LAUNCH_CHECK();
  LGP_HIP(hipMemcpyAsync(hout, d, sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
  LGP_HIP(hipStreamSynchronize(ctx->stream));
  return LGP_OK;
}

template <typename T>
int dot_impl(lgp_ctx *ctx, size_t n, const T *x, const T *y, T *hout) {
  LGP_REQUIRE(ctx != nullptr && hout != nullptr && (n == 0 || (x != nullptr && y != nullptr)), "bad argument");
  DeviceGuard dg(ctx->device);
  T *d = reinterpret_cast<T *>(ctx->d_scratch);
  hipLaunchKernelGGL((dot_kernel<T>), dim3(1), dim3(1024), 0, ctx->stream, n, x, (size_t)1, y, (size_t)1, d);
  LGP_LAUNCH_CHECK();
  LGP_HIP(hipMemcpyAsync(hout, d, sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
  LGP_HIP(hipStreamSynchronize(ctx->stream));
  return LGP_OK;
}

}  // namespace

static unsigned grid_for_public(size_t rows, size_t cols) { return grid_for(rows, cols); }

template <typename T>
int lgp_colsumsq_impl(lgp_ctx *ctx, hipStream_t stream, size_t rows, size_t cols, const T *A, size_t lda, T *out) {
  if (cols == 0) return LGP_OK;
  LGP_REQUIRE(out != nullptr && (rows == 0 || (A != nullptr && lda >= cols)), "bad argument");
  LGP_HIP(hipMemsetAsync(out, 0, cols * sizeof(T), stream));
  if (rows == 0) return LGP_OK;
  const size_t nbx = (cols + 255) / 256;
  size_t chunks = (size_t)ctx->cus * 8 / nbx;
  if (chunks < 1) chunks = 1;
  if (chunks > rows) chunks = rows;
  const size_t rpb = (rows + chunks - 1) / chunks;
  chunks = (rows + rpb - 1) / rpb;
  hipLaunchKernelGGL((colsumsq_kernel<T>), dim3((unsigned)(nbx * chunks)), dim3(256), 0, stream, rows, cols, A, lda, out, rpb);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}
template int lgp_colsumsq_impl<double>(lgp_ctx *, hipStream_t, size_t, size_t, const double *, size_t, double *);
template int lgp_colsumsq_impl<float>(lgp_ctx *, hipStream_t, size_t, size_t, const float *, size_t, float *);

template <typename T>
int lgp_dot_strided_device(lgp_ctx *ctx, hipStream_t stream, size_t n, const T *x, size_t incx, const T *y, size_t incy, T *dout) {
  hipLaunchKernelGGL((dot_kernel<T>), dim3(1), dim3(1024), 0, stream, n, x, incx, y, incy, dout);
  LGP_LAUNCH_CHECK();
  (void)ctx;
  return LGP_OK;
}
template int lgp_dot_strided_device<double>(lgp_ctx *, hipStream_t, size_t, const double *, size_t, const double *, size_t, double *);
template int lgp_dot_strided_device<float>(lgp_ctx *, hipStream_t, size_t, const float *, size_t, const float *, size_t, float *);

template <typename T>
int lgp_logdet_device(lgp_ctx *ctx, hipStream_t stream, size_t n, const T *L, size_t ldl, T *dout) {
  hipLaunchKernelGGL((logdet_kernel<T>), dim3(1), dim3(1024), 0, stream, n, L, ldl, dout);
  LGP_LAUNCH_CHECK();
  (void)ctx;
  return LGP_OK;
}
template int lgp_logdet_device<double>(lgp_ctx *, hipStream_t, size_t, const double *, size_t, double *);
template int lgp_logdet_device<float>(lgp_ctx *, hipStream_t, size_t, const float *, size_t, float *);

extern "C" int lgp_colabssum_f64(lgp_ctx *c, size_t rows, size_t cols, const double *A, size_t lda, double *out) {
  LGP_REQUIRE(c != nullptr, "null context");
  if (cols == 0) return LGP_OK;
  LGP_REQUIRE(out != nullptr && (rows == 0 || (A != nullptr && lda >= cols)), "bad argument");
  DeviceGuard dg(c->device);
  if (rows == 0) {
    LGP_HIP(hipMemsetAsync(out, 0, cols * sizeof(double), c->stream));
    return LGP_OK;
  }
  const size_t nbx = (cols + 255) / 256;
  size_t chunks = (size_t)c->cus * 8 / nbx;
  if (chunks < 1) chunks = 1;
  if (chunks > rows) chunks = rows;
  const size_t rpb = (rows + chunks - 1) / chunks;
  chunks = (rows + rpb - 1) / rpb;
  double *part = nullptr;
  LGP_CHECK(lgp_ws_alloc(c, c->stream, chunks * cols * sizeof(double), reinterpret_cast<void **>(&part)));
  WsGuard part_guard(c, c->stream, part);
  hipLaunchKernelGGL((colabssum_kernel<double>), dim3((unsigned)(nbx * chunks)), dim3(256), 0, c->stream, rows, cols, A, lda, part, rpb);
  LGP_LAUNCH_CHECK();
  hipLaunchKernelGGL((colabssum_reduce_kernel<double>), dim3((unsigned)nbx), dim3(256), 0, c->stream, cols, chunks, part, out);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

extern "C" {
int lgp_axpby_f64(lgp_ctx *c, size_t r, size_t n, double a, const double *A, size_t lda, double b, const double *B, size_t ldb, double *C, size_t ldc) { return axpby_impl<double>(c, r, n, a, A, lda, b, B, ldb, C, ldc); }
int lgp_axpby_f32(lgp_ctx *c, size_t r, size_t n, float a, const float *A, size_t lda, float b, const float *B, size_t ldb, float *C, size_t ldc) { return axpby_impl<float>(c, r, n, a, A, lda, b, B, ldb, C, ldc); }
int lgp_mul_f64(lgp_ctx *c, size_t r, size_t n, double a, const double *A, size_t lda, const double *B, size_t ldb, double b, const double *C0, size_t ldc0, double *C, size_t ldc) { return mul_impl<double>(c, r, n, a, A, lda, B, ldb, b, C0, ldc0, C, ldc); }
int lgp_mul_f32(lgp_ctx *c, size_t r, size_t n, float a, const float *A, size_t lda, const float *B, size_t ldb, float b, const float *C0, size_t ldc0, float *C, size_t ldc) { return mul_impl<float>(c, r, n, a, A, lda, B, ldb, b, C0, ldc0, C, ldc); }
int lgp_unary_f64(lgp_ctx *c, size_t r, size_t n, double *A, size_t lda, int op, double a, double b) { return unary_impl<double>(c, r, n, A, lda, op, a, b); }
int lgp_unary_f32(lgp_ctx *c, size_t r, size_t n, float *A, size_t lda, int op, float a, float b) { return unary_impl<float>(c, r, n, A, lda, op, a, b); }
int lgp_add_diag_f64(lgp_ctx *c, size_t n, double *A, size_t lda, const double *v, double a) { return add_diag_impl<double>(c, n, A, lda, v, a); }
int lgp_add_diag_f32(lgp_ctx *c, size_t n, float *A, size_t lda, const float *v, float a) { return add_diag_impl<float>(c, n, A, lda, v, a); }
int lgp_add_scalar_f64(lgp_ctx *c, size_t r, size_t n, double *A, size_t lda, double v) { return add_scalar_impl<double>(c, r, n, A, lda, v); }
int lgp_add_scalar_f32(lgp_ctx *c, size_t r, size_t n, float *A, size_t lda, float v) { return add_scalar_impl<float>(c, r, n, A, lda, v); }
int lgp_convert_f32_f64(lgp_ctx *c, size_t rows, size_t cols, double *d, size_t ldd, const float *s, size_t lds) { return convert_impl<double, float>(c, rows, cols, d, ldd, s, lds); }
int lgp_convert_f64_f32(lgp_ctx *c, size_t rows, size_t cols, float *d, size_t ldd, const double *s, size_t lds) { return convert_impl<float, double>(c, rows, cols, d, ldd, s, lds); }
int lgp_transpose_f64(lgp_ctx *c, double *d, size_t ldd, const double *s, size_t lds, size_t r, size_t n) { return transpose_impl<double>(c, d, ldd, s, lds, r, n); }
int lgp_transpose_f32(lgp_ctx *c, float *d, size_t ldd, const float *s, size_t lds, size_t r, size_t n) { return transpose_impl<float>(c, d, ldd, s, lds, r, n); }
int lgp_check_matrix_f64(lgp_ctx *c, size_t r, size_t n, const double *A, size_t lda, int sym, double rtol, double atol, int *f) { return check_impl<double>(c, r, n, A, lda, sym, rtol, atol, f); }
int lgp_check_matrix_f32(lgp_ctx *c, size_t r, size_t n, const float *A, size_t lda, int sym, float rtol, float atol, int *f) { return check_impl<float>(c, r, n, A, lda, sym, rtol, atol, f); }
int lgp_check_matrix_dev_f64(lgp_ctx *c, size_t r, size_t n, const double *A, size_t lda, int sym, double rtol, double atol, int *df) { return check_dev_impl<double>(c, r, n, A, lda, sym, rtol, atol, df); }
int lgp_check_matrix_dev_f32(lgp_ctx *c, size_t r, size_t n, const float *A, size_t lda, int sym, float rtol, float atol, int *df) { return check_dev_impl<float>(c, r, n, A, lda, sym, rtol, atol, df); }
int lgp_allclose_f64(lgp_ctx *c, size_t r, size_t n, const double *A, size_t lda, const double *B, size_t ldb, double rtol, double atol, int *f) { return allclose_impl<double>(c, r, n, A, lda, B, ldb, rtol, atol, f); }
int lgp_allclose_f32(lgp_ctx *c, size_t r, size_t n, const float *A, size_t lda, const float *B, size_t ldb, float rtol, float atol, int *f) { return allclose_impl<float>(c, r, n, A, lda, B, ldb, rtol, atol, f); }
int lgp_colsumsq_f64(lgp_ctx *c, size_t r, size_t n, const double *A, size_t lda, double *o) {
  LGP_REQUIRE(c != nullptr, "null context");
  DeviceGuard dg(c->device);
  return lgp_colsumsq_impl<double>(c, c->stream, r, n, A, lda, o);
}
int lgp_colsumsq_f32(lgp_ctx *c, size_t r, size_t n, const float *A, size_t lda, float *o) {
  LGP_REQUIRE(c != nullptr, "null context");
  DeviceGuard dg(c->device);
  return lgp_colsumsq_impl<float>(c, c->stream, r, n, A, lda, o);
}
int lgp_logdet_diag_f64(lgp_ctx *c, size_t n, const double *L, size_t ldl, double *h) { return logdet_impl<double>(c, n, L, ldl, h); }
int lgp_logdet_diag_f32(lgp_ctx *c, size_t n, const float *L, size_t ldl, float *h) { return logdet_impl<float>(c, n, L, ldl, h); }
int lgp_dot_f64(lgp_ctx *c, size_t n, const double *x, const double *y, double *h) { return dot_impl<double>(c, n, x, y, h); }
int lgp_dot_f32(lgp_ctx *c, size_t n, const float *x, const float *y, float *h) { return dot_impl<float>(c, n, x, y, h); }
}

// dout[0] = sum_ij A_ij B_ij; `dpart` is a device workspace of at least 1024 values
template <typename T>
int lgp_frob_device(lgp_ctx *ctx, hipStream_t stream, size_t rows, size_t cols, const T *A, size_t lda, const T *B, size_t ldb, T *dpart, T *dout) {
  (void)ctx;
  size_t chunks = rows < 1024 ? (rows ? rows : 1) : 1024;
  const size_t rpb = (rows + chunks - 1) / chunks;
  chunks = rows ? (rows + rpb - 1) / rpb : 1;
  hipLaunchKernelGGL((frob_partial_kernel<T>), dim3((unsigned)chunks), dim3(256), 0, stream, rows, cols, A, lda, B, ldb, dpart, rpb ? rpb : 1);
  LGP_LAUNCH_CHECK();
  hipLaunchKernelGGL((sum_partials_kernel<T>), dim3(1), dim3(256), 0, stream, chunks, dpart, dout);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}
template int lgp_frob_device<double>(lgp_ctx *, hipStream_t, size_t, size_t, const double *, size_t, const double *, size_t, double *, double *);
template int lgp_frob_device<float>(lgp_ctx *, hipStream_t, size_t, size_t, const float *, size_t, const float *, size_t, float *, float *);

namespace {
template <typename T>
__global__ void set_diag_kernel(size_t n, T *A, size_t lda, T value) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) A[i * lda + i] = value;
}
}  // namespace
template <typename T>
int lgp_set_diag_device(lgp_ctx *ctx, hipStream_t stream, size_t n, T *A, size_t lda, T value) {
  (void)ctx;
  hipLaunchKernelGGL((set_diag_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, A, lda, value);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}
template int lgp_set_diag_device<double>(lgp_ctx *, hipStream_t, size_t, double *, size_t, double);
template int lgp_set_diag_device<float>(lgp_ctx *, hipStream_t, size_t, float *, size_t, float);

// A_ij <- A_ij (/ or *) row_i (/ or *) col_j: the symmetric power-of-two scaling K / s / s[:, None] of a locally owned block column
// and the final L * s[:, None] of the distributed factorisation (_decomp.py:385,392); drow / dcol may be NULL
namespace {
__global__ void __launch_bounds__(256) scale_rc_kernel(size_t rows, size_t cols, double *A, size_t lda, const double *row, const double *col, int divide) {
  const size_t nbx = (cols + 255) / 256;
  for (size_t blk = blockIdx.x; blk < rows * nbx; blk += gridDim.x) {
    const size_t i = blk / nbx, j = (blk % nbx) * 256 + threadIdx.x;
    if (j < cols) {
      double v = A[i * lda + j];
      if (divide) {
        if (col) v = v / col[j];
        if (row) v = v / row[i];
      } else {
        if (col) v = v * col[j];
        if (row) v = v * row[i];
      }
      A[i * lda + j] = v;
    }
  }
}
// s_j = 2^rint(log2(d_j) / 2) where d_j != 0 else 1 from the diagonal elements d (_decomp.py:356-361)
__global__ void diag_pow2_kernel(size_t n, const double *A, size_t lda, double *s) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const double d = A[j * lda + j];
  s[j] = d != 0.0 ? exp2(rint(0.5 * log2(d))) : 1.0;
}
}  // namespace

extern "C" int lgp_scale_rows_cols_f64(lgp_ctx *ctx, size_t rows, size_t cols, double *dA, size_t lda, const double *drow, const double *dcol, int divide) {
  LGP_REQUIRE(ctx != nullptr && (dA != nullptr || rows * cols == 0), "null pointer");
  if (rows == 0 || cols == 0) return LGP_OK;
  LGP_REQUIRE(lda >= cols, "lda < cols");
  DeviceGuard dg(ctx->device);
  hipLaunchKernelGGL(scale_rc_kernel, dim3(grid_for_public(rows, cols)), dim3(256), 0, ctx->stream, rows, cols, dA, lda, drow, dcol, divide);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

extern "C" int lgp_diag_scale_pow2_f64(lgp_ctx *ctx, size_t n, const double *dA, size_t lda, double *ds) {
  LGP_REQUIRE(ctx != nullptr && dA != nullptr && ds != nullptr, "null pointer");
  if (n == 0) return LGP_OK;
  DeviceGuard dg(ctx->device);
  hipLaunchKernelGGL(diag_pow2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, dA, lda, ds);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

namespace {
__global__ void __launch_bounds__(1024) max_kernel(size_t n, const double *x, double *out) {
  __shared__ double sh[16];
  __shared__ int shnan[16];
  double m = -INFINITY;
  int nan = 0;
  for (size_t i = threadIdx.x; i < n; i += 1024) {
    const double v = x[i];
    nan |= (v != v);
    m = v > m ? v : m;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(m, off, 64);
    m = o > m ? o : m;
    nan |= __shfl_down(nan, off, 64);
  }
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = m; shnan[threadIdx.x >> 6] = nan; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = -INFINITY;
    int anynan = 0;
    for (int w = 0; w < 16; ++w) { t = sh[w] > t ? sh[w] : t; anynan |= shnan[w]; }
    out[0] = anynan ? (double)NAN : t;
  }
}
}  // namespace
extern "C" int lgp_max_f64(lgp_ctx *ctx, size_t n, const double *dx, double *dout) {
  LGP_REQUIRE(ctx != nullptr && dout != nullptr && (dx != nullptr || n == 0), "null pointer");
  DeviceGuard dg(ctx->device);
  hipLaunchKernelGGL(max_kernel, dim3(1), dim3(1024), 0, ctx->stream, n, dx, dout);
  LGP_LAUNCH_CHECK();
  return LGP_OK;
}

// ---- extreme eigenvalues of a symmetric matrix by Lanczos iterations, entirely on the device (GP(checkpos=True),
// _GP/_elements.py:651-668: the reference runs two LOBPCG iterations on a host copy).  Two launches per step, no host round trip:
//   matvec   w = K q_j - beta_{j-1} q_{j-1}           one wave per row, the vectors in double whatever the matrix's type
//   step     alpha_j = q_j . w;  w -= alpha_j q_j;  beta_j = |w|;  q_{j+1} = w / beta_j     one workgroup, sums in fixed order
// The three-term recurrence without reorthogonalisation: the Ritz values stay inside the spectrum (what the check needs; copies of
// converged ones may appear among 24 steps, which does not move the extremes).
namespace {
template <typename T>
__global__ void __launch_bounds__(256) lanczos_matvec_kernel(size_t n, const T *__restrict__ K, size_t ldk, const double *__restrict__ q,
                                                            const double *__restrict__ qprev, const double *bprev, double *__restrict__ w) {
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  const T *kr = K + row * ldk;
  double acc = 0.;
  for (size_t c = lane; c < n; c += 64) acc += (double)kr[c] * q[c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) w[row] = qprev != nullptr ? acc - bprev[0] * qprev[row] : acc;
}

__device__ __forceinline__ double block_sum_1024(double v, double *sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();                                       // (sh may still be read from the previous sum)
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.;
  for (int i = 0; i < 16; ++i) t += sh[i];
  return t;
}

__global__ void __launch_bounds__(1024) lanczos_step_kernel(size_t n, const double *__restrict__ q, double *__restrict__ w, double *alpha, double *beta,
                                                           const double *alpha0) {
  __shared__ double sh[16];
  double acc = 0.;
  for (size_t i = threadIdx.x; i < n; i += 1024) acc += q[i] * w[i];
  const double a = block_sum_1024(acc, sh);
  acc = 0.;
  for (size_t i = threadIdx.x; i < n; i += 1024) {
    const double v = w[i] - a * q[i];
    w[i] = v;
    acc += v * v;
  }
  const double b = sqrt(block_sum_1024(acc, sh));
  // an invariant subspace: the recurrence ends (the host cuts the tridiagonal matrix at the first such beta); later steps see q = 0
  const double ref = fabs(alpha0 != nullptr ? alpha0[0] : a);
  const double inv = b > 1e-14 * (ref > 1. ? ref : 1.) ? 1. / b : 0.;
  for (size_t i = threadIdx.x; i < n; i += 1024) w[i] *= inv;
  if (threadIdx.x == 0) { alpha[0] = a; beta[0] = inv != 0. ? b : 0.; }
}

template <typename T>
int lanczos_impl(lgp_ctx *ctx, size_t n, const T *dK, size_t ldk, size_t steps, const double *hq0, double *halpha, double *hbeta) {
  LGP_REQUIRE(ctx != nullptr && dK != nullptr && hq0 != nullptr && halpha != nullptr && hbeta != nullptr, "null pointer");
  LGP_REQUIRE(n >= 1 && ldk >= n && steps >= 1 && steps <= 1024, "bad sizes");
  DeviceGuard dg(ctx->device);
  hipStream_t st = ctx->stream;
  double *ws = nullptr;
  LGP_CHECK(lgp_ws_alloc(ctx, st, (3 * n + 2 * steps) * sizeof(double), reinterpret_cast<void **>(&ws)));
  WsGuard guard(ctx, st, ws);
  double *buf[3] = {ws, ws + n, ws + 2 * n}, *dalpha = ws + 3 * n, *dbeta = dalpha + steps;
  LGP_HIP(hipMemcpyAsync(buf[0], hq0, n * sizeof(double), hipMemcpyHostToDevice, st));
  for (size_t j = 0; j < steps; ++j) {
    double *q = buf[j % 3], *w = buf[(j + 1) % 3], *qprev = j ? buf[(j + 2) % 3] : nullptr;
    hipLaunchKernelGGL((lanczos_matvec_kernel<T>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, n, dK, ldk, q, qprev, j ? dbeta + (j - 1) : nullptr, w);
    LGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(lanczos_step_kernel, dim3(1), dim3(1024), 0, st, n, q, w, dalpha + j, dbeta + j, j ? dalpha : nullptr);
    LGP_LAUNCH_CHECK();
  }
  LGP_HIP(hipMemcpyAsync(halpha, dalpha, steps * sizeof(double), hipMemcpyDeviceToHost, st));
  LGP_HIP(hipMemcpyAsync(hbeta, dbeta, steps * sizeof(double), hipMemcpyDeviceToHost, st));
  LGP_HIP(hipStreamSynchronize(st));
  return LGP_OK;
}
}  // namespace

extern "C" int lgp_lanczos_f64(lgp_ctx *ctx, size_t n, const double *dK, size_t ldk, size_t steps, const double *hq0, double *halpha, double *hbeta) {
  return lanczos_impl<double>(ctx, n, dK, ldk, steps, hq0, halpha, hbeta);
}
extern "C" int lgp_lanczos_f32(lgp_ctx *ctx, size_t n, const float *dK, size_t ldk, size_t steps, const double *hq0, double *halpha, double *hbeta) {
  return lanczos_impl<float>(ctx, n, dK, ldk, steps, hq0, halpha, hbeta);
}

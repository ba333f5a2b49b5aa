// Probe 2: cycles per f64 MFMA / v_fma_f64, effective clock, and MFMA||VALU co-execution on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// mode: 0 = all waves mfma16x16x4, 1 = all waves fma, 2 = even waves mfma / odd waves fma (by wave id in block),
//       3 = each wave interleaves NF fma per mfma, 4 = mfma 4x4x4
template <int MODE, int NF>
__global__ void __launch_bounds__(512) probe(double* out, unsigned long long* stamps, int iters, double seed) {
  d4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
  double f[16];
  for (int i = 0; i < 16; ++i) f[i] = i + seed;
  double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3, c = 1e-9;
  int wave = threadIdx.x >> 6;
  bool do_mfma = (MODE == 0) || (MODE == 3) || (MODE == 2 && (wave & 1) == 0) || (MODE == 4);
  bool do_fma = (MODE == 1) || (MODE == 2 && (wave & 1) == 1);
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 4) {
          double r = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i][0], 0, 0, 0);
          acc[i][0] = r;
        } else {
          acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        if (MODE == 3) {
#pragma unroll
          for (int j = 0; j < NF; ++j) f[(i * NF + j) & 15] = __builtin_fma(a, f[(i * NF + j) & 15], c);
        }
      }
    }
  } else if (do_fma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 8; ++rep)
#pragma unroll
        for (int i = 0; i < 16; ++i) f[i] = __builtin_fma(a, f[i], c);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    stamps[(blockIdx.x * (blockDim.x >> 6) + wave) * 2] = t1 - t0;
    stamps[(blockIdx.x * (blockDim.x >> 6) + wave) * 2 + 1] = r1 - r0;
  }
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int ncu = p.multiProcessorCount;
  double* dout; unsigned long long* dst; CK(hipMalloc(&dout, ncu * 8 * 512 * 8)); CK(hipMalloc(&dst, ncu * 8 * 16 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int iters = 2048;
  auto run = [&](auto kern, const char* name, int blocks, int threads, double mfma_per_wave_iter, double fma_per_wave_iter, double mfma_flops) {
    float best = 1e9; 
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, dout, dst, iters, 1.0 + rep);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    int nw = blocks * threads / 64;
    std::vector<unsigned long long> st(nw * 2); CK(hipMemcpy(st.data(), dst, nw * 16, hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0; for (int i = 0; i < nw; ++i) { cyc += st[2 * i]; rt += st[2 * i + 1]; } cyc /= nw; rt /= nw;
    double waves = nw;
    double flops = waves * iters * (mfma_per_wave_iter * mfma_flops + fma_per_wave_iter * 128.0);
    printf("%-44s blk=%4d thr=%3d: %.3f ms  %.2f TF  wave-cycles/iter %.1f  clock %.0f MHz\n", name, blocks, threads, best, flops / best * 1e-9, cyc / iters, cyc / rt * 100.0);
  };
  // all-MFMA at 1,2 waves/SIMD
  run(probe<0, 0>, "mfma16x16x4 (8/iter) 1 wave/SIMD", ncu, 256, 8, 0, 2048);
  run(probe<0, 0>, "mfma16x16x4 (8/iter) 2 wave/SIMD", ncu, 512, 8, 0, 2048);
  run(probe<4, 0>, "mfma4x4x4 (8/iter) 1 wave/SIMD", ncu, 256, 8, 0, 512);
  run(probe<4, 0>, "mfma4x4x4 (8/iter) 2 wave/SIMD", ncu, 512, 8, 0, 512);
  run(probe<1, 0>, "v_fma_f64 (128/iter) 1 wave/SIMD", ncu, 256, 0, 128, 0);
  run(probe<1, 0>, "v_fma_f64 (128/iter) 2 wave/SIMD", ncu, 512, 0, 128, 0);
  run(probe<1, 0>, "v_fma_f64 (128/iter) 4 wave/SIMD", ncu * 2, 512, 0, 128, 0);
  // MODE 2 with 512 threads: waves 0,2,4,6 mfma; 1,3,5,7 fma. wave->SIMD mapping cyclic, so each SIMD likely gets one of each kind or two of the same
  run(probe<2, 0>, "mixed waves: even mfma / odd fma (8 waves)", ncu, 512, 4, 64, 2048);
  run(probe<3, 1>, "interleave 1 fma per mfma", ncu, 256, 8, 8, 2048);
  run(probe<3, 4>, "interleave 4 fma per mfma", ncu, 256, 8, 32, 2048);
  run(probe<3, 8>, "interleave 8 fma per mfma", ncu, 256, 8, 64, 2048);
  run(probe<3, 16>, "interleave 16 fma per mfma", ncu, 256, 8, 128, 2048);
  run(probe<3, 16>, "interleave 16 fma per mfma 2w/SIMD", ncu, 512, 8, 128, 2048);
  run(probe<3, 32>, "interleave 32 fma per mfma", ncu, 256, 8, 256, 2048);
  return 0;
}

// Probe: where do workgroups of a CU-masked stream land?  Prints (XCC, SE, SH, CU) histograms for a few masks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <tuple>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
__global__ void where(unsigned *out) {
  if (threadIdx.x == 0) {
    unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);       // HW_REG_XCC_ID[3:0]
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hw;
  }
  // stay a little so that workgroups spread over the allowed CUs
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}
}
int main() {
  const int nwg = 512;
  unsigned *d; CK(hipMalloc(&d, nwg * 8));
  std::vector<unsigned> h(nwg * 2);
  for (int test = 0; test < 4; ++test) {
    std::vector<uint32_t> mask(8, 0u);
    const char *name = "";
    if (test == 0) { mask[0] = 0xffu; name = "bits 0..7"; }
    if (test == 1) { mask[0] = 0x1u; name = "bit 0"; }
    if (test == 2) { mask[0] = 0x100u; name = "bit 8"; }
    if (test == 3) { for (auto &m : mask) m = 0xffffffffu; mask[0] &= ~0xffu; name = "all but bits 0..7"; }
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, 8, mask.data()));
    CK(hipMemsetAsync(d, 0xff, nwg * 8, s));
    const int launch = test == 3 ? nwg : 64;
    where<<<launch, 64, 0, s>>>(d);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost));
    std::map<std::tuple<int, int, int, int>, int> hist;
    std::map<int, int> perxcc;
    for (int i = 0; i < launch; ++i) {
      const unsigned xcc = h[2 * i] & 15, hw = h[2 * i + 1];
      const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      hist[{(int)xcc, se, sh, cu}]++;
      perxcc[xcc]++;
    }
    printf("mask %s: %zu distinct CUs;", name, hist.size());
    if (test < 3) { for (auto &kv : hist) printf(" (xcc%d se%d sh%d cu%d)x%d", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first), kv.second); }
    else { printf(" per-XCC distinct CUs:"); std::map<int,int> dc; for (auto &kv : hist) dc[std::get<0>(kv.first)]++; for (auto &kv : dc) printf(" xcc%d:%d", kv.first, kv.second); }
    printf("\n");
    CK(hipStreamDestroy(s));
  }
  return 0;
}

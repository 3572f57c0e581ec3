// Which physical CUs does a CU-masked stream reach?  Launches many small workgroups on a masked stream and histograms
// (XCC_ID, SE_ID, CU_ID) read from the hardware registers.  Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/cumask_probe tools/cumask_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <map>
#include <set>
__global__ void probe(unsigned *out) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // busy a little so that blocks spread
  float x = threadIdx.x;
  for (int i = 0; i < 20000; ++i) x = x * 1.0001f + 0.5f;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc + (x == 123.f); }
}
int main(int argc, char **argv) {
  int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  printf("CUs %d\n", ncu);
  const int words = (ncu + 31) / 32, blocks = 4096;
  unsigned *d; hipMalloc(&d, blocks * 8);
  std::vector<unsigned> h(blocks * 2);
  for (int test = 0; test < 4; ++test) {
    std::vector<uint32_t> mask(words, 0);
    const char *name = "";
    if (test == 0) { for (int i = 0; i < ncu; ++i) mask[i / 32] |= 1u << (i % 32); name = "all"; }
    if (test == 1) { for (int i = 0; i < 8; ++i) mask[i / 32] |= 1u << (i % 32); name = "bits 0..7"; }
    if (test == 2) { for (int i = 8; i < ncu; ++i) mask[i / 32] |= 1u << (i % 32); name = "bits 8..N-1"; }
    if (test == 3) { for (int i = 0; i < 32; ++i) mask[i / 32] |= 1u << (i % 32); name = "bits 0..31"; }
    hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, words, mask.data());
    if (e != hipSuccess) { printf("%s: create failed %s\n", name, hipGetErrorString(e)); continue; }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, s, d);
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, s, d);
    hipEventRecord(b, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (int i = 0; i < blocks; ++i) {
      unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
      unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;  // gfx9 HW_ID: CU_ID[11:8], SH_ID[12], SE_ID[15:13]
      per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
    }
    int total = 0; printf("%-12s %.3f ms  distinct CUs per XCC:", name, ms);
    for (auto &kv : per_xcc) { printf(" x%u:%zu", kv.first, kv.second.size()); total += kv.second.size(); }
    printf("  total %d\n", total);
    hipStreamDestroy(s);
  }
  return 0;
}

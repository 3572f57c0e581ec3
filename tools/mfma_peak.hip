// Sustained fp32 MFMA rate of the chip on random operands (no memory traffic in the loop): context for the
// "practical peak" quoted in DESIGN.md next to the nominal 157.3 TF/s.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512, 2) void mfma_loop(float *out, const float *in, int iters) {
  f32x16 acc[8];
  for (int q = 0; q < 8; ++q)
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 8 + i) & 4095]; b[i] = in[(threadIdx.x * 8 + i + 777) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[q], acc[q], 0, 0, 0);
  }
  float s = 0.f;
  for (int q = 0; q < 8; ++q)
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float *in, *out;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 512 * 4);
  float h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop, dim3(256), dim3(512), 0, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * 8 * iters * 64 * 4096.0;  // blocks * waves * iters * mfma/iter * flop/mfma
    printf("fp32 MFMA 32x32x2, 8 waves/CU, random operands: %.1f TF/s (%.1f ms)\n", flop / ms / 1e9, ms);
  }
  return 0;
}

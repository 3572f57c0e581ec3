// How the B-operand feed from LDS limits the fp32 MFMA rate: variants of the gemm_rs inner loop without any
// global traffic.  V=0: operands in registers; V=1: one ds_read_b32 per MFMA (row-major [r][256] image, as
// gemm_rs.hip); V=2: ds_read_b128 per 4 MFMAs ([r][32 lanes][8 tiles] image).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int V>
__global__ __launch_bounds__(512, 2) void loop(float *out, const float *in, int iters) {
  extern __shared__ float Bs[];  // 128 x 256 floats
  for (int i = threadIdx.x; i < 128 * 256; i += 512) Bs[i] = in[i & 4095];
  __syncthreads();
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  f32x16 acc[8];
  for (int q = 0; q < 8; ++q)
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = in[(threadIdx.x * 16 + i) & 4095];
  for (int it = 0; it < iters; ++it) {
    const int kc = it & 3;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float b[8];
      if (V == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) b[q] = a[(j + q) & 15];
      } else if (V == 1) {
        const float *bp = Bs + (kc * 32 + h * 16 + j) * 256 + m;
#pragma unroll
        for (int q = 0; q < 8; ++q) b[q] = bp[q * 32];
      } else {
        const float4 *bp = reinterpret_cast<const float4 *>(Bs + (kc * 32 + h * 16 + j) * 256 + m * 8);
        const float4 u = bp[0], w = bp[1];
        b[0] = u.x; b[1] = u.y; b[2] = u.z; b[3] = u.w; b[4] = w.x; b[5] = w.y; b[6] = w.z; b[7] = w.w;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[q], acc[q], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int q = 0; q < 8; ++q)
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int V>
void run(float *out, float *in, const char *what) {
  const int iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void *>(loop<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(loop<V>, dim3(256), dim3(512), 131072, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flop = 256.0 * 8 * iters * 128 * 4096.0;
  printf("%-40s %.1f TF/s\n", what, flop / best / 1e9);
}
int main() {
  float *in, *out;
  (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&out, 256 * 512 * 4);
  float hbuf[4096];
  for (int i = 0; i < 4096; ++i) hbuf[i] = (float)rand() / RAND_MAX - 0.5f;
  (void)hipMemcpy(in, hbuf, sizeof(hbuf), hipMemcpyHostToDevice);
  run<0>(out, in, "B from registers");
  run<1>(out, in, "B: ds_read_b32 per MFMA ([r][256])");
  run<2>(out, in, "B: 2 x ds_read_b128 per 8 MFMAs");
  return 0;
}

// Can one wave's vector ALU work run under the MFMAs of the OTHER wave on its SIMD?  (The row-streaming GEMM keeps 2 waves
// per SIMD and counts on one wave's epilogue running under its partner's MFMAs.)  One workgroup of 8 waves per CU, waves
// w and w+4 share a SIMD.  mode 0: all 8 waves run an MFMA loop; mode 1: all 8 run a VALU loop (independent FMAs);
// mode 2: waves 0-3 MFMA, waves 4-7 VALU - each SIMD holds one of each; mode 3: every wave alternates MFMA blocks and
// VALU blocks (one wave's own interleave).  Overlap: t(2) ~ max(t_mfma/2-per-wave..., ...) - see the printed ratios.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o gpurun_out/overlap && gpurun_out/overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512, 1) void probe(float *out, const float *in, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float a[4], b[4], v[16];
  for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x * 4 + i) & 4095]; b[i] = in[(threadIdx.x * 4 + i + 777) & 4095]; }
  for (int i = 0; i < 16; ++i) v[i] = in[(threadIdx.x + i * 64) & 4095];
  const bool do_mfma = MODE == 0 || MODE == 3 || (MODE == 2 && wave < 4);
  const bool do_valu = MODE == 1 || MODE == 3 || (MODE == 2 && wave >= 4);
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {   // 16 MFMAs = 1024 cycles of the matrix pipe
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[q], acc[q], 0, 0, 0);
    }
    if (do_valu) {   // 256 independent-chain FMAs = 1024 cycles of the vector ALU
#pragma unroll
      for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
    }
  }
  float s = 0.f;
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static float run(float *out, const float *in, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best;
}

int main() {
  float *in, *out;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 512 * 4);
  float h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int iters = 4000;
  const float t0 = run<0>(out, in, iters), t1 = run<1>(out, in, iters), t2 = run<2>(out, in, iters), t3 = run<3>(out, in, iters);
  printf("mode 0, 8 waves x MFMA blocks            : %.2f ms  (%.1f TF/s)\n", t0, 256.0 * 8 * iters * 16 * 4096.0 / t0 / 1e9);
  printf("mode 1, 8 waves x VALU blocks            : %.2f ms\n", t1);
  printf("mode 2, 4 waves MFMA + 4 waves VALU      : %.2f ms  (half of each: no overlap would be %.2f, full overlap %.2f)\n", t2,
         (t0 + t1) / 2, (t0 > t1 ? t0 : t1) / 2);
  printf("mode 3, every wave MFMA block then VALU  : %.2f ms  (no overlap %.2f, full overlap %.2f)\n", t3, t0 + t1, t0 > t1 ? t0 : t1);
  return 0;
}

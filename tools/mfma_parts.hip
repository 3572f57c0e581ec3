// Which part of gemm_rs's per-tile work costs MFMA time?  A synthetic wave loop shaped like the 1 M x 128 x 256
// forward (4 chunks of 16 steps x 8 MFMAs per 32-row tile, B from a [r][256] LDS image), with parts switched on:
//   bit 0: A chunk loaded from global memory one chunk ahead (4 x 16 B per lane)
//   bit 1: tile epilogue stores D (128 dword stores per lane) and zeroes the accumulators
//   bit 2: BatchNorm statistics in the epilogue (per-lane sums, fp64 accumulate)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int PARTS>
__global__ __launch_bounds__(512, 2) void loop(float *__restrict__ out, const float *__restrict__ in,
                                              const float *__restrict__ A, float *__restrict__ D, long long P, int tiles_per_wave) {
  extern __shared__ float Bs[];  // 128 x 256 floats
  for (int i = threadIdx.x; i < 128 * 256; i += 512) Bs[i] = in[i & 4095];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 31, h = lane >> 5;
  f32x16 acc[8];
  for (int q = 0; q < 8; ++q)
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  double ds[8], dq[8];
  for (int q = 0; q < 8; ++q) { ds[q] = 0; dq[q] = 0; }
  const long long nw = (long long)gridDim.x * 8;
  long long tile = (long long)blockIdx.x * 8 + wave;
  float4 cur[4], nxt[4];
  auto load = [&](float4 (&d)[4], long long tl, int kc) {
    const float *p = A + (tl * 32 + m) * 128 + kc * 32 + h * 16;
    for (int i = 0; i < 4; ++i) d[i] = (PARTS & 1) ? *reinterpret_cast<const float4 *>(p + 4 * i) : make_float4(1.f, 2.f, 3.f, 4.f);
  };
  load(cur, tile, 0);
  for (int t = 0; t < tiles_per_wave; ++t) {
    for (int kc = 0; kc < 4; ++kc) {
      const long long nt = kc == 3 ? tile + nw : tile;
      load(nxt, nt < P / 32 ? nt : tile, (kc + 1) & 3);
      float a[16];
      for (int i = 0; i < 4; ++i) { a[4 * i] = cur[i].x; a[4 * i + 1] = cur[i].y; a[4 * i + 2] = cur[i].z; a[4 * i + 3] = cur[i].w; }
      const float *bp = Bs + (kc * 32 + h * 16) * 256 + m;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float b[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) b[q] = bp[j * 256 + q * 32];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[q], acc[q], 0, 0, 0);
      }
      for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
    }
    if (PARTS & 6) {
      const long long row0 = tile * 32 + 4 * h;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float cs = 0.f, cq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long long row = row0 + (r & 3) + 8 * (r >> 2);
          const float v = acc[q][r];
          if (PARTS & 2) { if (row < P) D[row * 256 + q * 32 + m] = v; }
          if (PARTS & 4) { cs += v; cq += v * v; }
          acc[q][r] = 0.f;
        }
        ds[q] += cs; dq[q] += cq;
      }
    }
    tile += nw;
    if (tile >= P / 32) tile -= nw;
  }
  float s = 0.f;
  for (int q = 0; q < 8; ++q) {
    s += (float)(ds[q] + dq[q]);
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int PARTS>
void run(float *out, float *in, float *A, float *D, long long P, const char *what) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(loop<PARTS>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int tpw = (int)(P / 32 / (256 * 8));
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(loop<PARTS>, dim3(256), dim3(512), 131072, 0, out, in, A, D, P, tpw);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flop = 2.0 * P * 128 * 256;
  printf("%-52s %7.1f us  %6.1f TF/s\n", what, best * 1e3, flop / best / 1e9);
}
int main() {
  const long long P = 1048576;
  float *in, *out, *A, *D;
  (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&A, P * 128 * 4); (void)hipMalloc(&D, P * 256 * 4);
  float hbuf[4096];
  for (int i = 0; i < 4096; ++i) hbuf[i] = (float)rand() / RAND_MAX - 0.5f;
  (void)hipMemcpy(in, hbuf, sizeof(hbuf), hipMemcpyHostToDevice);
  for (long long o = 0; o < P * 128; o += 4096) (void)hipMemcpyAsync(A + o, in, 4096 * 4, hipMemcpyDeviceToDevice, 0);
  (void)hipDeviceSynchronize();
  run<0>(out, in, A, D, P, "MFMA + LDS B only");
  run<1>(out, in, A, D, P, "+ A streamed from HBM");
  run<2>(out, in, A, D, P, "+ D stored");
  run<4>(out, in, A, D, P, "+ statistics");
  run<3>(out, in, A, D, P, "+ A streamed + D stored");
  run<7>(out, in, A, D, P, "+ A streamed + D stored + statistics (= gemm_rs fwd)");
  return 0;
}

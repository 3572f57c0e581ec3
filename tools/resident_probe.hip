// What kind of kernel, resident on a second queue, slows the main queue down?  (MI355X; build: hipcc --offload-arch=gfx950
// -O3 tools/resident_probe.hip -o /tmp/resident_probe)   Main queue: 40 passes y += 1 over 256 MB.  Side queue: ONE
// workgroup running ~2 ms of (a) pure ALU, (b) ALU + s_barrier, (c) LDS traffic + s_barrier, (d) a dependent global load
// per iteration (cached), (e) the same with 1024 threads, (f) s_sleep only.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void add_one(float *x, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n / 4; i += stride) {
    float4 v = reinterpret_cast<float4 *>(x)[i];
    v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
    reinterpret_cast<float4 *>(x)[i] = v;
  }
}

template <int MODE>
__global__ void resident(const int *src, int *out, int iters) {
  __shared__ float sh[4096];
  int acc = threadIdx.x;
  float f = (float)threadIdx.x;
  int idx = threadIdx.x & 1023;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0 || MODE == 1) {
#pragma unroll
      for (int k = 0; k < 64; ++k) f = f * 1.0001f + 0.5f;
      if (MODE == 1) __syncthreads();
    } else if (MODE == 2) {
      sh[(threadIdx.x + i) & 4095] = f;
      __syncthreads();
      f += sh[(threadIdx.x * 7 + i) & 4095];
      __syncthreads();
    } else if (MODE == 3) {
      idx = src[idx & 1023];          // dependent load: one round trip per iteration
      acc += idx;
    } else if (MODE == 4) {
      __builtin_amdgcn_s_sleep(100);
    } else if (MODE == 6 || MODE == 7) {
#pragma unroll
      for (int k = 0; k < 64; ++k) f = f * 1.0001f + 0.5f;
      asm volatile("v_mov_b32 v127, 0" ::: "v127");
      if (MODE == 6 || (i & 7) == 0) __builtin_amdgcn_s_sleep(1);   // a yield point without a barrier
    } else if (MODE >= 14 && MODE <= 16) {   // like the sampling loop: uniform global load + ALU burst (+ a store by thread 0) + barrier
      const int o = src[idx & 1023];                       // every wave loads the same "winner" (an L2 hit)
      if (MODE != 16 || ((threadIdx.x >> 6) + i) % 4 == 0) {
#pragma unroll
        for (int k = 0; k < 64; ++k) f = f * 1.0001f + (float)o;
      }
      asm volatile("v_mov_b32 v127, 0" ::: "v127");
      if (MODE >= 15 && threadIdx.x == 0) out[1 + (i & 1023)] = o;
      idx = o;
      __syncthreads();
    } else if (MODE == 12 || MODE == 13) {   // unbalanced like the sampling kernel: few waves work, the rest wait at the barrier
      if ((threadIdx.x >> 6) < (MODE == 12 ? 1 : 4)) {
#pragma unroll
        for (int k = 0; k < 64; ++k) f = f * 1.0001f + 0.5f;
      }
      asm volatile("v_mov_b32 v127, 0" ::: "v127");
      __syncthreads();
    } else if (MODE >= 8) {   // 8: a barrier per 64 FMAs; 9: per 256; 10: per 1024; 11: per 4096
#pragma unroll
      for (int k = 0; k < 64; ++k) f = f * 1.0001f + 0.5f;
      asm volatile("v_mov_b32 v127, 0" ::: "v127");
      const int every = MODE == 8 ? 1 : MODE == 9 ? 4 : MODE == 10 ? 16 : 64;
      if (i % every == 0) __syncthreads();
    } else if (MODE == 5) {
#pragma unroll
      for (int k = 0; k < 64; ++k) f = f * 1.0001f + 0.5f;
      asm volatile("v_mov_b32 v127, 0" ::: "v127");   // allocate 128 VGPRs per wave: 16 waves fill the CU's register file
    }
  }
  if (acc == -12345 || f == -1.2345f) out[0] = acc + (int)f + idx;
}

static double ms_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main() {
  const size_t n = 64u << 20;
  float *x; int *src, *out;
  hipMalloc(&x, n * 4); hipMemset(x, 0, n * 4);
  hipMalloc(&src, 1024 * 4); hipMalloc(&out, 4 * 2048);
  std::vector<int> h(1024); for (int i = 0; i < 1024; ++i) h[i] = (i * 37 + 11) & 1023;
  hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
  hipStream_t mainq, side; hipStreamCreate(&mainq); hipStreamCreate(&side);
  hipEvent_t ev; hipEventCreate(&ev);
  auto work = [&]() { for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(add_one, dim3(4096), dim3(256), 0, mainq, x, n); };
  auto time_it = [&](auto fn) {
    for (int i = 0; i < 2; ++i) fn();
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 10; ++i) fn();
    hipDeviceSynchronize();
    return ms_since(t0) / 10;
  };
  const double base = time_it(work);
  printf("main queue alone: %.3f ms\n", base);
  struct Case { const char *name; int mode; int threads; int iters; };
  const Case cases[] = {{"pure ALU, 64 threads", 0, 64, 20000}, {"pure ALU, 1024 threads", 0, 1024, 4500}, {"pure ALU, 4 x 1024 threads", 0, 1024, 4500},
                        {"ALU + s_barrier, 1024 threads", 1, 1024, 4500}, {"LDS + 2 s_barrier, 1024 threads", 2, 1024, 20000},
                        {"dependent cached global load, 64 threads", 3, 64, 6000}, {"dependent cached global load, 1024 threads", 3, 1024, 6000},
                        {"s_sleep, 64 threads", 4, 64, 1500},
                        {"pure ALU, 1024 threads, 128 VGPRs each", 5, 1024, 4500}, {"pure ALU, 512 threads, 128 VGPRs each", 5, 512, 9000},
                        {"ALU + s_sleep 1 per 64 FMA, 1024 thr, 128 VGPRs", 6, 1024, 3500}, {"ALU + s_sleep 1 per 512 FMA, 1024 thr, 128 VGPRs", 7, 1024, 4300},
                        {"ALU + s_barrier per 64 FMA, 1024 thr, 128 VGPRs", 8, 1024, 4000}, {"ALU + s_barrier per 256 FMA, 1024 thr, 128 VGPRs", 9, 1024, 4300},
                        {"ALU + s_barrier per 1024 FMA, 1024 thr, 128 VGPRs", 10, 1024, 4400}, {"ALU + s_barrier per 4096 FMA, 1024 thr, 128 VGPRs", 11, 1024, 4450},
                        {"1 of 16 waves works, barrier per 64 FMA, 128 VGPRs", 12, 1024, 4400}, {"4 of 16 waves work, barrier per 64 FMA, 128 VGPRs", 13, 1024, 4400},
                        {"load + 64 FMA + barrier per iteration", 14, 1024, 2500}, {"... + store by thread 0, 80 KB LDS", 15, 1024, 2500},
                        {"... and only a quarter of the waves working", 16, 1024, 4000}};
  for (const Case &c : cases) {
    const int nb = (c.name[10] == '4' && c.name[12] == 'x') ? 4 : 1;
    auto side_k = [&]() {
      if (nb == 4) { hipLaunchKernelGGL(resident<0>, dim3(4), dim3(c.threads), 0, side, src, out, c.iters); return; }
      switch (c.mode) {
        case 0: hipLaunchKernelGGL(resident<0>, dim3(1), dim3(c.threads), 0, side, src, out, c.iters); break;
        case 1: hipLaunchKernelGGL(resident<1>, dim3(1), dim3(c.threads), 0, side, src, out, c.iters); break;
        case 2: hipLaunchKernelGGL(resident<2>, dim3(1), dim3(c.threads), 0, side, src, out, c.iters); break;
        case 3: hipLaunchKernelGGL(resident<3>, dim3(1), dim3(c.threads), 0, side, src, out, c.iters); break;
        case 6: hipLaunchKernelGGL(resident<6>, dim3(1), dim3(c.threads), 0, side, src, out, c.iters); break;
        case 7: hipLaunchKernelGGL(resident<7>, dim3(1), dim3(c.threads), 0, side, src, out, c.iters); break;
        case 8: hipLaunchKernelGGL(resident<8>, dim3(1), dim3(c.threads), 0, side, src, out, c.iters); break;
        case 14: hipLaunchKernelGGL(resident<14>, dim3(1), dim3(c.threads), 0, side, src, out, c.iters); break;
        case 15: hipLaunchKernelGGL(resident<15>, dim3(1), dim3(c.threads), 80 * 1024, side, src, out, c.iters); break;
        case 16: hipLaunchKernelGGL(resident<16>, dim3(1), dim3(c.threads), 80 * 1024, side, src, out, c.iters); break;
        case 12: hipLaunchKernelGGL(resident<12>, dim3(1), dim3(c.threads), 0, side, src, out, c.iters); break;
        case 13: hipLaunchKernelGGL(resident<13>, dim3(1), dim3(c.threads), 0, side, src, out, c.iters); break;
        case 9: hipLaunchKernelGGL(resident<9>, dim3(1), dim3(c.threads), 0, side, src, out, c.iters); break;
        case 10: hipLaunchKernelGGL(resident<10>, dim3(1), dim3(c.threads), 0, side, src, out, c.iters); break;
        case 11: hipLaunchKernelGGL(resident<11>, dim3(1), dim3(c.threads), 0, side, src, out, c.iters); break;
        case 5: hipLaunchKernelGGL(resident<5>, dim3(1), dim3(c.threads), 0, side, src, out, c.iters); break;
        default: hipLaunchKernelGGL(resident<4>, dim3(1), dim3(c.threads), 0, side, src, out, c.iters); break;
      }
    };
    const double alone = time_it(side_k);
    const double both = time_it([&]() { side_k(); work(); hipEventRecord(ev, side); hipStreamWaitEvent(mainq, ev, 0); });
    printf("%-44s resident alone %.3f ms; main queue beside it %.3f ms (+%.2f)\n", c.name, alone, both, both - base);
  }
  return 0;
}

// Backward of the LAST layer of a crop stack (conv K -> C, BatchNorm, ReLU, max over each crop's members; reference
// modules.py:104-124, pytorch_utils.py:61-113) whose output Y = X~ W^T was never stored (gb_gemm_fwd_pool), as
// "low rank + sparse".  X~ = relu(a2*Y2 + b2) are the (distinct) rows of the previous layer, w_u their multiplicity,
// P the rows of the full batch.  Behind the max-pool the gradient reaching the layer's BatchNorm is nonzero in ONE
// member row per (crop, channel): with g = dout*[out > 0] at row arg,
//     dY[u,c] = a_c g[u,c] - w_u f_c - w_u e_c y[u,c],     e_c = a_c dgamma_c rstd_c / P,  f_c = a_c dbeta_c / P - e_c mean_c,
//     dbeta_c = sum g,  dgamma_c = sum g (y* - mean_c) rstd_c   (y* = y at the arg row: gb_pool_pairs' ystar),
// hence, with y = X~ W^T,
//     dX~ = S - w (v + X~ M),   S[u,:] = sum_c a_c g[u,c] W[c,:]  (sparse),  v = W^T f,  M = W^T diag(e) W   (K x K),
//     dW  = T - f sx^T - diag(e) W G,   T[c,:] = a_c sum_u g[u,c] X~[u,:]  (sparse),  sx = sum_u w_u X~[u,:],
//                                       G = sum_u w_u X~[u,:]^T X~[u,:]   (K x K weighted Gram matrix).
// The dense P x C gradient is never formed: a K -> K product (gb_crop_bwd_dense, half the FLOP of the C -> K dgrad) and a
// K x K Gram product (gb_gemm_gram, half the FLOP of the C x K wgrad) replace the two dense products, the BatchNorm
// backward pass over P x C values and its column-sum pass; this file holds the sparse part and the small algebra.
// Cancellation: G - sx sx^T / P is the covariance of post-ReLU activations (mean^2 / E[x^2] ~ 1/3): one bit, not many.
#include "gb_common.h"

namespace gb {

constexpr int CB_TPB = 512;
constexpr int CB_ROWS = 128;  // rows of a seed staged per pass (a seed has <= 256 distinct rows)

// One persistent workgroup per CU, looping over seeds.  Thread (k = t % K, cq = t / K) owns column k of the staged rows
// and the channels c = 4 ci + cq (ci < C/4): its slice of T (C/4 = 64 accumulators) stays in registers for the whole
// launch; its slice of W is re-read from L2 eight channels at a time (kept in registers too, the allocator spilled
// everything else).  K = 128, C = 256: 512 threads.
// MEASURED (tools/crop_probe.py, 0.4 M rows, 4096 seeds): 3.4 ms per launch, 2.9 ms of it the ds_add_f32 of the sparse
// dX~ rows (700 M lane-atomics = 164 clocks per wave instruction), 0.3 ms the T loop, 0.23 ms everything else - against
// 0.25 ms for the two dense kernels it replaces.  The formulation is right (1e-6 of the dense backward), this kernel is
// not: it needs the dX~ part as a row-gather over per-row entry lists (no float atomics) before the path can be default.
// Everything is latency-bound with one workgroup (8 waves) per CU, so every phase keeps many independent memory
// operations in flight: the seed's (gradient, arg row) entries are read with all four crops unrolled, the rows are staged
// 8 at a time as 16-byte loads, and the inner loop is branch-free (an entry outside the staged rows adds 0 to row 0)
// with the four crops of a channel read as one 16-byte LDS word each.
template <int K, int C>
__global__ __launch_bounds__(CB_TPB, 1) void crop_bwd_sparse_kernel(
    const float *__restrict__ dout, const float *__restrict__ out, const int32_t *__restrict__ arg,
    const float *__restrict__ ystar, const float *__restrict__ ab, const float *__restrict__ y2,
    const float *__restrict__ ab2, const float *__restrict__ w3, const float *__restrict__ row_w,
    const int64_t *__restrict__ off, const int32_t *__restrict__ cnt, long long R, int D, float *__restrict__ sdx,
    float *__restrict__ tpart, double *__restrict__ rpart) {
  // tpart [gridDim.x][C][K], rpart [gridDim.x][2C + K]: per-workgroup partial sums, written with plain stores (every
  // workgroup adding its 32 k values of T into ONE array was 8 M same-address atomics: 3 ms of a 3.3 ms launch)
  static_assert(K == 128 && K * 4 == CB_TPB && C % 4 == 0 && C <= CB_TPB, "thread layout");
  constexpr int CI = C / 4;
  constexpr int RG = CB_TPB / (K / 4);      // 16 row groups of 32 threads, each thread 4 consecutive k (staging layout)
  constexpr int RPT = CB_ROWS / RG;         // 8 rows per thread and pass
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // (g, arg) first: their 2 x 64 unrolled reads then differ by an immediate offset below 64 KB from one base register
  // (placed behind the 128 KB of rows, every read needed its own address register: 128 of them, all spilled)
  float *ga = lds;                          // [C][4]         a_c * g of the seed's (channel, crop) entries
  int *ai = reinterpret_cast<int *>(ga + 4 * C);  // [C][4]   their arg rows, relative to the seed's first row
  float *xs = lds + 8 * C;                  // [CB_ROWS][K]   staged X~ rows
  float *dxs = xs + CB_ROWS * K;            // [CB_ROWS][K]   sparse dX~ rows
  const int t = threadIdx.x, k = t % K, cq = t / K;
  const int k4 = (t % (K / 4)) * 4, rg = t / (K / 4);
  float tacc[CI];
#pragma unroll
  for (int ci = 0; ci < CI; ++ci) tacc[ci] = 0.f;
  float a2[4], b2[4], sxacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) { a2[e] = ab2[k4 + e]; b2[e] = ab2[K + k4 + e]; }
  double sb = 0.0, sgm = 0.0;               // threads t < C: dbeta / dgamma partial sums of channel t
  float ca = 0.f, cmean = 0.f, crstd = 0.f;
  if (t < C) { ca = ab[t]; cmean = ab[2 * C + t]; crstd = ab[3 * C + t]; }
  for (long long r = blockIdx.x; r < R; r += gridDim.x) {
    const long long u0 = off[r];
    const int n = cnt[r];
    if (t < C) {
      float vo[4], vd[4], vy[4];
      int va[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {          // D <= 4: all crops' loads in flight together
        const size_t at = (size_t)(r * D + (d < D ? d : 0)) * C + t;
        vo[d] = out[at]; vd[d] = dout[at]; vy[d] = ystar[at]; va[d] = arg[at];
      }
      float fb = 0.f, fg = 0.f, gq[4];
      int iq[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const float g = (d < D && vo[d] > 0.f) ? vd[d] : 0.f;
        gq[d] = ca * g;
        iq[d] = va[d] - (int)u0;
        fb += g;
        fg += g * ((vy[d] - cmean) * crstd);
      }
      *reinterpret_cast<float4 *>(ga + 4 * t) = make_float4(gq[0], gq[1], gq[2], gq[3]);
      *reinterpret_cast<int4 *>(ai + 4 * t) = make_int4(iq[0], iq[1], iq[2], iq[3]);
      sb += (double)fb;
      sgm += (double)fg;
    }
    for (int base = 0; base < n; base += CB_ROWS) {
      const int rn = n - base < CB_ROWS ? n - base : CB_ROWS;
      const unsigned rowbase = (unsigned)(u0 + base);   // 32-bit row arithmetic: P < 2^31 / K (host-checked)
#pragma unroll
      for (int half = 0; half < 2; ++half) {  // two batches of 4 rows: 4 independent 16-byte loads in flight per thread
        float4 yv[RPT / 2];
        float wv[RPT / 2];
#pragma unroll
        for (int j = 0; j < RPT / 2; ++j) {   // clamped addresses: all loads of a batch are issued before any is used
          const int i = rg + RG * (half * (RPT / 2) + j);
          const unsigned u = rowbase + (unsigned)(i < rn ? i : rn - 1);
          yv[j] = *reinterpret_cast<const float4 *>(y2 + (size_t)u * K + k4);
          wv[j] = row_w[u];
        }
#pragma unroll
        for (int j = 0; j < RPT / 2; ++j) {
          const int i = rg + RG * (half * (RPT / 2) + j);
          if (i < rn) {
            const float q[4] = {yv[j].x, yv[j].y, yv[j].z, yv[j].w};
            float x[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float z = a2[e] * q[e] + b2[e];
              x[e] = z > 0.f ? z : 0.f;
              sxacc[e] += wv[j] * x[e];
            }
            *reinterpret_cast<float4 *>(xs + i * K + k4) = make_float4(x[0], x[1], x[2], x[3]);
            *reinterpret_cast<float4 *>(dxs + i * K + k4) = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      }
      __syncthreads();
      // an offset the compiler cannot see through: (g, arg) do not change between the passes of a seed, and without it
      // all 2 x 64 16-byte reads are hoisted out of the pass loop - 512 live registers, 700 spilled
      int opaque = 0;
      asm volatile("" : "+v"(opaque));
#pragma unroll
      for (int cg = 0; cg < CI / 8; ++cg) {  // 8 channels at a time: their W values are requested together (L2 hits)
        float w8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w8[j] = w3[(size_t)(4 * (8 * cg + j) + cq) * K + k];
#pragma unroll
        for (int j = 0; j < 8; ++j) {        // a wave shares cq, hence the channel: LDS broadcast reads
          const int ci = 8 * cg + j;
          const float4 g4 = *reinterpret_cast<const float4 *>(ga + opaque + 4 * (4 * ci + cq));
          const int4 i4 = *reinterpret_cast<const int4 *>(ai + opaque + 4 * (4 * ci + cq));
          const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
          const int iv[4] = {i4.x - base, i4.y - base, i4.z - base, i4.w - base};
          float xv[4], gg[4];
          int ii[4];
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const bool ok = iv[d] >= 0 && iv[d] < rn;
            ii[d] = ok ? iv[d] : 0;
            gg[d] = ok ? gv[d] : 0.f;
            xv[d] = xs[ii[d] * K + k];
          }
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            tacc[ci] += gg[d] * xv[d];
            atomicAdd(&dxs[ii[d] * K + k], gg[d] * w8[j]);   // the four cq groups may meet in one row: LDS float atomic
          }
          // pin the accumulation here: left alone, the compiler defers all 256 products of a pass to its end and
          // keeps their 512 operands live (600 spilled registers)
          asm volatile("" : "+v"(tacc[ci]) : : "memory");
          if (j & 1) __builtin_amdgcn_sched_barrier(0);   // two channels' LDS traffic in flight at a time
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the groups apart: hoisting all 64 channels' reads spills hundreds of registers
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
        const int i = rg + RG * j;
        if (i < rn)
          *reinterpret_cast<float4 *>(sdx + (u0 + base + i) * K + k4) = *reinterpret_cast<const float4 *>(dxs + i * K + k4);
      }
      __syncthreads();
    }
  }
  float *tp = tpart + (size_t)blockIdx.x * C * K;
#pragma unroll
  for (int ci = 0; ci < CI; ++ci) tp[(size_t)(4 * ci + cq) * K + k] = tacc[ci];
  double *rp = rpart + (size_t)blockIdx.x * (2 * C + K);
  if (t < C) {
    rp[t] = sb;
    rp[C + t] = sgm;
  }
  __syncthreads();                           // the row buffers are free: fold the 16 row groups' sx partials through them
  float *fold = xs;                          // [RG][K]
#pragma unroll
  for (int e = 0; e < 4; ++e) fold[rg * K + k4 + e] = sxacc[e];
  __syncthreads();
  if (t < K) {
    double v = 0.0;
    for (int j = 0; j < RG; ++j) v += (double)fold[j * K + t];
    rp[2 * C + t] = v;
  }
}

// red [2C] and sx [K] = the workgroups' partials (rpart [nb][2C + K]) summed in workgroup order
__global__ __launch_bounds__(256) void crop_bwd_reduce_kernel(const double *__restrict__ rpart, int nb, int n,
                                                              double *__restrict__ red) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double v = 0.0;
  for (int b = 0; b < nb; ++b) v += rpart[(size_t)b * n + i];
  red[i] = v;
}

// e, f (C), v (K), M (K,K) from the BatchNorm-backward sums; one workgroup per row j of M, thread k.
__global__ __launch_bounds__(256) void crop_bwd_coef_kernel(const double *__restrict__ red, const float *__restrict__ ab,
                                                            const float *__restrict__ w3, int K, int C, double invP,
                                                            int training, float *__restrict__ ef,
                                                            float *__restrict__ vvec, float *__restrict__ mmat,
                                                            float *__restrict__ dbeta, float *__restrict__ dgamma) {
  extern __shared__ float s_e[];  // [C] e, then [C] f
  float *s_f = s_e + C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const double db = red[c], dg = red[C + c];
    const float a = ab[c], mean = ab[2 * C + c], rstd = ab[3 * C + c];
    const float e = training ? (float)((double)a * dg * (double)rstd * invP) : 0.f;
    const float f = training ? (float)((double)a * db * invP) - e * mean : 0.f;
    s_e[c] = e;
    s_f[c] = f;
    if (blockIdx.x == 0) {
      ef[c] = e;
      ef[C + c] = f;
      dbeta[c] = (float)db;
      dgamma[c] = (float)dg;
    }
  }
  __syncthreads();
  const int j = blockIdx.x;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float m = 0.f, v = 0.f;
    for (int c = 0; c < C; ++c) {
      const float wk = w3[(size_t)c * K + k];
      m += (w3[(size_t)c * K + j] * s_e[c]) * wk;
      v += s_f[c] * wk;
    }
    mmat[(size_t)j * K + k] = m;
    if (j == 0) vvec[k] = v;
  }
}

// dW[c,k] = T[c,k] - f_c sx_k - e_c sum_j W[c,j] G[j,k]; one workgroup per channel c, thread k.
__global__ __launch_bounds__(256) void crop_bwd_dw_kernel(const float *__restrict__ tpart, int nb,
                                                          const float *__restrict__ ef, const double *__restrict__ sx,
                                                          const float *__restrict__ w3, const float *__restrict__ gmat,
                                                          int K, int C, float *__restrict__ dw) {
  const int c = blockIdx.x;
  const float e = ef[c], f = ef[C + c];
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float s = 0.f;
    for (int j = 0; j < K; ++j) s += w3[(size_t)c * K + j] * gmat[(size_t)j * K + k];
    float tsum = 0.f;                        // T[c,k]: the sparse kernel's per-workgroup partials, in workgroup order
    for (int b = 0; b < nb; ++b) tsum += tpart[((size_t)b * C + c) * K + k];
    dw[(size_t)c * K + k] = tsum - f * (float)sx[k] - e * s;
  }
}

}  // namespace gb

using namespace gb;

extern "C" int gb_crop_bwd_ok(int K, int C, int D) { return K == 128 && C == 256 && D >= 1 && D <= 4; }

// workgroups gb_crop_bwd_sparse launches on the current device for R seeds = the leading dimension of its partial buffers
extern "C" int gb_crop_bwd_blocks(long long R) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  return (int)(R < cus ? (R < 1 ? 1 : R) : cus);
}

extern "C" int gb_crop_bwd_sparse(const float *dout, const float *out, const int32_t *arg, const float *ystar,
                                  const float *ab, const float *y2, const float *ab2, const float *w3,
                                  const float *row_w, const int64_t *off, const int32_t *cnt, long long R, int D, int K,
                                  int C, float *sdx, float *tpart, double *rpart, int nb, double *red, void *stream) {
  if (R < 0 || !gb_crop_bwd_ok(K, C, D) || !dout || !out || !arg || !ystar || !ab || !y2 || !ab2 || !w3 || !row_w ||
      !off || !cnt || !sdx || !tpart || !rpart || !red || nb != gb_crop_bwd_blocks(R))
    return GB_EINVAL;
  if (R == 0) return GB_OK;
  static std::atomic<unsigned long long> attr{0};
  auto kern = crop_bwd_sparse_kernel<128, 256>;
  const int lds_bytes = (2 * CB_ROWS * 128 + 2 * 4 * 256) * (int)sizeof(float);
  allow_dynamic_lds(kern, lds_bytes, attr);
  hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(CB_TPB), lds_bytes, as_stream(stream), dout, out, arg, ystar, ab, y2,
                     ab2, w3, row_w, off, cnt, R, D, sdx, tpart, rpart);
  int rc = check_launch("gb_crop_bwd_sparse");
  if (rc != GB_OK) return rc;
  const int n = 2 * C + K;
  hipLaunchKernelGGL(crop_bwd_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), rpart, nb,
                     n, red);
  return check_launch("gb_crop_bwd_sparse (reduce)");
}

extern "C" int gb_crop_bwd_coef(const double *red, const float *ab, const float *w3, int K, int C, long long P_total,
                                int training, float *ef, float *vvec, float *mmat, float *dbeta, float *dgamma,
                                void *stream) {
  if (K < 1 || C < 1 || C > 4096 || P_total < 1 || !red || !ab || !w3 || !ef || !vvec || !mmat || !dbeta || !dgamma)
    return GB_EINVAL;
  hipLaunchKernelGGL(crop_bwd_coef_kernel, dim3((unsigned)K), dim3(256), 2 * C * sizeof(float), as_stream(stream), red, ab,
                     w3, K, C, 1.0 / (double)P_total, training, ef, vvec, mmat, dbeta, dgamma);
  return check_launch("gb_crop_bwd_coef");
}

extern "C" int gb_crop_bwd_dw(const float *tpart, int nb, const float *ef, const double *sx, const float *w3,
                              const float *gmat, int K, int C, float *dw, void *stream) {
  if (K < 1 || C < 1 || nb < 1 || !tpart || !ef || !sx || !w3 || !gmat || !dw) return GB_EINVAL;
  hipLaunchKernelGGL(crop_bwd_dw_kernel, dim3((unsigned)C), dim3(256), 0, as_stream(stream), tpart, nb, ef, sx, w3, gmat, K,
                     C, dw);
  return check_launch("gb_crop_bwd_dw");
}

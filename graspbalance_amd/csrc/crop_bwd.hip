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
// and the channels c = 4 ci + cq (ci < C/4): its slice of W (C/4 registers) and of T (C/4 accumulators) stay in
// registers for the whole launch.  K = 128, C = 256: 512 threads, 64 + 64 such registers.
template <int K, int C>
__global__ __launch_bounds__(CB_TPB, 1) void crop_bwd_sparse_kernel(
    const float *__restrict__ dout, const float *__restrict__ out, const int32_t *__restrict__ arg,
    const float *__restrict__ ystar, const float *__restrict__ ab, const float *__restrict__ y2,
    const float *__restrict__ ab2, const float *__restrict__ w3, const float *__restrict__ row_w,
    const int64_t *__restrict__ off, const int32_t *__restrict__ cnt, long long R, int D, float *__restrict__ sdx,
    float *__restrict__ tmat, double *__restrict__ red, double *__restrict__ sx) {
  static_assert(K * 4 == CB_TPB && C % 4 == 0 && C <= CB_TPB, "thread layout");
  constexpr int CI = C / 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *xs = lds;                          // [CB_ROWS][K]   staged X~ rows
  float *dxs = xs + CB_ROWS * K;            // [CB_ROWS][K]   sparse dX~ rows
  float *ga = dxs + CB_ROWS * K;            // [4][C]         a_c * g of the seed's (crop, channel) entries
  int *ai = reinterpret_cast<int *>(ga + 4 * C);  // [4][C]   their arg rows, relative to the seed's first row
  const int t = threadIdx.x, k = t % K, cq = t / K;
  float wreg[CI], tacc[CI];
#pragma unroll
  for (int ci = 0; ci < CI; ++ci) {
    wreg[ci] = w3[(size_t)(4 * ci + cq) * K + k];
    tacc[ci] = 0.f;
  }
  const float a2 = ab2[k], b2 = ab2[K + k];
  float sxacc = 0.f;
  double sb = 0.0, sgm = 0.0;               // threads t < C: dbeta / dgamma partial sums of channel t
  float ca = 0.f, cmean = 0.f, crstd = 0.f;
  if (t < C) { ca = ab[t]; cmean = ab[2 * C + t]; crstd = ab[3 * C + t]; }
  for (long long r = blockIdx.x; r < R; r += gridDim.x) {
    const long long u0 = off[r];
    const int n = cnt[r];
    if (t < C) {
      float fb = 0.f, fg = 0.f;
      for (int d = 0; d < D; ++d) {
        const size_t at = (size_t)(r * D + d) * C + t;
        const float g = out[at] > 0.f ? dout[at] : 0.f;
        ga[d * C + t] = ca * g;
        ai[d * C + t] = arg[at] - (int)u0;
        fb += g;
        fg += g * ((ystar[at] - cmean) * crstd);
      }
      sb += (double)fb;
      sgm += (double)fg;
    }
    for (int base = 0; base < n; base += CB_ROWS) {
      const int rn = n - base < CB_ROWS ? n - base : CB_ROWS;
      for (int i = cq; i < rn; i += 4) {
        const long long u = u0 + base + i;
        const float z = a2 * y2[u * K + k] + b2;
        const float x = z > 0.f ? z : 0.f;
        xs[i * K + k] = x;
        dxs[i * K + k] = 0.f;
        sxacc += row_w[u] * x;
      }
      __syncthreads();
#pragma unroll
      for (int ci = 0; ci < CI; ++ci) {
        for (int d = 0; d < D; ++d) {  // a wave shares cq, hence the channel: uniform branch, LDS broadcast reads
          const int e = d * C + 4 * ci + cq;
          const float gv = ga[e];
          const int i = ai[e] - base;
          if (gv != 0.f && i >= 0 && i < rn) {
            tacc[ci] += gv * xs[i * K + k];
            atomicAdd(&dxs[i * K + k], gv * wreg[ci]);   // the four cq groups may meet in one row: LDS float atomic
          }
        }
      }
      __syncthreads();
      for (int i = cq; i < rn; i += 4) sdx[(u0 + base + i) * K + k] = dxs[i * K + k];
      __syncthreads();
    }
  }
#pragma unroll
  for (int ci = 0; ci < CI; ++ci)
    if (tacc[ci] != 0.f) atomicAdd(tmat + (size_t)(4 * ci + cq) * K + k, tacc[ci]);
  atomicAdd(sx + k, (double)sxacc);
  if (t < C) {
    atomicAdd(red + t, sb);
    atomicAdd(red + C + t, sgm);
  }
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
__global__ __launch_bounds__(256) void crop_bwd_dw_kernel(const float *__restrict__ tmat, const float *__restrict__ ef,
                                                          const double *__restrict__ sx, const float *__restrict__ w3,
                                                          const float *__restrict__ gmat, int K, int C,
                                                          float *__restrict__ dw) {
  const int c = blockIdx.x;
  const float e = ef[c], f = ef[C + c];
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float s = 0.f;
    for (int j = 0; j < K; ++j) s += w3[(size_t)c * K + j] * gmat[(size_t)j * K + k];
    dw[(size_t)c * K + k] = tmat[(size_t)c * K + k] - f * (float)sx[k] - e * s;
  }
}

}  // namespace gb

using namespace gb;

extern "C" int gb_crop_bwd_ok(int K, int C, int D) { return K == 128 && C == 256 && D >= 1 && D <= 4; }

extern "C" int gb_crop_bwd_sparse(const float *dout, const float *out, const int32_t *arg, const float *ystar,
                                  const float *ab, const float *y2, const float *ab2, const float *w3,
                                  const float *row_w, const int64_t *off, const int32_t *cnt, long long R, int D, int K,
                                  int C, float *sdx, float *tmat, double *red, double *sx, void *stream) {
  if (R < 0 || !gb_crop_bwd_ok(K, C, D) || !dout || !out || !arg || !ystar || !ab || !y2 || !ab2 || !w3 || !row_w ||
      !off || !cnt || !sdx || !tmat || !red || !sx)
    return GB_EINVAL;
  if (R == 0) return GB_OK;
  static std::atomic<unsigned long long> attr{0};
  auto kern = crop_bwd_sparse_kernel<128, 256>;
  const int lds_bytes = (2 * CB_ROWS * 128 + 2 * 4 * 256) * (int)sizeof(float);
  allow_dynamic_lds(kern, lds_bytes, attr);
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  const long long blocks = R < cus ? R : cus;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(CB_TPB), lds_bytes, as_stream(stream), dout, out, arg, ystar, ab, y2,
                     ab2, w3, row_w, off, cnt, R, D, sdx, tmat, red, sx);
  return check_launch("gb_crop_bwd_sparse");
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

extern "C" int gb_crop_bwd_dw(const float *tmat, const float *ef, const double *sx, const float *w3, const float *gmat,
                              int K, int C, float *dw, void *stream) {
  if (K < 1 || C < 1 || !tmat || !ef || !sx || !w3 || !gmat || !dw) return GB_EINVAL;
  hipLaunchKernelGGL(crop_bwd_dw_kernel, dim3((unsigned)C), dim3(256), 0, as_stream(stream), tmat, ef, sx, w3, gmat, K, C,
                     dw);
  return check_launch("gb_crop_bwd_dw");
}

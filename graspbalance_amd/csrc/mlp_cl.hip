// Channel-last fused pieces of the set-abstraction MLPs (SharedMLP = 1x1 conv + BatchNorm + ReLU,
// PointNet/pytorch_utils.py:5-182, then max over the nsample axis, pointnet2_modules.py:165-169).
//
// The reference materialises (B,C,m,ns) tensors channel-major and makes ~10 full passes over them per
// layer (cat, conv in NCHW with layout transposes, BN statistics, BN apply, ReLU, max_pool2d, and the
// mirror passes in backward).  Here activations are position-major rows  act[p][c],
// p = (b*m + j)*ns + k,  so a row is one contiguous C-vector: the gather of a neighbour's features is
// one coalesced row copy, BN statistics are column sums, affine+ReLU(+max over ns) is one pass, and
// the feature gradient scatter is a row-contiguous atomic (full-rate on MI355X).
//
//   gb_group_concat_cl      X0[p] = [ (xyz[idx[p]] - centre) (* 1/radius | rotated by R_j) , feat[idx[p]] ]
//   gb_group_concat_cl_grad dfeat[b, idx[p], :] += dX0[p, 3:]
//   gb_col_stats            sum[c] += sum_p y, sumsq[c] += sum_p y^2          (fp64 accumulators)
//   gb_bn_finalize          a = gamma*rstd, b = beta - mean*a, running stats update
//   gb_affine_act           z = act(a*y + b (+ residual))
//   gb_affine_relu_maxpool  out[r,c] = max_k relu(a*y[r*ns+k,c] + b), argmax k
//   gb_bn_bwd_stats(_pool)  dbeta = sum dA, dgamma = sum dA*xhat
//   gb_bn_bwd_apply(_pool)  dy = a*(dA - dbeta/P - xhat*dgamma/P)
#include "gb_common.h"

namespace gb {

constexpr int CL_TPB = 256;

// ------------------------------------------------------------------------------------------------
// group + concat, channel-last.  mode 0: centred ; 1: centred * scale (normalize_xyz: on the GPU torch
// evaluates `grouped_xyz /= radius` as a multiplication by the fp32 reciprocal, so scale = 1.0f/radius
// keeps this bit-identical to the unfused path) ; 2: centred then rotated p^T R (CylinderQueryAndGroup).
__global__ __launch_bounds__(CL_TPB) void group_concat_cl_kernel(
    const float *__restrict__ xyz, const float *__restrict__ new_xyz, const int32_t *__restrict__ idx,
    const float *__restrict__ feat, const float *__restrict__ rot, float *__restrict__ out, int n, int m,
    int ns, int c, int mode, float scale, long long total_rows) {
  const int cw = 3 + c;                       // output row width
  const int lanes_per_row = (cw + 3) / 4;     // each thread writes up to 4 consecutive floats
  const long long gid = (long long)blockIdx.x * CL_TPB + threadIdx.x;
  const long long row = gid / lanes_per_row;
  const int part = (int)(gid % lanes_per_row);
  if (row >= total_rows) return;
  const long long grp = row / ns;             // (b*m + j)
  const int bi = (int)(grp / m);
  const int id = idx[row];
  float *dst = out + row * cw;
  const int col0 = part * 4;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int col = col0 + t;
    if (col >= cw) break;
    float v;
    if (col < 3) {
      const float *p = xyz + ((size_t)bi * n + id) * 3;
      const float *q = new_xyz + grp * 3;
      if (mode == 2) {
        const float *r = rot + grp * 9;
        const float x = p[0] - q[0], y = p[1] - q[1], z = p[2] - q[2];
        // torch.matmul(grouped_xyz_(B,m,ns,3), rot(B,m,3,3)): out[col] = x*R[0][col] + y*R[1][col] + z*R[2][col]
        v = ((x * r[col]) + (y * r[3 + col])) + (z * r[6 + col]);
      } else {
        v = p[col] - q[col];
        if (mode == 1) v = v * scale;
      }
    } else {
      v = feat[((size_t)bi * n + id) * c + (col - 3)];
    }
    dst[col] = v;
  }
}

__global__ __launch_bounds__(CL_TPB) void group_concat_cl_grad_kernel(
    const float *__restrict__ dx0, const int32_t *__restrict__ idx, float *__restrict__ dfeat, int n, int m,
    int ns, int c, long long total_rows) {
  const int cw = 3 + c;
  const long long gid = (long long)blockIdx.x * CL_TPB + threadIdx.x;
  const long long row = gid / c;
  const int col = (int)(gid % c);
  if (row >= total_rows) return;
  const int bi = (int)(row / ((long long)m * ns));
  const int id = idx[row];
  atomicAdd(dfeat + ((size_t)bi * n + id) * c + col, dx0[row * cw + 3 + col]);
}

// ------------------------------------------------------------------------------------------------
// Column reductions over a (P, C) row-major matrix.  A block owns ROWS_PER_BLOCK rows; threads are
// laid out (row lane, column) so a wave reads contiguous row segments; per-thread fp64 partials,
// LDS reduce over the row lanes, one fp64 atomic per column per block.
constexpr int ROWS_PER_BLOCK = 1024;

template <class F>  // F(row, col) -> (v1, v2) contributions
__device__ __forceinline__ void col_reduce2(long long P, int C, double *__restrict__ out1,
                                            double *__restrict__ out2, F f) {
  __shared__ double s1[CL_TPB], s2[CL_TPB];
  const long long row0 = (long long)blockIdx.x * ROWS_PER_BLOCK;
  const long long row1 = row0 + ROWS_PER_BLOCK < P ? row0 + ROWS_PER_BLOCK : P;
  for (int cbase = 0; cbase < C; cbase += CL_TPB) {
    const int cols = C - cbase < CL_TPB ? C - cbase : CL_TPB;  // columns handled this round
    int rl = CL_TPB / cols;                                     // row lanes
    if (rl < 1) rl = 1;
    const int col = threadIdx.x % cols;
    const int lane_r = threadIdx.x / cols;
    double a1 = 0.0, a2 = 0.0;
    if (lane_r < rl) {
      for (long long r = row0 + lane_r; r < row1; r += rl) {
        float v1, v2;
        f(r, cbase + col, v1, v2);
        a1 += (double)v1;
        a2 += (double)v2;
      }
    }
    s1[threadIdx.x] = a1;
    s2[threadIdx.x] = a2;
    __syncthreads();
    if (threadIdx.x < cols) {
      double t1 = 0.0, t2 = 0.0;
      for (int l = 0; l < rl; ++l) {
        t1 += s1[l * cols + threadIdx.x];
        t2 += s2[l * cols + threadIdx.x];
      }
      atomicAdd(out1 + cbase + threadIdx.x, t1);
      atomicAdd(out2 + cbase + threadIdx.x, t2);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(CL_TPB) void col_stats_kernel(const float *__restrict__ y, long long P, int C,
                                                            double *__restrict__ sum, double *__restrict__ sumsq) {
  col_reduce2(P, C, sum, sumsq, [&](long long r, int c, float &v1, float &v2) {
    const float v = y[r * C + c];
    v1 = v;
    v2 = v * v;
  });
}

// stats: [sum(C), sumsq(C)] fp64 in;  ab: [a(C), b(C), mean(C), rstd(C)] fp32 out
__global__ void bn_finalize_kernel(const double *__restrict__ stats, long long P, int C,
                                   const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                   float momentum, float *__restrict__ running_mean,
                                   float *__restrict__ running_var, float *__restrict__ ab, int training) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean, var;
  if (training) {
    const double m = stats[c] / (double)P;
    double v = stats[C + c] / (double)P - m * m;  // biased variance (normalisation)
    if (v < 0.0) v = 0.0;
    mean = (float)m;
    var = (float)v;
    if (running_mean) {
      const double unbiased = P > 1 ? v * (double)P / (double)(P - 1) : v;
      running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  } else {
    mean = running_mean[c];
    var = running_var[c];
  }
  const float rstd = 1.0f / sqrtf(var + eps);
  const float a = gamma[c] * rstd;
  ab[c] = a;
  ab[C + c] = beta[c] - mean * a;
  ab[2 * C + c] = mean;
  ab[3 * C + c] = rstd;
}

// z = act(a*y + b + residual)
template <int VEC>
__global__ __launch_bounds__(CL_TPB) void affine_act_kernel(const float *__restrict__ y,
                                                             const float *__restrict__ ab,
                                                             const float *__restrict__ residual,
                                                             float *__restrict__ z, long long total, int C,
                                                             int relu) {
  const long long e = ((long long)blockIdx.x * CL_TPB + threadIdx.x) * VEC;
  if (e >= total) return;
  const int c = (int)(e % C);
  float v[VEC], r[VEC];
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4 *>(y + e);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    if (residual) {
      const float4 q = *reinterpret_cast<const float4 *>(residual + e);
      r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w;
    }
  } else {
    v[0] = y[e];
    if (residual) r[0] = residual[e];
  }
#pragma unroll
  for (int t = 0; t < VEC; ++t) {
    float o = ab[c + t] * v[t] + ab[C + c + t];
    if (residual) o += r[t];
    if (relu) o = o > 0.f ? o : 0.f;
    v[t] = o;
  }
  if constexpr (VEC == 4) *reinterpret_cast<float4 *>(z + e) = make_float4(v[0], v[1], v[2], v[3]);
  else z[e] = v[0];
}

// out[r,c] = max_k relu(a*y[(r*ns+k),c] + b) ; arg[r,c] = first k attaining it
__global__ __launch_bounds__(CL_TPB) void affine_relu_maxpool_kernel(const float *__restrict__ y,
                                                                      const float *__restrict__ ab,
                                                                      float *__restrict__ out,
                                                                      int32_t *__restrict__ arg, long long R,
                                                                      int ns, int C) {
  const long long gid = (long long)blockIdx.x * CL_TPB + threadIdx.x;
  if (gid >= R * C) return;
  const long long r = gid / C;
  const int c = (int)(gid % C);
  const float a = ab[c], b = ab[C + c];
  const float *src = y + (r * ns) * C + c;
  float best = -INFINITY;
  int bk = 0;
  for (int k = 0; k < ns; ++k) {
    float o = a * src[(size_t)k * C] + b;
    o = o > 0.f ? o : 0.f;
    if (o > best) { best = o; bk = k; }
  }
  out[gid] = best;
  arg[gid] = bk;
}

// dense BN(+ReLU)(+residual) backward, pass 1: dA = dOut * [z > 0]; dbeta = sum dA, dgamma = sum dA*xhat
__global__ __launch_bounds__(CL_TPB) void bn_bwd_stats_kernel(const float *__restrict__ dout,
                                                               const float *__restrict__ y,
                                                               const float *__restrict__ ab,
                                                               const float *__restrict__ residual, long long P,
                                                               int C, int relu, double *__restrict__ dbeta,
                                                               double *__restrict__ dgamma) {
  col_reduce2(P, C, dbeta, dgamma, [&](long long r, int c, float &v1, float &v2) {
    const float yy = y[r * C + c];
    float g = dout[r * C + c];
    if (relu) {
      float z = ab[c] * yy + ab[C + c];
      if (residual) z += residual[r * C + c];
      if (!(z > 0.f)) g = 0.f;
    }
    v1 = g;
    v2 = g * ((yy - ab[2 * C + c]) * ab[3 * C + c]);
  });
}

// pass 2: dy = a*(dA - dbeta/P - xhat*dgamma/P)  (training) or a*dA (eval); optionally dA out (residual grad)
__global__ __launch_bounds__(CL_TPB) void bn_bwd_apply_kernel(const float *__restrict__ dout,
                                                               const float *__restrict__ y,
                                                               const float *__restrict__ ab,
                                                               const float *__restrict__ residual,
                                                               const double *__restrict__ dstats, long long P,
                                                               int C, int relu, int training,
                                                               float *__restrict__ dy, float *__restrict__ dres) {
  const long long e = (long long)blockIdx.x * CL_TPB + threadIdx.x;
  if (e >= P * C) return;
  const int c = (int)(e % C);
  const float yy = y[e];
  float g = dout[e];
  if (relu) {
    float z = ab[c] * yy + ab[C + c];
    if (residual) z += residual[e];
    if (!(z > 0.f)) g = 0.f;
  }
  if (dres) dres[e] = g;
  float d = g;
  if (training) {
    const float xhat = (yy - ab[2 * C + c]) * ab[3 * C + c];
    d = g - (float)(dstats[c] / (double)P) - xhat * (float)(dstats[C + c] / (double)P);
  }
  dy[e] = ab[c] * d;
}

// pooled variants: dOut is (R,C); the gradient reaches only the arg-max sample and only if out > 0
__global__ __launch_bounds__(CL_TPB) void bn_bwd_stats_pool_kernel(const float *__restrict__ dout,
                                                                    const float *__restrict__ out,
                                                                    const int32_t *__restrict__ arg,
                                                                    const float *__restrict__ y,
                                                                    const float *__restrict__ ab, long long R,
                                                                    int ns, int C, double *__restrict__ dbeta,
                                                                    double *__restrict__ dgamma) {
  col_reduce2(R, C, dbeta, dgamma, [&](long long r, int c, float &v1, float &v2) {
    const float g = out[r * C + c] > 0.f ? dout[r * C + c] : 0.f;
    const float yy = y[(r * ns + arg[r * C + c]) * C + c];
    v1 = g;
    v2 = g * ((yy - ab[2 * C + c]) * ab[3 * C + c]);
  });
}

__global__ __launch_bounds__(CL_TPB) void bn_bwd_apply_pool_kernel(const float *__restrict__ dout,
                                                                    const float *__restrict__ out,
                                                                    const int32_t *__restrict__ arg,
                                                                    const float *__restrict__ y,
                                                                    const float *__restrict__ ab,
                                                                    const double *__restrict__ dstats, long long P,
                                                                    int ns, int C, int training,
                                                                    float *__restrict__ dy) {
  const long long e = (long long)blockIdx.x * CL_TPB + threadIdx.x;
  if (e >= P * C) return;
  const int c = (int)(e % C);
  const long long p = e / C;
  const long long r = p / ns;
  const int k = (int)(p % ns);
  const long long rc = r * C + c;
  float g = (arg[rc] == k && out[rc] > 0.f) ? dout[rc] : 0.f;
  float d = g;
  if (training) {
    const float xhat = (y[e] - ab[2 * C + c]) * ab[3 * C + c];
    d = g - (float)(dstats[c] / (double)P) - xhat * (float)(dstats[C + c] / (double)P);
  }
  dy[e] = ab[c] * d;
}

static inline unsigned blocks_for(long long work) { return (unsigned)((work + CL_TPB - 1) / CL_TPB); }

}  // namespace gb

using namespace gb;

extern "C" int gb_group_concat_cl(const float *xyz, const float *new_xyz, const int32_t *idx,
                                  const float *feat, const float *rot, float *out, int b, int n, int m,
                                  int ns, int c, int mode, float scale, void *stream) {
  if (b < 0 || n < 1 || m < 0 || ns < 1 || c < 0 || !xyz || !new_xyz || !idx || !out) return GB_EINVAL;
  if (c > 0 && !feat) return GB_EINVAL;
  if (mode < 0 || mode > 2 || (mode == 2 && !rot)) return GB_EINVAL;
  const long long rows = (long long)b * m * ns;
  if (rows == 0) return GB_OK;
  const long long threads = rows * ((3 + c + 3) / 4);
  if (threads / CL_TPB > 0x7fffffffLL) return GB_ERANGE;
  hipLaunchKernelGGL(group_concat_cl_kernel, dim3(blocks_for(threads)), dim3(CL_TPB), 0, as_stream(stream), xyz,
                     new_xyz, idx, feat, rot, out, n, m, ns, c, mode, scale, rows);
  return check_launch("gb_group_concat_cl");
}

extern "C" int gb_group_concat_cl_grad(const float *dx0, const int32_t *idx, float *dfeat, int b, int n,
                                       int m, int ns, int c, void *stream) {
  if (b < 0 || n < 1 || m < 0 || ns < 1 || c < 1 || !dx0 || !idx || !dfeat) return GB_EINVAL;
  const long long rows = (long long)b * m * ns;
  if (rows == 0) return GB_OK;
  if (rows * c / CL_TPB > 0x7fffffffLL) return GB_ERANGE;
  hipLaunchKernelGGL(group_concat_cl_grad_kernel, dim3(blocks_for(rows * c)), dim3(CL_TPB), 0, as_stream(stream),
                     dx0, idx, dfeat, n, m, ns, c, rows);
  return check_launch("gb_group_concat_cl_grad");
}

extern "C" int gb_col_stats(const float *y, long long P, int C, double *stats, void *stream) {
  if (P < 0 || C < 1 || !y || !stats) return GB_EINVAL;
  if (P == 0) return GB_OK;
  hipLaunchKernelGGL(col_stats_kernel, dim3((unsigned)((P + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(CL_TPB), 0,
                     as_stream(stream), y, P, C, stats, stats + C);
  return check_launch("gb_col_stats");
}

extern "C" int gb_bn_finalize(const double *stats, long long P, int C, const float *gamma, const float *beta,
                              float eps, float momentum, float *running_mean, float *running_var, float *ab,
                              int training, void *stream) {
  if (C < 1 || !gamma || !beta || !ab) return GB_EINVAL;
  if (training && (!stats || P < 1)) return GB_EINVAL;
  if (!training && (!running_mean || !running_var)) return GB_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, as_stream(stream), stats, P, C, gamma,
                     beta, eps, momentum, running_mean, running_var, ab, training);
  return check_launch("gb_bn_finalize");
}

extern "C" int gb_affine_act(const float *y, const float *ab, const float *residual, float *z, long long P,
                             int C, int relu, void *stream) {
  if (P < 0 || C < 1 || !y || !ab || !z) return GB_EINVAL;
  const long long total = P * C;
  if (total == 0) return GB_OK;
  const bool vec = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(z) |
                                     reinterpret_cast<uintptr_t>(residual)) % 16 == 0);
  if (vec)
    hipLaunchKernelGGL((affine_act_kernel<4>), dim3(blocks_for(total / 4)), dim3(CL_TPB), 0, as_stream(stream), y, ab,
                       residual, z, total, C, relu);
  else
    hipLaunchKernelGGL((affine_act_kernel<1>), dim3(blocks_for(total)), dim3(CL_TPB), 0, as_stream(stream), y, ab,
                       residual, z, total, C, relu);
  return check_launch("gb_affine_act");
}

extern "C" int gb_affine_relu_maxpool(const float *y, const float *ab, float *out, int32_t *arg, long long R,
                                      int ns, int C, void *stream) {
  if (R < 0 || ns < 1 || C < 1 || !y || !ab || !out || !arg) return GB_EINVAL;
  if (R == 0) return GB_OK;
  hipLaunchKernelGGL(affine_relu_maxpool_kernel, dim3(blocks_for(R * C)), dim3(CL_TPB), 0, as_stream(stream), y, ab,
                     out, arg, R, ns, C);
  return check_launch("gb_affine_relu_maxpool");
}

extern "C" int gb_bn_bwd_stats(const float *dout, const float *y, const float *ab, const float *residual,
                               long long P, int C, int relu, double *dstats, void *stream) {
  if (P < 0 || C < 1 || !dout || !y || !ab || !dstats) return GB_EINVAL;
  if (P == 0) return GB_OK;
  hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3((unsigned)((P + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(CL_TPB),
                     0, as_stream(stream), dout, y, ab, residual, P, C, relu, dstats, dstats + C);
  return check_launch("gb_bn_bwd_stats");
}

extern "C" int gb_bn_bwd_apply(const float *dout, const float *y, const float *ab, const float *residual,
                               const double *dstats, long long P, int C, int relu, int training, float *dy,
                               float *dres, void *stream) {
  if (P < 0 || C < 1 || !dout || !y || !ab || !dy || (training && !dstats)) return GB_EINVAL;
  if (P == 0) return GB_OK;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks_for(P * C)), dim3(CL_TPB), 0, as_stream(stream), dout, y, ab,
                     residual, dstats, P, C, relu, training, dy, dres);
  return check_launch("gb_bn_bwd_apply");
}

extern "C" int gb_bn_bwd_stats_pool(const float *dout, const float *out, const int32_t *arg, const float *y,
                                    const float *ab, long long R, int ns, int C, double *dstats, void *stream) {
  if (R < 0 || ns < 1 || C < 1 || !dout || !out || !arg || !y || !ab || !dstats) return GB_EINVAL;
  if (R == 0) return GB_OK;
  hipLaunchKernelGGL(bn_bwd_stats_pool_kernel, dim3((unsigned)((R + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)),
                     dim3(CL_TPB), 0, as_stream(stream), dout, out, arg, y, ab, R, ns, C, dstats, dstats + C);
  return check_launch("gb_bn_bwd_stats_pool");
}

extern "C" int gb_bn_bwd_apply_pool(const float *dout, const float *out, const int32_t *arg, const float *y,
                                    const float *ab, const double *dstats, long long R, int ns, int C,
                                    int training, float *dy, void *stream) {
  if (R < 0 || ns < 1 || C < 1 || !dout || !out || !arg || !y || !ab || !dy || (training && !dstats)) return GB_EINVAL;
  if (R == 0) return GB_OK;
  hipLaunchKernelGGL(bn_bwd_apply_pool_kernel, dim3(blocks_for(R * ns * C)), dim3(CL_TPB), 0, as_stream(stream), dout,
                     out, arg, y, ab, dstats, R * ns, ns, C, training, dy);
  return check_launch("gb_bn_bwd_apply_pool");
}

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
#include <string.h>

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

// Feature propagation's front end, channel-last (reference pointnet2_modules.py:402-435: three_interpolate of the coarse
// level's features, torch.cat with the skip features, then the SharedMLP): row j of the MLP's input
//   X0[j] = [ (k[i0]*w0 + k[i1]*w1) + k[i2]*w2 (C2 columns) , skip[j] (C1 columns) ]
// written once, straight from the channel-last buffers the neighbouring stacks keep (three_interpolate_kernel's
// expression: the same values as interpolate -> cat -> transpose, which cost four launches and three passes).
__global__ __launch_bounds__(CL_TPB) void interp_concat_cl_kernel(const float *__restrict__ known, const int32_t *__restrict__ idx,
                                                                   const float *__restrict__ weight,
                                                                   const float *__restrict__ skip, float *__restrict__ out,
                                                                   int n, int m, int c2, int c1, long long total_rows) {
  const int cw = c2 + c1;
  const int parts = (cw + 3) / 4;
  const long long gid = (long long)blockIdx.x * CL_TPB + threadIdx.x;
  const long long row = gid / parts;
  const int col0 = (int)(gid % parts) * 4;
  if (row >= total_rows) return;
  const int bi = (int)(row / n);
  const int32_t *ix = idx + row * 3;
  const float *w = weight + row * 3;
  const float *kb = known + (size_t)bi * m * c2;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int col = col0 + t;
    if (col >= cw) break;
    float v;
    if (col < c2) v = ((kb[(size_t)ix[0] * c2 + col] * w[0]) + (kb[(size_t)ix[1] * c2 + col] * w[1])) + (kb[(size_t)ix[2] * c2 + col] * w[2]);
    else v = skip[row * c1 + (col - c2)];
    out[row * cw + col] = v;
  }
}

// ... and its backward: dknown[b, i_k, :] += dX0[j, :C2] * w_k - a row of C2 consecutive floats per (j, k): dense atomics
// (the channel-major three_interpolate_grad scatters single floats) - and dskip[j] = dX0[j, C2:]
__global__ __launch_bounds__(CL_TPB) void interp_concat_cl_grad_kernel(const float *__restrict__ dx0, const int32_t *__restrict__ idx,
                                                                        const float *__restrict__ weight,
                                                                        float *__restrict__ dknown, float *__restrict__ dskip,
                                                                        int n, int m, int c2, int c1, long long total_rows) {
  const int cw = c2 + c1;
  const long long gid = (long long)blockIdx.x * CL_TPB + threadIdx.x;
  const long long row = gid / cw;
  const int col = (int)(gid % cw);
  if (row >= total_rows) return;
  const float g = dx0[row * cw + col];
  if (col >= c2) {
    if (dskip) dskip[row * c1 + (col - c2)] = g;
    return;
  }
  if (!dknown) return;
  const int bi = (int)(row / n);
  const int32_t *ix = idx + row * 3;
  const float *w = weight + row * 3;
  float *kb = dknown + (size_t)bi * m * c2;
  atomicAdd(kb + (size_t)ix[0] * c2 + col, g * w[0]);   // (three_interpolate_grad_kernel's products)
  atomicAdd(kb + (size_t)ix[1] * c2 + col, g * w[1]);
  atomicAdd(kb + (size_t)ix[2] * c2 + col, g * w[2]);
}

// Many small copies as ONE launch: workgroup b copies tab[b] = (source address, destination element, count <= 8192) -
// the gather of 253 parameter gradients into the optimizer's flat buffer (flat_adam.FlatAdam.pack) took nine
// multi-tensor launches of torch's (0.17 ms for 36 MB).
struct CopySeg {
  const float *src;
  long long dst_off;
  long long n;
};
__global__ __launch_bounds__(CL_TPB) void copy_segments_kernel(const CopySeg *__restrict__ tab, float *__restrict__ dst) {
  const CopySeg sg = tab[blockIdx.x];
  float *d = dst + sg.dst_off;
  const int n = (int)sg.n;
  if (((reinterpret_cast<uintptr_t>(sg.src) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) {
    const int n4 = n >> 2;
    for (int i = threadIdx.x; i < n4; i += CL_TPB)
      reinterpret_cast<float4 *>(d)[i] = reinterpret_cast<const float4 *>(sg.src)[i];
    for (int i = (n4 << 2) + threadIdx.x; i < n; i += CL_TPB) d[i] = sg.src[i];
  } else {
    for (int i = threadIdx.x; i < n; i += CL_TPB) d[i] = sg.src[i];
  }
}

// ------------------------------------------------------------------------------------------------
// Column reductions over a (P, C) row-major matrix.  A block owns ROWS_PER_BLOCK rows; threads are
// laid out (row lane, column) so a wave reads contiguous row segments; per-thread fp64 partials,
// LDS reduce over the row lanes, one fp64 atomic per column per block.
// HBM-bound: each thread owns VEC (4 when C % 4 == 0) adjacent columns and reads them with one 16-byte
// load per row; 4 rows are in flight per thread (independent fp32 accumulators, flushed to fp64 after
// at most ROWS_PER_BLOCK / row-lanes rows, i.e. <= 64 terms per fp32 partial).
constexpr int ROWS_PER_BLOCK = 512;

template <int VEC, class F>  // F(row, col, v1[VEC], v2[VEC]) -> contributions of columns col..col+VEC-1
__device__ __forceinline__ void col_reduce2(long long P, int C, int rpb, double *__restrict__ out1,
                                            double *__restrict__ out2, F f) {
  __shared__ double s1[CL_TPB * VEC], s2[CL_TPB * VEC];
  const long long row0 = (long long)blockIdx.x * rpb;
  const long long row1 = row0 + rpb < P ? row0 + rpb : P;
  const int CV = C / VEC;  // column groups
  for (int cbase = 0; cbase < CV; cbase += CL_TPB) {
    const int cols = CV - cbase < CL_TPB ? CV - cbase : CL_TPB;  // column groups handled this round
    const int rl = CL_TPB / cols;                                 // row lanes (>= 1)
    const int cg = threadIdx.x % cols;
    const int lane_r = threadIdx.x / cols;
    const int col = (cbase + cg) * VEC;
    float p1[4][VEC], p2[4][VEC];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < VEC; ++t) { p1[u][t] = 0.f; p2[u][t] = 0.f; }
    if (lane_r < rl) {
      long long r = row0 + lane_r;
      for (; r + 3LL * rl < row1; r += 4LL * rl) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float v1[VEC], v2[VEC];
          f(r + (long long)u * rl, col, v1, v2);
#pragma unroll
          for (int t = 0; t < VEC; ++t) { p1[u][t] += v1[t]; p2[u][t] += v2[t]; }
        }
      }
      for (; r < row1; r += rl) {
        float v1[VEC], v2[VEC];
        f(r, col, v1, v2);
#pragma unroll
        for (int t = 0; t < VEC; ++t) { p1[0][t] += v1[t]; p2[0][t] += v2[t]; }
      }
    }
#pragma unroll
    for (int t = 0; t < VEC; ++t) {
      s1[threadIdx.x * VEC + t] = ((double)p1[0][t] + (double)p1[1][t]) + ((double)p1[2][t] + (double)p1[3][t]);
      s2[threadIdx.x * VEC + t] = ((double)p2[0][t] + (double)p2[1][t]) + ((double)p2[2][t] + (double)p2[3][t]);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < cols * VEC; o += CL_TPB) {  // o = cg*VEC + t
      double t1 = 0.0, t2 = 0.0;
      for (int l = 0; l < rl; ++l) {
        t1 += s1[l * cols * VEC + o];
        t2 += s2[l * cols * VEC + o];
      }
      atomicAdd(out1 + cbase * VEC + o, t1);
      atomicAdd(out2 + cbase * VEC + o, t2);
    }
    __syncthreads();
  }
}

template <int VEC>
__device__ __forceinline__ void load_vec(const float *p, float *v) {
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4 *>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    v[0] = p[0];
  }
}

template <int VEC>
__global__ __launch_bounds__(CL_TPB) void col_stats_kernel(const float *__restrict__ y, long long P, int C, int rpb,
                                                            double *__restrict__ sum, double *__restrict__ sumsq) {
  col_reduce2<VEC>(P, C, rpb, sum, sumsq, [&](long long r, int c, float *v1, float *v2) {
    load_vec<VEC>(y + r * C + c, v1);
#pragma unroll
    for (int t = 0; t < VEC; ++t) v2[t] = v1[t] * v1[t];
  });
}

// The closing pass of a split forward product (csrc/gemm_cl.hip): y = ((part_0 + part_1) + part_2) + ... in chunk order
// (bit-reproducible) AND the column sums of y and y^2 in the same sweep - what split_reduce_kernel followed by
// col_stats_kernel did in two launches and a second read of y.  C % 4 == 0, 16-byte aligned.
__global__ __launch_bounds__(CL_TPB) void split_col_stats_kernel(const float *__restrict__ part, int chunks,
                                                                  long long elems, float *__restrict__ y, long long P,
                                                                  int C, int rpb, double *__restrict__ sum,
                                                                  double *__restrict__ sumsq) {
  col_reduce2<4>(P, C, rpb, sum, sumsq, [&](long long r, int c, float *v1, float *v2) {
    const long long i = r * C + c;
    float4 acc = *reinterpret_cast<const float4 *>(part + i);
    for (int q = 1; q < chunks; ++q) {
      const float4 v = *reinterpret_cast<const float4 *>(part + (long long)q * elems + i);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4 *>(y + i) = acc;
    v1[0] = acc.x; v1[1] = acc.y; v1[2] = acc.z; v1[3] = acc.w;
#pragma unroll
    for (int t = 0; t < 4; ++t) v2[t] = v1[t] * v1[t];
  });
}

// stats: [sum(C), sumsq(C)] fp64 in;  ab: [a(C), b(C), mean(C), rstd(C)] fp32 out
__global__ void bn_finalize_kernel(const double *__restrict__ stats, int slots, long long P, int C,
                                   const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                   float momentum, float *__restrict__ running_mean,
                                   float *__restrict__ running_var, float *__restrict__ ab, int training) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (training) {
    double s1 = 0.0, s2 = 0.0;
    for (int sl = 0; sl < slots; ++sl) {  // partial sums of the GEMM epilogue's slot rows
      s1 += stats[(size_t)sl * 2 * C + c];
      s2 += stats[(size_t)sl * 2 * C + C + c];
    }
    const BnFinalize f = {gamma, beta, running_mean, running_var, ab, P, eps, momentum, 1};
    bn_finalize_column(f, s1, s2, c, C);
    return;
  }
  const float mean = running_mean[c], var = running_var[c];
  const float rstd = 1.0f / sqrtf(var + eps);
  const float a = gamma[c] * rstd;
  ab[c] = a;
  ab[C + c] = beta[c] - mean * a;
  ab[2 * C + c] = mean;
  ab[3 * C + c] = rstd;
}

// BatchNorm finalisation of a 3-input linear layer y = x W^T from the 12 moments of its input rows (gb_moments3: s = sum
// w x, M = sum w x x^T over the batch): sum y_c = W_c . s, sum y_c^2 = W_c^T M W_c - the layer's output is never formed.
__global__ void bn_finalize_lin3_kernel(const double *__restrict__ mom, const float *__restrict__ w, long long P, int C,
                                        const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                        float momentum, float *__restrict__ running_mean,
                                        float *__restrict__ running_var, float *__restrict__ ab) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double w0 = w[c * 3], w1 = w[c * 3 + 1], w2 = w[c * 3 + 2];
  const double s1 = w0 * mom[0] + w1 * mom[1] + w2 * mom[2];
  const double s2 = w0 * (w0 * mom[3] + w1 * mom[4] + w2 * mom[5]) + w1 * (w0 * mom[6] + w1 * mom[7] + w2 * mom[8]) +
                    w2 * (w0 * mom[9] + w1 * mom[10] + w2 * mom[11]);
  const BnFinalize f = {gamma, beta, running_mean, running_var, ab, P, eps, momentum, 1};
  bn_finalize_column(f, s1, s2, c, C);
}

// dst: fp64 [slots][2C] partial BatchNorm-backward sums -> dstats fp64 [2C] (their total, the form
// gb_bn_bwd_apply reads) and the parameter gradients dbeta = sum dA, dgamma = sum dA*xhat in fp32.
__global__ void bn_bwd_reduce_kernel(const double *__restrict__ dst, int slots, int C, double *__restrict__ dstats,
                                     float *__restrict__ dbeta, float *__restrict__ dgamma) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= 2 * C) return;
  double s = 0.0;
  for (int sl = 0; sl < slots; ++sl) s += dst[(size_t)sl * 2 * C + j];
  if (dstats) dstats[j] = s;
  if (j < C) dbeta[j] = (float)s;
  else dgamma[j - C] = (float)s;
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

// 4 columns per thread, 8 rows of the group in flight: the scalar kernel above issues one dependent
// dword load per sample (3.4 TB/s); this one streams 16-byte loads (same first-maximum tie rule)
__global__ __launch_bounds__(CL_TPB) void affine_relu_maxpool_vec_kernel(const float *__restrict__ y,
                                                                          const float *__restrict__ ab,
                                                                          float *__restrict__ out,
                                                                          int32_t *__restrict__ arg, long long R,
                                                                          int ns, int C) {
  const int c4 = C / 4;
  const long long gid = (long long)blockIdx.x * CL_TPB + threadIdx.x;
  if (gid >= R * c4) return;
  const long long r = gid / c4;
  const int c = (int)(gid % c4) * 4;
  float a[4], b[4], best[4];
  int bk[4];
  load_vec<4>(ab + c, a);
  load_vec<4>(ab + C + c, b);
#pragma unroll
  for (int t = 0; t < 4; ++t) { best[t] = -INFINITY; bk[t] = 0; }
  const float *src = y + (r * ns) * C + c;
  for (int k0 = 0; k0 < ns; k0 += 8) {
    float v[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k0 + u < ns) load_vec<4>(src + (size_t)(k0 + u) * C, v[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k0 + u < ns) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float o = a[t] * v[u][t] + b[t];
          o = o > 0.f ? o : 0.f;
          if (o > best[t]) { best[t] = o; bk[t] = k0 + u; }
        }
      }
  }
  *reinterpret_cast<float4 *>(out + r * C + c) = make_float4(best[0], best[1], best[2], best[3]);
  *reinterpret_cast<int4 *>(arg + r * C + c) = make_int4(bk[0], bk[1], bk[2], bk[3]);
}

// dense BN(+ReLU)(+residual) backward, pass 1: dA = dOut * [z > 0]; dbeta = sum dA, dgamma = sum dA*xhat
template <int VEC>
__global__ __launch_bounds__(CL_TPB) void bn_bwd_stats_kernel(const float *__restrict__ dout,
                                                               const float *__restrict__ y,
                                                               const float *__restrict__ ab,
                                                               const float *__restrict__ residual, long long P,
                                                               int C, int rpb, int relu,
                                                               double *__restrict__ dbeta,
                                                               double *__restrict__ dgamma) {
  col_reduce2<VEC>(P, C, rpb, dbeta, dgamma, [&](long long r, int c, float *v1, float *v2) {
    float yy[VEC], g[VEC], rs[VEC];
    load_vec<VEC>(y + r * C + c, yy);
    load_vec<VEC>(dout + r * C + c, g);
    if (residual) load_vec<VEC>(residual + r * C + c, rs);
#pragma unroll
    for (int t = 0; t < VEC; ++t) {
      float gg = g[t];
      if (relu) {
        float z = ab[c + t] * yy[t] + ab[C + c + t];
        if (residual) z += rs[t];
        if (!(z > 0.f)) gg = 0.f;
      }
      v1[t] = gg;
      v2[t] = gg * ((yy[t] - ab[2 * C + c + t]) * ab[3 * C + c + t]);
    }
  });
}

// The closing pass of a split dgrad product: dz = the chunk-ordered sum of the partial products, stored, and the
// BatchNorm-backward sums of the layer dz is the gradient of (ReLU mask from its pre-BN output y) in the same sweep -
// split_reduce_kernel + bn_bwd_stats_kernel<4> in one launch and one read of dz less.
__global__ __launch_bounds__(CL_TPB) void split_bn_bwd_stats_kernel(const float *__restrict__ part, int chunks,
                                                                     long long elems, float *__restrict__ dz,
                                                                     const float *__restrict__ y,
                                                                     const float *__restrict__ ab, long long P, int C,
                                                                     int rpb, double *__restrict__ dbeta,
                                                                     double *__restrict__ dgamma) {
  col_reduce2<4>(P, C, rpb, dbeta, dgamma, [&](long long r, int c, float *v1, float *v2) {
    const long long i = r * C + c;
    float4 acc = *reinterpret_cast<const float4 *>(part + i);
    for (int q = 1; q < chunks; ++q) {
      const float4 v = *reinterpret_cast<const float4 *>(part + (long long)q * elems + i);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4 *>(dz + i) = acc;
    const float g[4] = {acc.x, acc.y, acc.z, acc.w};
    float yy[4];
    load_vec<4>(y + i, yy);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float z = ab[c + t] * yy[t] + ab[C + c + t];
      const float gg = z > 0.f ? g[t] : 0.f;
      v1[t] = gg;
      v2[t] = gg * ((yy[t] - ab[2 * C + c + t]) * ab[3 * C + c + t]);
    }
  });
}

// pass 2: dy = a*(dA - dbeta/P - xhat*dgamma/P)  (training) or a*dA (eval); optionally dA out (residual grad)
template <int VEC>
__global__ __launch_bounds__(CL_TPB) void bn_bwd_apply_kernel(const float *__restrict__ dout,
                                                               const float *__restrict__ y,
                                                               const float *__restrict__ ab,
                                                               const float *__restrict__ residual,
                                                               const double *__restrict__ dstats, long long P,
                                                               int C, int relu, int training,
                                                               float *__restrict__ dy, float *__restrict__ dres) {
  const long long e = ((long long)blockIdx.x * CL_TPB + threadIdx.x) * VEC;
  if (e >= P * C) return;
  const int c = (int)(e % C);
  const double invP = 1.0 / (double)P;
  float yy[VEC], g[VEC], rs[VEC], d[VEC];
  load_vec<VEC>(y + e, yy);
  load_vec<VEC>(dout + e, g);
  if (residual) load_vec<VEC>(residual + e, rs);
#pragma unroll
  for (int t = 0; t < VEC; ++t) {
    if (relu) {
      float z = ab[c + t] * yy[t] + ab[C + c + t];
      if (residual) z += rs[t];
      if (!(z > 0.f)) g[t] = 0.f;
    }
    float dd = g[t];
    if (training) {
      const float xhat = (yy[t] - ab[2 * C + c + t]) * ab[3 * C + c + t];
      dd = g[t] - (float)(dstats[c + t] * invP) - xhat * (float)(dstats[C + c + t] * invP);
    }
    d[t] = ab[c + t] * dd;
  }
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4 *>(dy + e) = make_float4(d[0], d[1], d[2], d[3]);
    if (dres) *reinterpret_cast<float4 *>(dres + e) = make_float4(g[0], g[1], g[2], g[3]);
  } else {
    dy[e] = d[0];
    if (dres) dres[e] = g[0];
  }
}

// Same arithmetic, organised for bandwidth: a thread owns 4 columns for a whole chunk of rows, so the six
// per-column coefficients (a, b, mean, rstd, dbeta/P, dgamma/P) are fetched once into registers instead
// of once per 4 elements (the element-per-thread form above is bound by those extra L1 requests: 2.1 TB/s),
// and 4 rows are in flight per thread.  Requires C % 4 == 0, C/4 <= 256 and 16-byte aligned rows.
__global__ __launch_bounds__(CL_TPB) void bn_bwd_apply_rows_kernel(const float *__restrict__ dout,
                                                                    const float *__restrict__ y,
                                                                    const float *__restrict__ ab,
                                                                    const float *__restrict__ residual,
                                                                    const double *__restrict__ dstats, long long P,
                                                                    int C, int relu, int training,
                                                                    float *__restrict__ dy, float *__restrict__ dres,
                                                                    int rows_per_block, float *__restrict__ dbeta,
                                                                    float *__restrict__ dgamma) {
  const int tpr = C / 4, rpp = CL_TPB / tpr;  // threads per row, rows per pass
  const int cg = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  if (rl >= rpp) return;
  const int c = cg * 4;
  const double invP = 1.0 / (double)P;
  float ka[4], kb[4], km[4], kr[4], k1[4], k2[4];
  load_vec<4>(ab + c, ka);
  load_vec<4>(ab + C + c, kb);
  load_vec<4>(ab + 2 * C + c, km);
  load_vec<4>(ab + 3 * C + c, kr);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    k1[t] = training ? (float)(dstats[c + t] * invP) : 0.f;
    k2[t] = training ? (float)(dstats[C + c + t] * invP) : 0.f;
  }
  if (dbeta && blockIdx.x == 0 && rl == 0) {
    // gb_bn_bwd_apply_g: the layer's parameter gradients (fp32) are the sums this kernel reads anyway - the separate
    // gb_bn_bwd_reduce launch of a one-slot-row layer only converted them (41 launches of ~5 us per train step)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      dbeta[c + t] = (float)dstats[c + t];
      dgamma[c + t] = (float)dstats[C + c + t];
    }
  }
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > P) r1 = P;
  for (long long r = r0 + rl; r < r1; r += 4 * rpp) {
    float yy[4][4], g[4][4], rs[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long rr = r + (long long)u * rpp;
      if (rr < r1) {
        load_vec<4>(y + rr * C + c, yy[u]);
        load_vec<4>(dout + rr * C + c, g[u]);
        if (residual) load_vec<4>(residual + rr * C + c, rs[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long rr = r + (long long)u * rpp;
      if (rr < r1) {
        float d[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (relu) {
            float z = ka[t] * yy[u][t] + kb[t];
            if (residual) z += rs[u][t];
            if (!(z > 0.f)) g[u][t] = 0.f;
          }
          float dd = g[u][t];
          if (training) {
            const float xhat = (yy[u][t] - km[t]) * kr[t];
            dd = g[u][t] - k1[t] - xhat * k2[t];
          }
          d[t] = ka[t] * dd;
        }
        *reinterpret_cast<float4 *>(dy + rr * C + c) = make_float4(d[0], d[1], d[2], d[3]);
        if (dres) *reinterpret_cast<float4 *>(dres + rr * C + c) = make_float4(g[u][0], g[u][1], g[u][2], g[u][3]);
      }
    }
  }
}

// ---- rows with multiplicities and cylinder membership (csrc/cyl_rows.hip) ------------------------------------
// Seed r owns the distinct rows [off[r], off[r] + cnt[r]); row_mem has bit d set when the row's point lies in the
// seed's d-th cylinder; row_w is how many slots of the original (D x ns) grouping hold it.

constexpr int MEM_RIF_BWD = 4;  // rows in flight per thread in bn_bwd_apply_members_kernel

// out[(r*D + d), c] = max over the rows of seed r with bit d of relu(a*y + b); arg = that row's absolute index.
// A thread owns 4 columns of one seed and streams the seed's rows (4 in flight).
template <int D>
__global__ __launch_bounds__(CL_TPB) void affine_relu_maxpool_members_kernel(
    const float *__restrict__ y, const float *__restrict__ ab, const int32_t *__restrict__ row_mem,
    const int64_t *__restrict__ off, const int32_t *__restrict__ cnt, float *__restrict__ out,
    int32_t *__restrict__ arg, long long R, int C) {
  const int tpg = C / 4, gpb = CL_TPB / tpg;
  const long long r = (long long)blockIdx.x * gpb + threadIdx.x / tpg;
  if (threadIdx.x / tpg >= gpb || r >= R) return;
  const int c = (threadIdx.x % tpg) * 4;
  float a[4], b[4], best[D][4];
  int bk[D][4];
  load_vec<4>(ab + c, a);
  load_vec<4>(ab + C + c, b);
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int t = 0; t < 4; ++t) { best[d][t] = -INFINITY; bk[d][t] = 0; }
  const long long u0 = off[r], u1 = u0 + cnt[r];
  for (long long u = u0; u < u1; u += 4) {
    float v[4][4];
    int mem[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (u + q < u1) { load_vec<4>(y + (u + q) * C + c, v[q]); mem[q] = row_mem[u + q]; }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (u + q < u1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float o = a[t] * v[q][t] + b[t];
          o = o > 0.f ? o : 0.f;
#pragma unroll
          for (int d = 0; d < D; ++d)
            if (((mem[q] >> d) & 1) && o > best[d][t]) { best[d][t] = o; bk[d][t] = (int)(u + q); }
        }
      }
  }
#pragma unroll
  for (int d = 0; d < D; ++d) {
    *reinterpret_cast<float4 *>(out + (r * D + d) * C + c) = make_float4(best[d][0], best[d][1], best[d][2], best[d][3]);
    *reinterpret_cast<int4 *>(arg + (r * D + d) * C + c) = make_int4(bk[d][0], bk[d][1], bk[d][2], bk[d][3]);
  }
}

// Second half of the pooled last layer (gb_gemm_fwd_pool): seed r's rows [off, off + cnt) span the 32-row tiles
// t0 = off / 32 .. t1 = (off + cnt - 1) / 32, and the GEMM left, for each of them, the extreme of sign(gamma)*y over
// the seed's members of crop d in that tile at pairs[(t + r)][d][c].  With a = gamma*rstd of the same sign, max over
// the crop of relu(a*y + b) = relu(a*y* + b) at that extreme y*: out and ystar = y* (what the backward needs: it finds
// the arg-max row by value in the stored output).  A seed without rows wrote no slot: out = ystar = 0.
template <int D>
__global__ __launch_bounds__(CL_TPB) void pool_pairs_kernel(const float *__restrict__ pairs,
                                                            const int64_t *__restrict__ off,
                                                            const int32_t *__restrict__ cnt,
                                                            const float *__restrict__ ab,
                                                            const float *__restrict__ gamma, float *__restrict__ out,
                                                            float *__restrict__ ystar, long long R, int C) {
  const int gpb = CL_TPB / C;  // C threads per seed (one column each: consecutive lanes read consecutive slots)
  const long long r = (long long)blockIdx.x * gpb + threadIdx.x / C;
  if (threadIdx.x / C >= gpb || r >= R) return;
  const int c = threadIdx.x % C;
  const float a = ab[c], b = ab[C + c], sg = gamma[c] < 0.f ? -1.f : 1.f;
  const long long u0 = off[r], n = cnt[r];
  const long long t0 = u0 / 32, t1 = n > 0 ? (u0 + n - 1) / 32 : t0 - 1;   // no rows: no tile wrote a slot of this seed
#pragma unroll
  for (int d = 0; d < D; ++d) {
    float best = -INFINITY;
    for (long long t = t0; t <= t1; ++t) best = fmaxf(best, pairs[((size_t)(t + r) * D + d) * C + c]);
    const bool any = best > -INFINITY;
    const float y = any ? sg * best : 0.f;
    float o = a * y + b;
    o = (any && o > 0.f) ? o : 0.f;
    const size_t at = (size_t)(r * D + d) * C + c;
    out[at] = o;
    ystar[at] = y;
  }
}

// BatchNorm + ReLU + member-max-pool backward for the distinct rows: the gradient of row u sums, over the cylinders
// d whose arg-max it is, dout*[out > 0]; every COPY of the row also receives the -(dbeta/P + xhat*dgamma/P) terms, so
// with multiplicity w:  dy[u] = a*(g - w*(dbeta/P) - xhat*(w*dgamma/P)),  P = rows of the original batch.
// BYVAL: no arg rows - `arg` is reinterpreted as ystar ((R*D), C) floats and `row_mem` gives the member bits: the gradient
// of crop d goes to the FIRST row of the seed that is a member of d and whose y equals the crop's extreme y* (the
// row an arg-max in row order would name; the pooled GEMM epilogue leaves values only).
template <int D, bool BYVAL = false>
__global__ __launch_bounds__(CL_TPB) void bn_bwd_apply_members_kernel(
    const float *__restrict__ dout, const float *__restrict__ out, const int32_t *__restrict__ arg,
    const float *__restrict__ y, const float *__restrict__ ab, const double *__restrict__ dstats,
    const float *__restrict__ row_w, const int64_t *__restrict__ off, const int32_t *__restrict__ cnt, long long R,
    int C, double invP, int training, float *__restrict__ dy, const int32_t *__restrict__ row_mem = nullptr) {
  const int tpg = C / 4, gpb = CL_TPB / tpg;
  const long long r = (long long)blockIdx.x * gpb + threadIdx.x / tpg;
  if (threadIdx.x / tpg >= gpb || r >= R) return;
  const int c = (threadIdx.x % tpg) * 4;
  float ka[4], km[4], kr[4], k1[4], k2[4], g[D][4];
  int ar[D][4];
  load_vec<4>(ab + c, ka);
  load_vec<4>(ab + 2 * C + c, km);
  load_vec<4>(ab + 3 * C + c, kr);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    k1[t] = training ? (float)(dstats[c + t] * invP) : 0.f;
    k2[t] = training ? (float)(dstats[C + c + t] * invP) : 0.f;
  }
#pragma unroll
  for (int d = 0; d < D; ++d) {
    float o[4];
    load_vec<4>(out + (r * D + d) * C + c, o);
    load_vec<4>(dout + (r * D + d) * C + c, g[d]);
    const int4 q = *reinterpret_cast<const int4 *>(arg + (r * D + d) * C + c);   // BYVAL: the bits of ystar
    ar[d][0] = q.x; ar[d][1] = q.y; ar[d][2] = q.z; ar[d][3] = q.w;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (!(o[t] > 0.f)) g[d][t] = 0.f;
  }
  const long long u0 = off[r], u1 = u0 + cnt[r];
  for (long long u = u0; u < u1; u += MEM_RIF_BWD) {
    float v[MEM_RIF_BWD][4], w[MEM_RIF_BWD];
    int mb[MEM_RIF_BWD];
#pragma unroll
    for (int q = 0; q < MEM_RIF_BWD; ++q) {
      // unconditional loads from a clamped row index: under `if (u + q < u1)` the four loads were waited for one
      // by one (295 -> 180 us per launch)
      const long long uu = u + q < u1 ? u + q : u1 - 1;
      load_vec<4>(y + uu * C + c, v[q]);
      w[q] = row_w[uu];
      mb[q] = BYVAL ? row_mem[uu] : 0;
    }
#pragma unroll
    for (int q = 0; q < MEM_RIF_BWD; ++q)
      if (u + q < u1) {
        float dd[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float gs = 0.f;
#pragma unroll
          for (int d = 0; d < D; ++d) {
            if constexpr (BYVAL) {
              const bool hit = ((mb[q] >> d) & 1) && v[q][t] == __int_as_float(ar[d][t]);
              gs += hit ? g[d][t] : 0.f;
              if (hit) g[d][t] = 0.f;   // first match only (rows are visited in order)
            } else {
              gs += (ar[d][t] == (int)(u + q)) ? g[d][t] : 0.f;
            }
          }
          if (training) {
            const float xhat = (v[q][t] - km[t]) * kr[t];
            gs = gs - w[q] * k1[t] - xhat * (w[q] * k2[t]);
          }
          dd[t] = ka[t] * gs;
        }
        *reinterpret_cast<float4 *>(dy + (u + q) * C + c) = make_float4(dd[0], dd[1], dd[2], dd[3]);
      }
  }
}

// dense weighted form of bn_bwd_apply_rows_kernel (ReLU, no residual): dy = a*(dA*[z>0] - w*dbeta/P - xhat*w*dgamma/P)
__global__ __launch_bounds__(CL_TPB) void bn_bwd_apply_w_kernel(const float *__restrict__ dout, const float *__restrict__ y,
                                                                 const float *__restrict__ ab,
                                                                 const double *__restrict__ dstats,
                                                                 const float *__restrict__ row_w, long long rows, int C,
                                                                 double invP, int training, float *__restrict__ dy,
                                                                 int rows_per_block,
                                                                 const long long *__restrict__ rows_dev) {
  if (rows_dev) {   // the caller's device-side row count (<= the capacity `rows` the grid was sized for)
    const long long pd = *rows_dev;
    rows = pd < rows ? (pd > 0 ? pd : 0) : rows;
  }
  const int tpr = C / 4, rpp = CL_TPB / tpr;
  const int cg = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  if (rl >= rpp) return;
  const int c = cg * 4;
  float ka[4], kb[4], km[4], kr[4], k1[4], k2[4];
  load_vec<4>(ab + c, ka);
  load_vec<4>(ab + C + c, kb);
  load_vec<4>(ab + 2 * C + c, km);
  load_vec<4>(ab + 3 * C + c, kr);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    k1[t] = training ? (float)(dstats[c + t] * invP) : 0.f;
    k2[t] = training ? (float)(dstats[C + c + t] * invP) : 0.f;
  }
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  for (long long r = r0 + rl; r < r1; r += 4 * rpp) {
    float yy[4][4], g[4][4], w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long rr = r + (long long)u * rpp;
      if (rr < r1) {
        load_vec<4>(y + rr * C + c, yy[u]);
        load_vec<4>(dout + rr * C + c, g[u]);
        w[u] = row_w[rr];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long rr = r + (long long)u * rpp;
      if (rr < r1) {
        float d[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float z = ka[t] * yy[u][t] + kb[t];
          float dd = z > 0.f ? g[u][t] : 0.f;
          if (training) {
            const float xhat = (yy[u][t] - km[t]) * kr[t];
            dd = dd - w[u] * k1[t] - xhat * (w[u] * k2[t]);
          }
          d[t] = ka[t] * dd;
        }
        *reinterpret_cast<float4 *>(dy + rr * C + c) = make_float4(d[0], d[1], d[2], d[3]);
      }
    }
  }
}

// pooled variants: dOut is (R,C); the gradient reaches only the arg-max sample and only if out > 0
template <int VEC>
__global__ __launch_bounds__(CL_TPB) void bn_bwd_stats_pool_kernel(const float *__restrict__ dout,
                                                                    const float *__restrict__ out,
                                                                    const int32_t *__restrict__ arg,
                                                                    const float *__restrict__ y,
                                                                    const float *__restrict__ ab, long long R,
                                                                    int ns, int C, int rpb,
                                                                    double *__restrict__ dbeta,
                                                                    double *__restrict__ dgamma) {
  col_reduce2<VEC>(R, C, rpb, dbeta, dgamma, [&](long long r, int c, float *v1, float *v2) {
    float o[VEC], g[VEC];
    load_vec<VEC>(out + r * C + c, o);
    load_vec<VEC>(dout + r * C + c, g);
#pragma unroll
    for (int t = 0; t < VEC; ++t) {
      const float gg = o[t] > 0.f ? g[t] : 0.f;
      const float yy = y[(r * ns + arg[r * C + c + t]) * C + c + t];
      v1[t] = gg;
      v2[t] = gg * ((yy - ab[2 * C + c + t]) * ab[3 * C + c + t]);
    }
  });
}

// one thread per (pooled row r, VEC channels): the row's gradient / arg-max are read once, then the
// ns samples of the row are streamed (16-byte loads and stores)
template <int VEC>
__global__ __launch_bounds__(CL_TPB) void bn_bwd_apply_pool_kernel(const float *__restrict__ dout,
                                                                    const float *__restrict__ out,
                                                                    const int32_t *__restrict__ arg,
                                                                    const float *__restrict__ y,
                                                                    const float *__restrict__ ab,
                                                                    const double *__restrict__ dstats, long long R,
                                                                    int ns, int C, int training,
                                                                    float *__restrict__ dy) {
  const int CV = C / VEC;
  const long long gid = (long long)blockIdx.x * CL_TPB + threadIdx.x;
  if (gid >= R * CV) return;
  const long long r = gid / CV;
  const int c = (int)(gid % CV) * VEC;
  const double invP = 1.0 / ((double)R * ns);
  float g[VEC], a[VEC], mean[VEC], rstd[VEC], m1[VEC], m2[VEC];
  int ak[VEC];
  {
    float o[VEC];
    load_vec<VEC>(out + r * C + c, o);
    load_vec<VEC>(dout + r * C + c, g);
#pragma unroll
    for (int t = 0; t < VEC; ++t) {
      if (!(o[t] > 0.f)) g[t] = 0.f;
      ak[t] = arg[r * C + c + t];
      a[t] = ab[c + t];
      mean[t] = ab[2 * C + c + t];
      rstd[t] = ab[3 * C + c + t];
      m1[t] = training ? (float)(dstats[c + t] * invP) : 0.f;
      m2[t] = training ? (float)(dstats[C + c + t] * invP) : 0.f;
    }
  }
  const float *src = y + (r * ns) * C + c;
  float *dst = dy + (r * ns) * C + c;
  for (int k = 0; k < ns; ++k) {
    float yy[VEC], d[VEC];
    load_vec<VEC>(src + (size_t)k * C, yy);
#pragma unroll
    for (int t = 0; t < VEC; ++t) {
      const float gg = ak[t] == k ? g[t] : 0.f;
      d[t] = training ? a[t] * (gg - m1[t] - ((yy[t] - mean[t]) * rstd[t]) * m2[t]) : a[t] * gg;
    }
    if constexpr (VEC == 4) *reinterpret_cast<float4 *>(dst + (size_t)k * C) = make_float4(d[0], d[1], d[2], d[3]);
    else dst[(size_t)k * C] = d[0];
  }
}

static inline unsigned blocks_for(long long work) { return (unsigned)((work + CL_TPB - 1) / CL_TPB); }

// rows per reduction block: enough blocks (~2048) to fill 256 CUs, at most ROWS_PER_BLOCK rows each
static inline int rows_per_block(long long rows) {
  long long rpb = (rows + 2047) / 2048;
  if (rpb < 16) rpb = 16;
  if (rpb > ROWS_PER_BLOCK) rpb = ROWS_PER_BLOCK;
  return (int)rpb;
}

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

extern "C" int gb_copy_segments(const void *table, int segments, float *dst, void *stream) {
  if (segments < 0 || (segments > 0 && (!table || !dst)) || reinterpret_cast<uintptr_t>(table) % 8 != 0) return GB_EINVAL;
  if (segments == 0) return GB_OK;
  static_assert(sizeof(CopySeg) == 24, "three 8-byte fields: the caller's table layout");
  hipLaunchKernelGGL(copy_segments_kernel, dim3((unsigned)segments), dim3(CL_TPB), 0, as_stream(stream),
                     static_cast<const CopySeg *>(table), dst);
  return check_launch("gb_copy_segments");
}

extern "C" int gb_interp_concat_cl(const float *known, const int32_t *idx, const float *weight, const float *skip,
                                   float *out, int b, int n, int m, int c2, int c1, void *stream) {
  if (b < 0 || n < 0 || m < 1 || c2 < 1 || c1 < 0 || !known || !idx || !weight || !out || (c1 > 0 && !skip)) return GB_EINVAL;
  const long long rows = (long long)b * n;
  if (rows == 0) return GB_OK;
  const long long threads = rows * ((c2 + c1 + 3) / 4);
  if (threads / CL_TPB > 0x7fffffffLL) return GB_ERANGE;
  hipLaunchKernelGGL(interp_concat_cl_kernel, dim3(blocks_for(threads)), dim3(CL_TPB), 0, as_stream(stream), known, idx,
                     weight, skip, out, n, m, c2, c1, rows);
  return check_launch("gb_interp_concat_cl");
}

extern "C" int gb_interp_concat_cl_grad(const float *dx0, const int32_t *idx, const float *weight, float *dknown,
                                        float *dskip, int b, int n, int m, int c2, int c1, void *stream) {
  if (b < 0 || n < 0 || m < 1 || c2 < 1 || c1 < 0 || !dx0 || !idx || !weight) return GB_EINVAL;
  const long long rows = (long long)b * n;
  if (rows == 0 || (!dknown && !dskip)) return GB_OK;
  const long long threads = rows * (c2 + c1);
  if (threads / CL_TPB > 0x7fffffffLL) return GB_ERANGE;
  hipLaunchKernelGGL(interp_concat_cl_grad_kernel, dim3(blocks_for(threads)), dim3(CL_TPB), 0, as_stream(stream), dx0, idx,
                     weight, dknown, dskip, n, m, c2, c1, rows);
  return check_launch("gb_interp_concat_cl_grad");
}

static bool fin_ok(const GbBnFinalize *fin) {
  return !fin || (fin->gamma && fin->beta && fin->ab && fin->P >= 1 && fin->training == 1 &&
                  (!fin->running_mean == !fin->running_var));
}

extern "C" int gb_col_stats(const float *y, long long P, int C, double *stats, const GbBnFinalize *fin, void *stream) {
  if (P < 0 || C < 1 || !y || !stats || !fin_ok(fin)) return GB_EINVAL;
  if (P == 0) return fin ? GB_EINVAL : GB_OK;
  static_assert(sizeof(BnFinalize) == sizeof(GbBnFinalize), "device view of GbBnFinalize");
  const int rpb = rows_per_block(P);
  const dim3 grid((unsigned)((P + rpb - 1) / rpb));
  if (C % 4 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0)
    hipLaunchKernelGGL((col_stats_kernel<4>), grid, dim3(CL_TPB), 0, as_stream(stream), y, P, C, rpb, stats, stats + C);
  else
    hipLaunchKernelGGL((col_stats_kernel<1>), grid, dim3(CL_TPB), 0, as_stream(stream), y, P, C, rpb, stats, stats + C);
  const int rc = check_launch("gb_col_stats");
  if (rc != GB_OK || !fin) return rc;
  return gb_bn_finalize(stats, 1, fin->P, C, fin->gamma, fin->beta, fin->eps, fin->momentum, fin->running_mean,
                        fin->running_var, fin->ab, 1, stream);
}

// y (P,C) = the chunk-ordered sum of `chunks` partial products (each P*C floats, back to back in `part`), its BatchNorm
// column sums into stats [2C] (caller-zeroed) and, with fin, the layer's finalisation - one launch for the first two.
// C % 4 == 0 and 16-byte aligned part / y: otherwise GB_EINVAL (gemm_cl.hip then takes the two-launch path).
extern "C" int gb_split_col_stats(const float *part, int chunks, float *y, long long P, int C, double *stats,
                                  const GbBnFinalize *fin, void *stream) {
  if (P < 1 || C < 4 || C % 4 || chunks < 1 || !part || !y || !stats || !fin_ok(fin) ||
      (reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(y)) % 16)
    return GB_EINVAL;
  const int rpb = rows_per_block(P);
  hipLaunchKernelGGL(split_col_stats_kernel, dim3((unsigned)((P + rpb - 1) / rpb)), dim3(CL_TPB), 0, as_stream(stream), part,
                     chunks, P * C, y, P, C, rpb, stats, stats + C);
  const int rc = check_launch("gb_split_col_stats");
  if (rc != GB_OK || !fin) return rc;
  return gb_bn_finalize(stats, 1, fin->P, C, fin->gamma, fin->beta, fin->eps, fin->momentum, fin->running_mean,
                        fin->running_var, fin->ab, 1, stream);
}

extern "C" int gb_bn_finalize(const double *stats, int slots, long long P, int C, const float *gamma, const float *beta,
                              float eps, float momentum, float *running_mean, float *running_var, float *ab,
                              int training, void *stream) {
  if (C < 1 || !gamma || !beta || !ab) return GB_EINVAL;
  if (training && (!stats || P < 1 || slots < 1)) return GB_EINVAL;
  if (!training && (!running_mean || !running_var)) return GB_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, as_stream(stream), stats, slots, P, C, gamma,
                     beta, eps, momentum, running_mean, running_var, ab, training);
  return check_launch("gb_bn_finalize");
}

extern "C" int gb_bn_finalize_lin3(const double *mom, const float *w, long long P, int C, const float *gamma,
                                   const float *beta, float eps, float momentum, float *running_mean,
                                   float *running_var, float *ab, void *stream) {
  if (C < 1 || P < 1 || !mom || !w || !gamma || !beta || !ab) return GB_EINVAL;
  hipLaunchKernelGGL(bn_finalize_lin3_kernel, dim3((C + 127) / 128), dim3(128), 0, as_stream(stream), mom, w, P, C, gamma,
                     beta, eps, momentum, running_mean, running_var, ab);
  return check_launch("gb_bn_finalize_lin3");
}

extern "C" int gb_bn_bwd_reduce(const double *dst, int slots, int C, double *dstats, float *dbeta, float *dgamma,
                                void *stream) {
  if (C < 1 || slots < 1 || !dst || !dbeta || !dgamma) return GB_EINVAL;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((2 * C + 127) / 128), dim3(128), 0, as_stream(stream), dst, slots, C,
                     dstats, dbeta, dgamma);
  return check_launch("gb_bn_bwd_reduce");
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
  const bool vec = C % 4 == 0 && (reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(ab) |
                                  reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(arg)) % 16 == 0;
  if (vec)
    hipLaunchKernelGGL(affine_relu_maxpool_vec_kernel, dim3(blocks_for(R * (C / 4))), dim3(CL_TPB), 0, as_stream(stream),
                       y, ab, out, arg, R, ns, C);
  else
    hipLaunchKernelGGL(affine_relu_maxpool_kernel, dim3(blocks_for(R * C)), dim3(CL_TPB), 0, as_stream(stream), y, ab,
                       out, arg, R, ns, C);
  return check_launch("gb_affine_relu_maxpool");
}

// the layer's gb_bn_bwd_reduce as a second launch of the same call (dbeta != NULL)
static int reduce_after(int rc, const double *dst, int slots, int C, double *total, float *dbeta, float *dgamma,
                        void *stream) {
  if (rc != GB_OK || !dbeta) return rc;
  return gb_bn_bwd_reduce(dst, slots, C, total, dbeta, dgamma, stream);
}

extern "C" int gb_bn_bwd_stats(const float *dout, const float *y, const float *ab, const float *residual,
                               long long P, int C, int relu, double *dstats, float *dbeta, float *dgamma,
                               void *stream) {
  if (P < 0 || C < 1 || !dout || !y || !ab || !dstats || (!dbeta != !dgamma)) return GB_EINVAL;
  if (P == 0) return reduce_after(GB_OK, dstats, 1, C, nullptr, dbeta, dgamma, stream);
  const int rpb = rows_per_block(P);
  const dim3 grid((unsigned)((P + rpb - 1) / rpb));
  const bool vec = C % 4 == 0 && (reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dout) |
                                  reinterpret_cast<uintptr_t>(residual)) % 16 == 0;
  if (vec)
    hipLaunchKernelGGL((bn_bwd_stats_kernel<4>), grid, dim3(CL_TPB), 0, as_stream(stream), dout, y, ab, residual, P,
                       C, rpb, relu, dstats, dstats + C);
  else
    hipLaunchKernelGGL((bn_bwd_stats_kernel<1>), grid, dim3(CL_TPB), 0, as_stream(stream), dout, y, ab, residual, P,
                       C, rpb, relu, dstats, dstats + C);
  return reduce_after(check_launch("gb_bn_bwd_stats"), dstats, 1, C, nullptr, dbeta, dgamma, stream);
}

// dz (P,C) = chunk-ordered sum of `chunks` partial products (back to back in part) + the BatchNorm-backward sums
// dstats [2C] (caller-zeroed) of the ReLU layer whose pre-BN output is y and whose table is ab, in one launch.
// C % 4 == 0, 16-byte aligned part / dz / y (else GB_EINVAL: gemm_cl.hip then takes the two-launch path).
extern "C" int gb_split_bn_bwd_stats(const float *part, int chunks, float *dz, const float *y, const float *ab, long long P,
                                     int C, double *dstats, void *stream) {
  if (P < 1 || C < 4 || C % 4 || chunks < 1 || !part || !dz || !y || !ab || !dstats ||
      (reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(y)) % 16)
    return GB_EINVAL;
  const int rpb = rows_per_block(P);
  hipLaunchKernelGGL(split_bn_bwd_stats_kernel, dim3((unsigned)((P + rpb - 1) / rpb)), dim3(CL_TPB), 0, as_stream(stream),
                     part, chunks, P * C, dz, y, ab, P, C, rpb, dstats, dstats + C);
  return check_launch("gb_split_bn_bwd_stats");
}

static int bn_bwd_apply_impl(const float *dout, const float *y, const float *ab, const float *residual,
                             const double *dstats, long long P, int C, int relu, int training, float *dy, float *dres,
                             float *dbeta, float *dgamma, void *stream) {
  if (P < 0 || C < 1 || !dout || !y || !ab || !dy || (training && !dstats)) return GB_EINVAL;
  if ((!dbeta != !dgamma) || (dbeta && !dstats)) return GB_EINVAL;
  if (P == 0) return dbeta ? gb_bn_bwd_reduce(dstats, 1, C, nullptr, dbeta, dgamma, stream) : GB_OK;
  const bool vec = C % 4 == 0 && (reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dout) |
                                  reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(dy) |
                                  reinterpret_cast<uintptr_t>(dres)) % 16 == 0;
  if (vec && C / 4 <= CL_TPB && reinterpret_cast<uintptr_t>(ab) % 16 == 0 && P >= 1024) {
    const int rpp = CL_TPB / (C / 4);
    long long rpb = (P + 4095) / 4096;  // ~4096 workgroups, each a whole number of 4-row passes
    rpb = (rpb + 4 * rpp - 1) / (4 * rpp) * (4 * rpp);
    hipLaunchKernelGGL(bn_bwd_apply_rows_kernel, dim3((unsigned)((P + rpb - 1) / rpb)), dim3(CL_TPB), 0,
                       as_stream(stream), dout, y, ab, residual, dstats, P, C, relu, training, dy, dres, (int)rpb, dbeta,
                       dgamma);
    return check_launch("gb_bn_bwd_apply");
  }
  if (vec)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<4>), dim3(blocks_for(P * C / 4)), dim3(CL_TPB), 0, as_stream(stream), dout,
                       y, ab, residual, dstats, P, C, relu, training, dy, dres);
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<1>), dim3(blocks_for(P * C)), dim3(CL_TPB), 0, as_stream(stream), dout, y,
                       ab, residual, dstats, P, C, relu, training, dy, dres);
  const int rc = check_launch("gb_bn_bwd_apply");
  if (rc != GB_OK || !dbeta) return rc;
  return gb_bn_bwd_reduce(dstats, 1, C, nullptr, dbeta, dgamma, stream);  // the generic kernels do not emit them
}

extern "C" int gb_bn_bwd_apply(const float *dout, const float *y, const float *ab, const float *residual,
                               const double *dstats, long long P, int C, int relu, int training, float *dy,
                               float *dres, void *stream) {
  return bn_bwd_apply_impl(dout, y, ab, residual, dstats, P, C, relu, training, dy, dres, nullptr, nullptr, stream);
}

// gb_bn_bwd_apply that also emits the layer's parameter gradients dbeta = sum dA, dgamma = sum dA*xhat in fp32 from the
// SAME fp64 sums (dstats: one slot row, i.e. already the totals): saves the gb_bn_bwd_reduce launch whose only work for
// such a layer is that conversion.
extern "C" int gb_bn_bwd_apply_g(const float *dout, const float *y, const float *ab, const float *residual,
                                 const double *dstats, long long P, int C, int relu, int training, float *dy,
                                 float *dres, float *dbeta, float *dgamma, void *stream) {
  if (!dbeta || !dgamma) return GB_EINVAL;
  return bn_bwd_apply_impl(dout, y, ab, residual, dstats, P, C, relu, training, dy, dres, dbeta, dgamma, stream);
}

static bool members_ok(long long R, int D, int C, const void *a, const void *b, const void *c, const void *d) {
  return R >= 0 && D >= 1 && D <= 4 && C >= 16 && C % 4 == 0 && C / 4 <= CL_TPB &&
         (reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
          reinterpret_cast<uintptr_t>(d)) % 16 == 0;
}

extern "C" int gb_affine_relu_maxpool_members(const float *y, const float *ab, const int32_t *row_mem, const int64_t *off,
                                              const int32_t *cnt, float *out, int32_t *arg, long long R, int D, int C,
                                              void *stream) {
  if (!y || !ab || !row_mem || !off || !cnt || !out || !arg || !members_ok(R, D, C, y, ab, out, arg)) return GB_EINVAL;
  if (R == 0) return GB_OK;
  const int gpb = CL_TPB / (C / 4);
  const dim3 grid((unsigned)((R + gpb - 1) / gpb));
#define GB_MP(D_) hipLaunchKernelGGL((affine_relu_maxpool_members_kernel<D_>), grid, dim3(CL_TPB), 0, as_stream(stream), y, \
                                     ab, row_mem, off, cnt, out, arg, R, C)
  if (D == 1) GB_MP(1); else if (D == 2) GB_MP(2); else if (D == 3) GB_MP(3); else GB_MP(4);
#undef GB_MP
  return check_launch("gb_affine_relu_maxpool_members");
}

extern "C" int gb_pool_pairs(const float *pairs, const int64_t *off, const int32_t *cnt, const float *ab,
                             const float *gamma, float *out, float *ystar, long long R, int D, int C, void *stream) {
  if (R < 0 || D < 1 || D > 4 || C < 1 || C > CL_TPB || CL_TPB % C || !pairs || !off || !cnt || !ab || !gamma || !out ||
      !ystar)
    return GB_EINVAL;
  if (R == 0) return GB_OK;
  const int gpb = CL_TPB / C;
  const dim3 grid((unsigned)((R + gpb - 1) / gpb));
#define GB_PP(D_) hipLaunchKernelGGL((pool_pairs_kernel<D_>), grid, dim3(CL_TPB), 0, as_stream(stream), pairs, off, cnt, ab, \
                                     gamma, out, ystar, R, C)
  if (D == 1) GB_PP(1); else if (D == 2) GB_PP(2); else if (D == 3) GB_PP(3); else GB_PP(4);
#undef GB_PP
  return check_launch("gb_pool_pairs");
}

static int apply_members_impl(const float *dout, const float *out, const int32_t *arg, const float *ystar, const float *y,
                              const float *ab, const double *dstats, const float *row_w, const int32_t *row_mem,
                              const int64_t *off, const int32_t *cnt, long long R, int D, int C, long long P_total,
                              int training, float *dy, void *stream) {
  const void *sel = arg ? static_cast<const void *>(arg) : static_cast<const void *>(ystar);
  if (!dout || !out || !sel || !y || !ab || !row_w || !off || !cnt || !dy || (training && !dstats) || P_total < 1 ||
      (!arg && !row_mem) || !members_ok(R, D, C, dout, out, sel, y) ||
      (reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(ab)) % 16)
    return GB_EINVAL;
  if (R == 0) return GB_OK;
  const int gpb = CL_TPB / (C / 4);
  const dim3 grid((unsigned)((R + gpb - 1) / gpb));
  const int32_t *sel32 = static_cast<const int32_t *>(sel);
#define GB_MB(D_)                                                                                                          \
  do {                                                                                                                     \
    if (arg) hipLaunchKernelGGL((bn_bwd_apply_members_kernel<D_, false>), grid, dim3(CL_TPB), 0, as_stream(stream), dout, out, \
                                sel32, y, ab, dstats, row_w, off, cnt, R, C, 1.0 / (double)P_total, training, dy, row_mem);     \
    else hipLaunchKernelGGL((bn_bwd_apply_members_kernel<D_, true>), grid, dim3(CL_TPB), 0, as_stream(stream), dout, out,      \
                            sel32, y, ab, dstats, row_w, off, cnt, R, C, 1.0 / (double)P_total, training, dy, row_mem);         \
  } while (0)
  if (D == 1) GB_MB(1); else if (D == 2) GB_MB(2); else if (D == 3) GB_MB(3); else GB_MB(4);
#undef GB_MB
  return check_launch("gb_bn_bwd_apply_members");
}

extern "C" int gb_bn_bwd_apply_members(const float *dout, const float *out, const int32_t *arg, const float *y,
                                       const float *ab, const double *dstats, const float *row_w, const int64_t *off,
                                       const int32_t *cnt, long long R, int D, int C, long long P_total, int training,
                                       float *dy, void *stream) {
  if (!arg) return GB_EINVAL;
  return apply_members_impl(dout, out, arg, nullptr, y, ab, dstats, row_w, nullptr, off, cnt, R, D, C, P_total, training, dy,
                            stream);
}

// ... with the arg-max row found BY VALUE: crop d's gradient goes to the first member row of the seed whose y equals
// ystar[(r*D + d), c] (gb_pool_pairs after a values-only gb_gemm_fwd_pool); row_mem = the rows' member bits.
extern "C" int gb_bn_bwd_apply_members_v(const float *dout, const float *out, const float *ystar, const float *y,
                                         const float *ab, const double *dstats, const float *row_w,
                                         const int32_t *row_mem, const int64_t *off, const int32_t *cnt, long long R, int D,
                                         int C, long long P_total, int training, float *dy, void *stream) {
  if (!ystar) return GB_EINVAL;
  return apply_members_impl(dout, out, nullptr, ystar, y, ab, dstats, row_w, row_mem, off, cnt, R, D, C, P_total, training,
                            dy, stream);
}

// rows_dev (optional, device): the actual row count when `rows` is only the capacity the caller sized its buffers for
extern "C" int gb_bn_bwd_apply_w(const float *dout, const float *y, const float *ab, const double *dstats,
                                 const float *row_w, long long rows, long long P_total, int C, int training, float *dy,
                                 const long long *rows_dev, void *stream) {
  if (rows < 0 || P_total < 1 || C < 4 || C % 4 != 0 || C / 4 > CL_TPB || !dout || !y || !ab || !row_w || !dy ||
      (training && !dstats))
    return GB_EINVAL;
  if ((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(ab) |
       reinterpret_cast<uintptr_t>(dy)) % 16 || reinterpret_cast<uintptr_t>(rows_dev) % 8)
    return GB_EINVAL;
  if (rows == 0) return GB_OK;
  const int rpp = CL_TPB / (C / 4);
  long long rpb = (rows + 4095) / 4096;
  rpb = (rpb + 4 * rpp - 1) / (4 * rpp) * (4 * rpp);
  hipLaunchKernelGGL(bn_bwd_apply_w_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(CL_TPB), 0, as_stream(stream),
                     dout, y, ab, dstats, row_w, rows, C, 1.0 / (double)P_total, training, dy, (int)rpb, rows_dev);
  return check_launch("gb_bn_bwd_apply_w");
}

extern "C" int gb_bn_bwd_stats_pool(const float *dout, const float *out, const int32_t *arg, const float *y,
                                    const float *ab, long long R, int ns, int C, double *dstats, float *dbeta,
                                    float *dgamma, void *stream) {
  if (R < 0 || ns < 0 || C < 1 || !dout || !out || !arg || !y || !ab || !dstats || (!dbeta != !dgamma))
    return GB_EINVAL;  // ns = 0: arg is an absolute row index
  if (R == 0) return reduce_after(GB_OK, dstats, 1, C, nullptr, dbeta, dgamma, stream);
  const int rpb = rows_per_block(R);
  const dim3 grid((unsigned)((R + rpb - 1) / rpb));
  const bool vec = C % 4 == 0 && (reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(dout)) % 16 == 0;
  if (vec)
    hipLaunchKernelGGL((bn_bwd_stats_pool_kernel<4>), grid, dim3(CL_TPB), 0, as_stream(stream), dout, out, arg, y, ab,
                       R, ns, C, rpb, dstats, dstats + C);
  else
    hipLaunchKernelGGL((bn_bwd_stats_pool_kernel<1>), grid, dim3(CL_TPB), 0, as_stream(stream), dout, out, arg, y, ab,
                       R, ns, C, rpb, dstats, dstats + C);
  return reduce_after(check_launch("gb_bn_bwd_stats_pool"), dstats, 1, C, nullptr, dbeta, dgamma, stream);
}

extern "C" int gb_bn_bwd_apply_pool(const float *dout, const float *out, const int32_t *arg, const float *y,
                                    const float *ab, const double *dstats, long long R, int ns, int C,
                                    int training, float *dy, void *stream) {
  if (R < 0 || ns < 1 || C < 1 || !dout || !out || !arg || !y || !ab || !dy || (training && !dstats)) return GB_EINVAL;
  if (R == 0) return GB_OK;
  const bool vec = C % 4 == 0 && (reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dout) |
                                  reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(dy)) % 16 == 0;
  if (vec)
    hipLaunchKernelGGL((bn_bwd_apply_pool_kernel<4>), dim3(blocks_for(R * (C / 4))), dim3(CL_TPB), 0, as_stream(stream),
                       dout, out, arg, y, ab, dstats, R, ns, C, training, dy);
  else
    hipLaunchKernelGGL((bn_bwd_apply_pool_kernel<1>), dim3(blocks_for(R * C)), dim3(CL_TPB), 0, as_stream(stream), dout,
                       out, arg, y, ab, dstats, R, ns, C, training, dy);
  return check_launch("gb_bn_bwd_apply_pool");
}

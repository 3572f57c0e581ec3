// gather_points / group_points (+ grads), three_nn, three_interpolate (+ grad), knn1 for gfx950.
// Replace PointNet/_ext_src/src/{sampling_gpu.cu:13-61, group_points_gpu.cu:17-101,
// interpolate_gpu.cu:14-159} and their pointnet2_batch/src "_fast" twins, plus the k = 1 use of
// KNN/Pytorch_CUDA_KNN/cuda/knn.cu.
//
// These are HBM-bound index/copy kernels: one thread per output element with the channel loop
// inside (the index is read once, reused for every channel), 16-byte stores where the row length
// allows, fp32 scatter-adds as no-return global_atomic_add_f32 (-munsafe-fp-atomics).
#include "gb_common.h"

namespace gb {

constexpr int TPB = 256;
constexpr int CCHUNK = 8;  // channels per thread

// out[b,c,e] = points[b,c,idx[b,e]]   for e in [0, L): gather (L = m) and group (L = m*nsample)
template <int VEC>
__global__ __launch_bounds__(TPB) void gather_rows_kernel(const float *__restrict__ points,
                                                           const int32_t *__restrict__ idx,
                                                           float *__restrict__ out, int c, int n,
                                                           int L) {
  const int e = (blockIdx.x * TPB + threadIdx.x) * VEC;
  if (e >= L) return;
  const int bi = blockIdx.z;
  const int cbeg = blockIdx.y * CCHUNK;
  const int cend = cbeg + CCHUNK < c ? cbeg + CCHUNK : c;
  const int32_t *ix = idx + (size_t)bi * L + e;
  int32_t id[VEC];
  if constexpr (VEC == 4) {
    const int4 v = *reinterpret_cast<const int4 *>(ix);
    id[0] = v.x; id[1] = v.y; id[2] = v.z; id[3] = v.w;
  } else {
    id[0] = ix[0];
  }
  for (int l = cbeg; l < cend; ++l) {
    const float *src = points + ((size_t)bi * c + l) * n;
    float *dst = out + ((size_t)bi * c + l) * L + e;
    if constexpr (VEC == 4) {
      float4 v;
      v.x = src[id[0]]; v.y = src[id[1]]; v.z = src[id[2]]; v.w = src[id[3]];
      *reinterpret_cast<float4 *>(dst) = v;
    } else {
      dst[0] = src[id[0]];
    }
  }
}

// grad_points[b,c,idx[b,e]] += grad_out[b,c,e]
template <int VEC>
__global__ __launch_bounds__(TPB) void scatter_rows_kernel(const float *__restrict__ grad_out,
                                                            const int32_t *__restrict__ idx,
                                                            float *__restrict__ grad_points, int c,
                                                            int n, int L) {
  const int e = (blockIdx.x * TPB + threadIdx.x) * VEC;
  if (e >= L) return;
  const int bi = blockIdx.z;
  const int cbeg = blockIdx.y * CCHUNK;
  const int cend = cbeg + CCHUNK < c ? cbeg + CCHUNK : c;
  const int32_t *ix = idx + (size_t)bi * L + e;
  int32_t id[VEC];
  if constexpr (VEC == 4) {
    const int4 v = *reinterpret_cast<const int4 *>(ix);
    id[0] = v.x; id[1] = v.y; id[2] = v.z; id[3] = v.w;
  } else {
    id[0] = ix[0];
  }
  for (int l = cbeg; l < cend; ++l) {
    float *dst = grad_points + ((size_t)bi * c + l) * n;
    const float *src = grad_out + ((size_t)bi * c + l) * L + e;
    if constexpr (VEC == 4) {
      const float4 v = *reinterpret_cast<const float4 *>(src);
      atomicAdd(dst + id[0], v.x);
      atomicAdd(dst + id[1], v.y);
      atomicAdd(dst + id[2], v.z);
      atomicAdd(dst + id[3], v.w);
    } else {
      atomicAdd(dst + id[0], src[0]);
    }
  }
}

// grad_points[b,c,idx[b,e]] += grad_out[b,c,e] without global atomics: one workgroup owns CH whole
// rows (b, c0..c0+CH-1) of grad_points in LDS (CH*n floats), streams the L = m*nsample positions
// with coalesced reads, accumulates with LDS float atomics (ds_add_f32) and adds the rows back with
// coalesced stores.  Scattered GLOBAL float atomics run at ~0.08 TB/s on MI355X (64 different rows per
// wave instruction); LDS atomics do not leave the CU.
// Ball-query rows are padded with copies of their first index: when nsample is a power of two <= 64
// a row is an aligned lane segment, the padded lanes' values are summed with a segmented butterfly
// and added once by the row's first lane, so a padded row costs no 60-way same-address conflict.
constexpr int SG_TPB = 512;

template <int CH, int NS_LOG2>  // NS_LOG2 < 0: no row structure assumed
__global__ __launch_bounds__(SG_TPB) void scatter_rows_lds_kernel(const float *__restrict__ grad_out,
                                                                   const int32_t *__restrict__ idx,
                                                                   float *__restrict__ grad_points,
                                                                   int c, int n, int L) {
  extern __shared__ float s_acc[];  // [CH][n]
  const int bi = blockIdx.y;
  const int c0 = blockIdx.x * CH;
  const int nch = c - c0 < CH ? c - c0 : CH;
  for (int t = threadIdx.x; t < CH * n; t += SG_TPB) s_acc[t] = 0.f;
  __syncthreads();
  const int32_t *ix = idx + (size_t)bi * L;
  const float *src = grad_out + ((size_t)bi * c + c0) * L;
  const int Lpad = ceil_div(L, SG_TPB) * SG_TPB;  // keep whole waves active for the shuffles
  for (int e = threadIdx.x; e < Lpad; e += SG_TPB) {
    const bool live = e < L;
    const int id = live ? ix[e] : 0;
    bool pad = false;
    if constexpr (NS_LOG2 >= 0) {
      constexpr int NS = 1 << NS_LOG2;
      const int first = __shfl(id, (threadIdx.x & 63) & ~(NS - 1));
      pad = ((e & (NS - 1)) != 0) && (id == first);
    }
#pragma unroll
    for (int l = 0; l < CH; ++l) {
      if (l < nch) {
        float v = live ? src[(size_t)l * L + e] : 0.f;
        if constexpr (NS_LOG2 >= 0) {
          float extra = pad ? v : 0.f;
#pragma unroll
          for (int off = 1; off < (1 << NS_LOG2); off <<= 1) extra += __shfl_xor(extra, off);
          if ((e & ((1 << NS_LOG2) - 1)) == 0) v += extra;
        }
        if (live && !pad) atomicAdd(&s_acc[l * n + id], v);
      }
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < nch * n; t += SG_TPB) {
    float *dst = grad_points + ((size_t)bi * c + c0) * n;
    dst[t] += s_acc[t];
  }
}

static int launch_gather_rows(const float *points, const int32_t *idx, float *out, int b, int c,
                              int n, long long L, hipStream_t s, const char *what) {
  if (b == 0 || c == 0 || L == 0) return GB_OK;
  if (L > 0x7fffffffLL || b > 65535 || ceil_div(c, CCHUNK) > 65535) return GB_ERANGE;
  const bool vec = (L % 4 == 0) && ((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
  if (vec) {
    dim3 grid(ceil_div((int)(L / 4), TPB), ceil_div(c, CCHUNK), b);
    hipLaunchKernelGGL((gather_rows_kernel<4>), grid, dim3(TPB), 0, s, points, idx, out, c, n, (int)L);
  } else {
    dim3 grid(ceil_div((int)L, TPB), ceil_div(c, CCHUNK), b);
    hipLaunchKernelGGL((gather_rows_kernel<1>), grid, dim3(TPB), 0, s, points, idx, out, c, n, (int)L);
  }
  return check_launch(what);
}

template <int CH>
static void launch_scatter_lds(const float *grad_out, const int32_t *idx, float *grad_points, int b, int c,
                               int n, int L, int nsample, hipStream_t s) {
  dim3 grid(ceil_div(c, CH), b);
  const size_t lds = (size_t)CH * n * sizeof(float);
  int ns_log2 = -1;
  if (nsample >= 2 && nsample <= 64 && (nsample & (nsample - 1)) == 0 && L % nsample == 0)
    for (ns_log2 = 0; (1 << ns_log2) < nsample; ++ns_log2) {}
#define GB_SG(NSL)                                                                                     \
  hipLaunchKernelGGL((scatter_rows_lds_kernel<CH, NSL>), grid, dim3(SG_TPB), lds, s, grad_out, idx,    \
                     grad_points, c, n, L)
  switch (ns_log2) {
    case 1: GB_SG(1); break;
    case 2: GB_SG(2); break;
    case 3: GB_SG(3); break;
    case 4: GB_SG(4); break;
    case 5: GB_SG(5); break;
    case 6: GB_SG(6); break;
    default: GB_SG(-1); break;
  }
#undef GB_SG
}

// nsample = 0: no row structure (gather_grad); otherwise rows of `nsample` consecutive positions
static int launch_scatter_rows(const float *grad_out, const int32_t *idx, float *grad_points, int b,
                               int c, int n, long long L, int nsample, hipStream_t s, const char *what) {
  if (b == 0 || c == 0 || L == 0) return GB_OK;
  if (L > 0x7fffffffLL || b > 65535 || ceil_div(c, CCHUNK) > 65535) return GB_ERANGE;
  // LDS path: CH rows of n floats per workgroup within 64 KiB (2 workgroups per CU)
  if (n <= 16384 && L >= 4096) {
    if (n <= 2048) launch_scatter_lds<8>(grad_out, idx, grad_points, b, c, n, (int)L, nsample, s);
    else if (n <= 4096) launch_scatter_lds<4>(grad_out, idx, grad_points, b, c, n, (int)L, nsample, s);
    else if (n <= 8192) launch_scatter_lds<2>(grad_out, idx, grad_points, b, c, n, (int)L, nsample, s);
    else launch_scatter_lds<1>(grad_out, idx, grad_points, b, c, n, (int)L, nsample, s);
    return check_launch(what);
  }
  const bool vec = (L % 4 == 0) && ((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(grad_out)) % 16 == 0);
  if (vec) {
    dim3 grid(ceil_div((int)(L / 4), TPB), ceil_div(c, CCHUNK), b);
    hipLaunchKernelGGL((scatter_rows_kernel<4>), grid, dim3(TPB), 0, s, grad_out, idx, grad_points, c, n, (int)L);
  } else {
    dim3 grid(ceil_div((int)L, TPB), ceil_div(c, CCHUNK), b);
    hipLaunchKernelGGL((scatter_rows_kernel<1>), grid, dim3(TPB), 0, s, grad_out, idx, grad_points, c, n, (int)L);
  }
  return check_launch(what);
}

// ---- three_nn: one thread per unknown point, known points streamed through LDS ---------------
constexpr int NN_TILE = 1024;

__global__ __launch_bounds__(TPB) void three_nn_kernel(const float *__restrict__ unknown,
                                                        const float *__restrict__ known,
                                                        float *__restrict__ dist2,
                                                        int32_t *__restrict__ idx, int n, int m) {
  __shared__ float s_k[NN_TILE * 3];
  const int bi = blockIdx.y;
  const int j = blockIdx.x * TPB + threadIdx.x;
  const float *kn = known + (size_t)bi * m * 3;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (j < n) {
    const f3 u = reinterpret_cast<const f3 *>(unknown + (size_t)bi * n * 3)[j];
    ux = u.x; uy = u.y; uz = u.z;
  }
  // The reference keeps the bests in double initialised to 1e40 and compares the float distance
  // against them with '<' (interpolate_gpu.cu:32-51).  Every stored best is a float value, so
  // float bests initialised to +inf decide identically, and (float)1e40 == +inf on output.
  float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
  int i1 = 0, i2 = 0, i3 = 0;
  for (int base = 0; base < m; base += NN_TILE) {
    const int cntk = m - base < NN_TILE ? m - base : NN_TILE;
    __syncthreads();
    for (int t = threadIdx.x; t < cntk * 3; t += TPB) s_k[t] = kn[(size_t)base * 3 + t];
    __syncthreads();
    if (j < n) {
      for (int k = 0; k < cntk; ++k) {
        const float dx = ux - s_k[k * 3 + 0], dy = uy - s_k[k * 3 + 1], dz = uz - s_k[k * 3 + 2];
        const float d = ((dx * dx) + (dy * dy)) + (dz * dz);
        const int kk = base + k;
        if (d < b1) {
          b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = kk;
        } else if (d < b2) {
          b3 = b2; i3 = i2; b2 = d; i2 = kk;
        } else if (d < b3) {
          b3 = d; i3 = kk;
        }
      }
    }
  }
  if (j < n) {
    float *od = dist2 + ((size_t)bi * n + j) * 3;
    int32_t *oi = idx + ((size_t)bi * n + j) * 3;
    od[0] = b1; od[1] = b2; od[2] = b3;
    oi[0] = i1; oi[1] = i2; oi[2] = i3;
  }
}

// out[b,c,j] = (p[i1]*w1 + p[i2]*w2) + p[i3]*w3
__global__ __launch_bounds__(TPB) void three_interpolate_kernel(const float *__restrict__ points,
                                                                 const int32_t *__restrict__ idx,
                                                                 const float *__restrict__ weight,
                                                                 float *__restrict__ out, int c, int m,
                                                                 int n) {
  const int j = blockIdx.x * TPB + threadIdx.x;
  if (j >= n) return;
  const int bi = blockIdx.z;
  const int cbeg = blockIdx.y * CCHUNK;
  const int cend = cbeg + CCHUNK < c ? cbeg + CCHUNK : c;
  const int32_t *ix = idx + ((size_t)bi * n + j) * 3;
  const float *w = weight + ((size_t)bi * n + j) * 3;
  const int a0 = ix[0], a1 = ix[1], a2 = ix[2];
  const float w0 = w[0], w1 = w[1], w2 = w[2];
  for (int l = cbeg; l < cend; ++l) {
    const float *src = points + ((size_t)bi * c + l) * m;
    out[((size_t)bi * c + l) * n + j] = ((src[a0] * w0) + (src[a1] * w1)) + (src[a2] * w2);
  }
}

__global__ __launch_bounds__(TPB) void three_interpolate_grad_kernel(
    const float *__restrict__ grad_out, const int32_t *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points, int c, int n, int m) {
  const int j = blockIdx.x * TPB + threadIdx.x;
  if (j >= n) return;
  const int bi = blockIdx.z;
  const int cbeg = blockIdx.y * CCHUNK;
  const int cend = cbeg + CCHUNK < c ? cbeg + CCHUNK : c;
  const int32_t *ix = idx + ((size_t)bi * n + j) * 3;
  const float *w = weight + ((size_t)bi * n + j) * 3;
  const int a0 = ix[0], a1 = ix[1], a2 = ix[2];
  const float w0 = w[0], w1 = w[1], w2 = w[2];
  for (int l = cbeg; l < cend; ++l) {
    float *dst = grad_points + ((size_t)bi * c + l) * m;
    const float g = grad_out[((size_t)bi * c + l) * n + j];
    atomicAdd(dst + a0, g * w0);
    atomicAdd(dst + a1, g * w1);
    atomicAdd(dst + a2, g * w2);
  }
}

// ---- knn1: nearest reference column of each query column, 1-based int64 ----------------------
constexpr int KNN_TILE = 512;
constexpr int KNN_MAXDIM = 8;

__global__ __launch_bounds__(TPB) void knn1_kernel(const float *__restrict__ ref,
                                                    const float *__restrict__ query,
                                                    int64_t *__restrict__ idx, int dim, int nref,
                                                    int nq) {
  __shared__ float s_r[KNN_MAXDIM * KNN_TILE];
  const int bi = blockIdx.y;
  const int q = blockIdx.x * TPB + threadIdx.x;
  const float *r = ref + (size_t)bi * dim * nref;
  const float *qq = query + (size_t)bi * dim * nq;
  float qv[KNN_MAXDIM];
#pragma unroll
  for (int h = 0; h < KNN_MAXDIM; ++h) qv[h] = (h < dim && q < nq) ? qq[(size_t)h * nq + q] : 0.f;
  float best = INFINITY;
  int besti = 0;
  for (int base = 0; base < nref; base += KNN_TILE) {
    const int cntk = nref - base < KNN_TILE ? nref - base : KNN_TILE;
    __syncthreads();
    for (int t = threadIdx.x; t < dim * KNN_TILE; t += TPB) {
      const int h = t / KNN_TILE, k = t % KNN_TILE;
      s_r[t] = k < cntk ? r[(size_t)h * nref + base + k] : 0.f;
    }
    __syncthreads();
    if (q < nq) {
      for (int k = 0; k < cntk; ++k) {
        float d = 0.0f;
#pragma unroll
        for (int h = 0; h < KNN_MAXDIM; ++h)
          if (h < dim) {
            const float t = s_r[h * KNN_TILE + k] - qv[h];
            d = d + (t * t);
          }
        if ((base + k) == 0 || d < best) { best = d; besti = base + k; }
      }
    }
  }
  if (q < nq) idx[(size_t)bi * nq + q] = (int64_t)besti + 1;
}

// k nearest references per query, k <= KNN_MAXK (knn.cu:113-176 / cpu/knn_cpu.cpp:4-55 for any k): a thread keeps its
// query's k best (distance, index) pairs sorted in registers; a candidate enters only with a STRICTLY smaller distance than
// an entry and is placed before the first strictly larger one - equal distances stay in index order, which is the order
// of the reference's stable sorts.  idx (b, k, nq) int64, 1-based, row i = the (i+1)-th nearest.
constexpr int KNN_MAXK = 16;

template <int KK>
__global__ __launch_bounds__(TPB) void knnk_kernel(const float *__restrict__ ref, const float *__restrict__ query,
                                                    int64_t *__restrict__ idx, int dim, int nref, int nq, int k) {
  __shared__ float s_r[KNN_MAXDIM * KNN_TILE];
  const int bi = blockIdx.y;
  const int q = blockIdx.x * TPB + threadIdx.x;
  const float *r = ref + (size_t)bi * dim * nref;
  const float *qq = query + (size_t)bi * dim * nq;
  float qv[KNN_MAXDIM];
#pragma unroll
  for (int h = 0; h < KNN_MAXDIM; ++h) qv[h] = (h < dim && q < nq) ? qq[(size_t)h * nq + q] : 0.f;
  float bd[KK];
  int bidx[KK];
#pragma unroll
  for (int i = 0; i < KK; ++i) { bd[i] = INFINITY; bidx[i] = -1; }
  for (int base = 0; base < nref; base += KNN_TILE) {
    const int cntk = nref - base < KNN_TILE ? nref - base : KNN_TILE;
    __syncthreads();
    for (int t = threadIdx.x; t < dim * KNN_TILE; t += TPB) {
      const int h = t / KNN_TILE, kk = t % KNN_TILE;
      s_r[t] = kk < cntk ? r[(size_t)h * nref + base + kk] : 0.f;
    }
    __syncthreads();
    if (q < nq) {
      for (int kk = 0; kk < cntk; ++kk) {
        float d = 0.0f;
#pragma unroll
        for (int h = 0; h < KNN_MAXDIM; ++h)
          if (h < dim) {
            const float t = s_r[h * KNN_TILE + kk] - qv[h];
            d = d + (t * t);
          }
        // one pass of an insertion: the entry moves down while the slot above it is strictly larger (or still empty);
        // NaN distances never enter (every comparison with them is false), as in the reference's `<`
        float cd = d;
        int ci = base + kk;
        bool carrying = false, done = false;
#pragma unroll
        for (int i = 0; i < KK; ++i) {
          // the candidate takes the first slot whose entry is strictly larger (or empty); from there on every entry
          // moves down by one (the one pushed out of slot k-1 is dropped)
          const bool empty = bidx[i] < 0;
          const bool take = i < k && !done && (carrying || empty || cd < bd[i]);
          const float td = bd[i];
          const int ti = bidx[i];
          bd[i] = take ? cd : td;
          bidx[i] = take ? ci : ti;
          cd = take ? td : cd;
          ci = take ? ti : ci;
          carrying = carrying || take;
          done = done || (take && empty);
        }
      }
    }
  }
  if (q < nq)
#pragma unroll
    for (int i = 0; i < KK; ++i)
      if (i < k) idx[((size_t)bi * k + i) * nq + q] = (int64_t)bidx[i] + 1;
}

// dim == 3 (the only case GraspBalance uses): one WAVE per 4 queries, the 64 lanes split the reference
// columns (coalesced reads, no LDS), then a DPP arg-min with lowest-index tie-break.  The generic kernel
// above gives a query to one thread, which leaves a 300 x 300 problem on 2 workgroups.
constexpr int KNN_QW = 4;

__global__ __launch_bounds__(TPB) void knn1_dim3_kernel(const float *__restrict__ ref,
                                                         const float *__restrict__ query,
                                                         int64_t *__restrict__ idx, int nref, int nq) {
  const int bi = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int q0 = (blockIdx.x * (TPB / 64) + (threadIdx.x >> 6)) * KNN_QW;
  if (q0 >= nq) return;
  const float *r = ref + (size_t)bi * 3 * nref;
  const float *qq = query + (size_t)bi * 3 * nq;
  float qx[KNN_QW], qy[KNN_QW], qz[KNN_QW], best[KNN_QW];
  int besti[KNN_QW];
#pragma unroll
  for (int j = 0; j < KNN_QW; ++j) {
    const int q = q0 + j < nq ? q0 + j : nq - 1;
    qx[j] = qq[q]; qy[j] = qq[nq + q]; qz[j] = qq[2 * (size_t)nq + q];
    best[j] = INFINITY;
    besti[j] = 0x7fffffff;
  }
  for (int k = lane; k < nref; k += 64) {
    const float rx = r[k], ry = r[nref + k], rz = r[2 * (size_t)nref + k];
#pragma unroll
    for (int j = 0; j < KNN_QW; ++j) {
      const float tx = rx - qx[j], ty = ry - qy[j], tz = rz - qz[j];
      const float d = ((tx * tx) + (ty * ty)) + (tz * tz);  // == ((0 + tx^2) + ty^2) + tz^2 of knn_cpu.cpp
      if (d < best[j]) { best[j] = d; besti[j] = k; }
    }
  }
#pragma unroll
  for (int j = 0; j < KNN_QW; ++j) {
    const float dmin = -wave_max_f32(-best[j]);
    const unsigned cand = (best[j] == dmin) ? (unsigned)besti[j] : 0xFFFFFFFFu;
    unsigned win = wave_min_u32(cand);
    if (win >= (unsigned)nref) win = 0;  // nothing finite: the serial scan keeps index 0
    if (lane == 0 && q0 + j < nq) idx[(size_t)bi * nq + q0 + j] = (int64_t)win + 1;
  }
}

// ---- grasp-label gather (label_generation.py:60-99): for every seed r, with o = obj[r] (which
// object's label tensor) and j = pt[r] (which grasp point of that object):
//     out[r, v, :] = src_o[j, view_inds[o, v], :]          W floats per (point, view)
// The reference first permutes EVERY object tensor along the view axis (index_select, 52 MB per
// object) and then selects the seeds' rows; this composes the two index maps and touches only the
// rows that are kept.  One workgroup per seed, 16-byte copies.
// *addr = max(*addr, v) with torch.max's NaN rule (a NaN wins and stays)
__device__ __forceinline__ void atomic_max_nan(float *addr, float v) {
  int *ia = reinterpret_cast<int *>(addr);
  int old = *ia;
  while (true) {
    const float f = __int_as_float(old);
    if (f != f) return;
    if (v == v && !(v > f)) return;
    const int assumed = old;
    old = atomicCAS(ia, assumed, __float_as_int(v));
    if (old == assumed) return;
  }
}

// out_max (optional, caller-initialised to -inf): the maximum of everything gathered - the reference's
// `batch_grasp_label.max()` (label_generation.py:113) without another pass over the (B,Ns,V,A,D) tensor
constexpr int LG_MAX_SRC = 128;
struct LabelSrcTable {  // by value in the kernel arguments: no device-side pointer table to build or cache ...
  const float *p[LG_MAX_SRC];
  const float *const *dev;  // ... unless the caller keeps one (the *_dt entries): a captured launch then follows the table's
                            // CONTENT, i.e. a replayed graph reads whatever label tensors the table names this step
  __device__ __forceinline__ const float *at(int o) const { return dev ? dev[o] : p[o]; }
};

__global__ __launch_bounds__(TPB) void label_gather_kernel(const LabelSrcTable srcs, const int32_t *__restrict__ obj,
                                                            const int32_t *__restrict__ pt,
                                                            const int64_t *__restrict__ view_inds,
                                                            float *__restrict__ out, float *__restrict__ out_max,
                                                            float *__restrict__ out_col, int col_stride, int col_off,
                                                            int V, int W) {
  __shared__ float s_m[TPB / 64];
  const int r = blockIdx.x;
  const int o = obj[r];
  const float *src = srcs.at(o) + (size_t)pt[r] * V * W;
  const int64_t *vi = view_inds + (size_t)o * V;
  float *dst = out + (size_t)r * V * W;
  const int wc = out_col ? W / col_stride : 0;
  float *dcol = out_col ? out_col + (size_t)r * V * wc : nullptr;
  float mx = -INFINITY;
  bool nan = false;
  if ((W & 3) == 0) {
    const int w4 = W >> 2;
    for (int e = threadIdx.x; e < V * w4; e += TPB) {
      const int v = e / w4, q = e % w4;
      const float4 x = reinterpret_cast<const float4 *>(src + (size_t)vi[v] * W)[q];
      if (out) reinterpret_cast<float4 *>(dst)[e] = x;
      if (dcol) {
        const float xv[4] = {x.x, x.y, x.z, x.w};
        if (col_stride == 3 && col_off == 2) {
          // the offsets tensor (angle, depth, width): float4 q = 3a + r holds the widths of grasps 4a (r = 0: .z),
          // 4a + 1 (r = 1: .y), 4a + 2 and 4a + 3 (r = 2: .x, .w) - one constant division per 16 bytes
          const unsigned a = (unsigned)q / 3u, r = (unsigned)q - 3u * a;
          float *dw = dcol + (size_t)v * wc + 4u * a;
          if (r == 0) dw[0] = x.z;
          else if (r == 1) dw[1] = x.y;
          else { dw[2] = x.x; dw[3] = x.w; }
        } else if (col_stride == 3) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const unsigned i = 4u * (unsigned)q + t;
            if (i % 3u == (unsigned)col_off) dcol[(size_t)v * wc + i / 3u] = xv[t];
          }
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int i = 4 * q + t;
            if (i % col_stride == col_off) dcol[(size_t)v * wc + i / col_stride] = xv[t];
          }
        }
      }
      if (out_max) {
        mx = fmaxf(fmaxf(mx, x.x), fmaxf(fmaxf(x.y, x.z), x.w));
        nan |= (x.x != x.x) | (x.y != x.y) | (x.z != x.z) | (x.w != x.w);
      }
    }
  } else {
    for (int e = threadIdx.x; e < V * W; e += TPB) {
      const float x = src[(size_t)vi[e / W] * W + e % W];
      if (out) dst[e] = x;
      if (dcol && (e % W) % col_stride == col_off) dcol[(size_t)(e / W) * wc + (e % W) / col_stride] = x;
      if (out_max) { mx = fmaxf(mx, x); nan |= x != x; }
    }
  }
  if (out_max) {  // block-uniform
    if (nan) mx = NAN;
    // wave: NaN-aware maximum through shuffles, then the waves through LDS
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float other = __shfl_xor(mx, off);
      mx = (mx != mx || other != other) ? NAN : fmaxf(mx, other);
    }
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      float b = s_m[0];
      for (int w = 1; w < TPB / 64; ++w) b = (b != b || s_m[w] != s_m[w]) ? NAN : fmaxf(b, s_m[w]);
      atomic_max_nan(out_max, b);
    }
  }
}

// Score transform + per-view maximum of the gathered grasp labels (reference label_generation.py:112-116:
//   mask = (label > 0) & (width <= max_width); label[mask] = log(u_max / label[mask]); label[~mask] = 0;
//   view_score = max over the A*D grasps of a view)  as one pass instead of eight element-wise launches over
// the (B,Ns,V,A,D) tensor.  A thread owns 4 consecutive grasps (one 16-byte label load, their 12 offset
// floats as three 16-byte loads); the AD/4 threads of a view combine their maxima through LDS.
constexpr int LF_ROWS = 16;  // views per workgroup
__global__ void label_finish_kernel(const float *__restrict__ labels, const float *__restrict__ offsets,
                                    const float *__restrict__ widths, const float *__restrict__ u_max, float max_width, float *__restrict__ out,
                                    float *__restrict__ view_scores, int32_t *__restrict__ view_arg, long long rows,
                                    int ad4) {
  extern __shared__ float s_max[];  // [LF_ROWS][ad4] maxima, then [LF_ROWS][ad4] their positions (as int)
  int *s_pos = reinterpret_cast<int *>(s_max + LF_ROWS * ad4);
  const int rl = threadIdx.x / ad4, q = threadIdx.x % ad4;
  const long long row = (long long)blockIdx.x * LF_ROWS + rl;
  const float um = *u_max;
  float best = -INFINITY;
  int bpos = 4 * q;  // first position of `best` among this thread's 4 grasps
  if (row < rows) {
    const long long e4 = row * ad4 + q;  // index of this thread's group of 4 grasps
    const float4 l = reinterpret_cast<const float4 *>(labels)[e4];
    float wv[4];  // offsets[..., 2] of the 4 grasps
    if (widths) {
      const float4 w4 = reinterpret_cast<const float4 *>(widths)[e4];
      wv[0] = w4.x; wv[1] = w4.y; wv[2] = w4.z; wv[3] = w4.w;
    } else {
      const float4 o0 = reinterpret_cast<const float4 *>(offsets)[3 * e4];
      const float4 o1 = reinterpret_cast<const float4 *>(offsets)[3 * e4 + 1];
      const float4 o2 = reinterpret_cast<const float4 *>(offsets)[3 * e4 + 2];
      wv[0] = o0.z; wv[1] = o1.y; wv[2] = o2.x; wv[3] = o2.w;
    }
    const float lv[4] = {l.x, l.y, l.z, l.w};
    float r[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool m = lv[t] > 0.f && wv[t] <= max_width;
      const float c = lv[t] < 1e-30f ? 1e-30f : lv[t];  // clamp_min(1e-30) (NaN passes through, as in torch)
      r[t] = m ? logf(um / c) : 0.f;
      if (r[t] > best) bpos = 4 * q + t;
      best = fmaxf(best, r[t]);
      if (r[t] != r[t]) best = r[t];  // torch.max propagates NaN
    }
    reinterpret_cast<float4 *>(out)[e4] = make_float4(r[0], r[1], r[2], r[3]);
  }
  s_max[threadIdx.x] = best;
  s_pos[threadIdx.x] = bpos;
  __syncthreads();
  if (threadIdx.x < LF_ROWS) {
    const long long vr = (long long)blockIdx.x * LF_ROWS + threadIdx.x;
    if (vr < rows) {
      float b = -INFINITY;
      int bp = 0;
      bool nan = false;
      for (int i = 0; i < ad4; ++i) {
        const float v = s_max[threadIdx.x * ad4 + i];
        nan |= v != v;
        if (v > b) bp = s_pos[threadIdx.x * ad4 + i];  // first position of the maximum (torch.argmax order)
        b = fmaxf(b, v);
      }
      view_scores[vr] = nan ? NAN : b;
      if (view_arg) view_arg[vr] = bp;
    }
  }
}

// out[r, :] = srcs[obj[r]][pt[r], view_inds[obj[r], row_view[r]], :] - the label rows of ONE view per seed (the view the
// network picked): what label_generation.py:138-157 takes out of the (B,Ns,V,...) tensors, without building them.
__global__ __launch_bounds__(TPB) void label_gather_view_kernel(const LabelSrcTable srcs, const int32_t *__restrict__ obj,
                                                                 const int32_t *__restrict__ pt,
                                                                 const int64_t *__restrict__ view_inds,
                                                                 const int64_t *__restrict__ row_view,
                                                                 float *__restrict__ out, int R, int V, int W) {
  const long long e = (long long)blockIdx.x * TPB + threadIdx.x;
  if (e >= (long long)R * W) return;
  const int r = (int)(e / W), w = (int)(e % W);
  const int o = obj[r];
  const int64_t v = view_inds[(size_t)o * V + row_view[r]];
  out[e] = srcs.at(o)[((size_t)pt[r] * V + v) * W + w];
}

// label_finish_kernel reading the labels and the widths straight from the objects' label / offset tensors (two pointer
// tables): view row (r, v) = seed r's labels of template view v.  Nothing but the per-view maxima and their
// positions is written: the training step needs no more of the (B,Ns,V,A,D) tensors (the "lean" label matching).
struct LabelSrcTable2 {
  const float *lab[LG_MAX_SRC];
  const float *off[LG_MAX_SRC];
  const float *const *dev_lab, *const *dev_off;   // device-side tables (see LabelSrcTable)
};
__global__ void label_scores_kernel(const LabelSrcTable2 srcs, const int32_t *__restrict__ obj,
                                    const int32_t *__restrict__ pt, const int64_t *__restrict__ view_inds,
                                    const float *__restrict__ u_max, float max_width, float *__restrict__ view_scores,
                                    int32_t *__restrict__ view_arg, long long rows, int V, int ad4) {
  extern __shared__ float s_max[];  // [LF_ROWS][ad4] maxima, then [LF_ROWS][ad4] their positions (as int)
  int *s_pos = reinterpret_cast<int *>(s_max + LF_ROWS * ad4);
  const int rl = threadIdx.x / ad4, q = threadIdx.x % ad4;
  const long long row = (long long)blockIdx.x * LF_ROWS + rl;
  const float um = *u_max;
  float best = -INFINITY;
  int bpos = 4 * q;
  if (row < rows) {
    const int r = (int)(row / V), v = (int)(row % V);
    const int o = obj[r];
    const size_t at = ((size_t)pt[r] * V + (size_t)view_inds[(size_t)o * V + v]) * ad4 + q;  // group of 4 grasps
    const float *labp = srcs.dev_lab ? srcs.dev_lab[o] : srcs.lab[o];
    const float *offp = srcs.dev_off ? srcs.dev_off[o] : srcs.off[o];
    const float4 l = reinterpret_cast<const float4 *>(labp)[at];
    const float4 o0 = reinterpret_cast<const float4 *>(offp)[3 * at];
    const float4 o1 = reinterpret_cast<const float4 *>(offp)[3 * at + 1];
    const float4 o2 = reinterpret_cast<const float4 *>(offp)[3 * at + 2];
    const float wv[4] = {o0.z, o1.y, o2.x, o2.w};
    const float lv[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool m = lv[t] > 0.f && wv[t] <= max_width;
      const float c = lv[t] < 1e-30f ? 1e-30f : lv[t];
      const float rr = m ? logf(um / c) : 0.f;
      if (rr > best) bpos = 4 * q + t;
      best = fmaxf(best, rr);
      if (rr != rr) best = rr;
    }
  }
  s_max[threadIdx.x] = best;
  s_pos[threadIdx.x] = bpos;
  __syncthreads();
  if (threadIdx.x < LF_ROWS) {
    const long long vr = (long long)blockIdx.x * LF_ROWS + threadIdx.x;
    if (vr < rows) {
      float b = -INFINITY;
      int bp = 0;
      bool nan = false;
      for (int i = 0; i < ad4; ++i) {
        const float v = s_max[threadIdx.x * ad4 + i];
        nan |= v != v;
        if (v > b) bp = s_pos[threadIdx.x * ad4 + i];
        b = fmaxf(b, v);
      }
      view_scores[vr] = nan ? NAN : b;
      if (view_arg) view_arg[vr] = bp;
    }
  }
}

}  // namespace gb

using namespace gb;

extern "C" int gb_label_finish(const float *labels, const float *offsets, const float *widths, const float *u_max,
                               float max_width, float *out, float *view_scores, int32_t *view_arg, long long rows,
                               int ad, void *stream) {
  if (rows < 0 || ad < 4 || ad % 4 != 0 || ad / 4 * LF_ROWS > 1024 || !labels || (!offsets && !widths) || !u_max ||
      !out || !view_scores)
    return GB_EINVAL;
  if ((reinterpret_cast<uintptr_t>(labels) | reinterpret_cast<uintptr_t>(offsets) | reinterpret_cast<uintptr_t>(widths) |
       reinterpret_cast<uintptr_t>(out)) % 16)
    return GB_EINVAL;
  if (rows == 0) return GB_OK;
  const int ad4 = ad / 4, threads = LF_ROWS * ad4;
  hipLaunchKernelGGL(label_finish_kernel, dim3((unsigned)((rows + LF_ROWS - 1) / LF_ROWS)), dim3(threads),
                     2 * threads * sizeof(float), as_stream(stream), labels, offsets, widths, u_max, max_width, out,
                     view_scores, view_arg, rows, ad4);
  return check_launch("gb_label_finish");
}

static int label_gather_impl(const float *const *srcs, bool dev, int nsrc, const int32_t *obj, const int32_t *pt,
                             const int64_t *view_inds, float *out, float *out_max, float *out_col, int col_stride,
                             int col_off, int R, int V, int W, void *stream) {
  if (R < 0 || V < 1 || W < 1 || nsrc < 1 || !srcs || !obj || !pt || !view_inds || (!out && !out_max && !out_col))
    return GB_EINVAL;  // out may be NULL: only the maximum and / or the column copy are wanted
  if (nsrc > LG_MAX_SRC) return GB_ERANGE;
  if (out_col && (col_stride < 1 || col_off < 0 || col_off >= col_stride || W % col_stride != 0)) return GB_EINVAL;
  if (R == 0) return GB_OK;
  LabelSrcTable tab;
  for (int i = 0; i < LG_MAX_SRC; ++i) tab.p[i] = (!dev && i < nsrc) ? srcs[i] : nullptr;
  tab.dev = dev ? srcs : nullptr;
  hipLaunchKernelGGL(label_gather_kernel, dim3(R), dim3(TPB), 0, as_stream(stream), tab, obj, pt, view_inds, out,
                     out_max, out_col, col_stride, col_off, V, W);
  return check_launch("gb_label_gather");
}

extern "C" int gb_label_gather(const float *const *srcs, int nsrc, const int32_t *obj, const int32_t *pt,
                               const int64_t *view_inds, float *out, float *out_max, float *out_col, int col_stride,
                               int col_off, int R, int V, int W, void *stream) {
  return label_gather_impl(srcs, false, nsrc, obj, pt, view_inds, out, out_max, out_col, col_stride, col_off, R, V, W, stream);
}

extern "C" int gb_label_gather_dt(const float *const *srcs_dev, int nsrc, const int32_t *obj, const int32_t *pt,
                                  const int64_t *view_inds, float *out, float *out_max, float *out_col, int col_stride,
                                  int col_off, int R, int V, int W, void *stream) {
  return label_gather_impl(srcs_dev, true, nsrc, obj, pt, view_inds, out, out_max, out_col, col_stride, col_off, R, V, W, stream);
}

static int label_gather_view_impl(const float *const *srcs, bool dev, int nsrc, const int32_t *obj, const int32_t *pt,
                                  const int64_t *view_inds, const int64_t *row_view, float *out, int R, int V, int W,
                                  void *stream) {
  if (R < 0 || V < 1 || W < 1 || nsrc < 1 || !srcs || !obj || !pt || !view_inds || !row_view || !out) return GB_EINVAL;
  if (nsrc > LG_MAX_SRC) return GB_ERANGE;
  if (R == 0) return GB_OK;
  LabelSrcTable tab;
  for (int i = 0; i < LG_MAX_SRC; ++i) tab.p[i] = (!dev && i < nsrc) ? srcs[i] : nullptr;
  tab.dev = dev ? srcs : nullptr;
  const long long total = (long long)R * W;
  hipLaunchKernelGGL(label_gather_view_kernel, dim3((unsigned)((total + TPB - 1) / TPB)), dim3(TPB), 0, as_stream(stream),
                     tab, obj, pt, view_inds, row_view, out, R, V, W);
  return check_launch("gb_label_gather_view");
}

extern "C" int gb_label_gather_view(const float *const *srcs, int nsrc, const int32_t *obj, const int32_t *pt,
                                    const int64_t *view_inds, const int64_t *row_view, float *out, int R, int V, int W,
                                    void *stream) {
  return label_gather_view_impl(srcs, false, nsrc, obj, pt, view_inds, row_view, out, R, V, W, stream);
}

extern "C" int gb_label_gather_view_dt(const float *const *srcs_dev, int nsrc, const int32_t *obj, const int32_t *pt,
                                       const int64_t *view_inds, const int64_t *row_view, float *out, int R, int V, int W,
                                       void *stream) {
  return label_gather_view_impl(srcs_dev, true, nsrc, obj, pt, view_inds, row_view, out, R, V, W, stream);
}

static int label_scores_impl(const float *const *label_srcs, const float *const *offset_srcs, bool dev, int nsrc,
                             const int32_t *obj, const int32_t *pt, const int64_t *view_inds, const float *u_max,
                             float max_width, float *view_scores, int32_t *view_arg, int R, int V, int ad, void *stream) {
  if (R < 0 || V < 1 || ad < 4 || ad % 4 != 0 || ad / 4 * LF_ROWS > 1024 || nsrc < 1 || !label_srcs || !offset_srcs ||
      !obj || !pt || !view_inds || !u_max || !view_scores)
    return GB_EINVAL;
  if (nsrc > LG_MAX_SRC) return GB_ERANGE;
  if (R == 0) return GB_OK;
  LabelSrcTable2 tab;
  for (int i = 0; i < LG_MAX_SRC; ++i) {
    tab.lab[i] = (!dev && i < nsrc) ? label_srcs[i] : nullptr;
    tab.off[i] = (!dev && i < nsrc) ? offset_srcs[i] : nullptr;
    if (!dev && i < nsrc && (reinterpret_cast<uintptr_t>(tab.lab[i]) | reinterpret_cast<uintptr_t>(tab.off[i])) % 16)
      return GB_EINVAL;   // (device-side tables: the caller vouches for the 16-byte alignment of every tensor)
  }
  tab.dev_lab = dev ? label_srcs : nullptr;
  tab.dev_off = dev ? offset_srcs : nullptr;
  const long long rows = (long long)R * V;
  const int ad4 = ad / 4, threads = LF_ROWS * ad4;
  hipLaunchKernelGGL(label_scores_kernel, dim3((unsigned)((rows + LF_ROWS - 1) / LF_ROWS)), dim3(threads),
                     2 * threads * sizeof(float), as_stream(stream), tab, obj, pt, view_inds, u_max, max_width, view_scores,
                     view_arg, rows, V, ad4);
  return check_launch("gb_label_scores");
}

extern "C" int gb_label_scores(const float *const *label_srcs, const float *const *offset_srcs, int nsrc,
                               const int32_t *obj, const int32_t *pt, const int64_t *view_inds, const float *u_max,
                               float max_width, float *view_scores, int32_t *view_arg, int R, int V, int ad,
                               void *stream) {
  return label_scores_impl(label_srcs, offset_srcs, false, nsrc, obj, pt, view_inds, u_max, max_width, view_scores, view_arg,
                           R, V, ad, stream);
}

extern "C" int gb_label_scores_dt(const float *const *label_srcs_dev, const float *const *offset_srcs_dev, int nsrc,
                                  const int32_t *obj, const int32_t *pt, const int64_t *view_inds, const float *u_max,
                                  float max_width, float *view_scores, int32_t *view_arg, int R, int V, int ad,
                                  void *stream) {
  return label_scores_impl(label_srcs_dev, offset_srcs_dev, true, nsrc, obj, pt, view_inds, u_max, max_width, view_scores,
                           view_arg, R, V, ad, stream);
}

extern "C" int gb_gather(const float *points, const int32_t *idx, float *out, int b, int c, int n,
                         int m, void *stream) {
  if (b < 0 || c < 0 || n < 1 || m < 0 || !points || !idx || !out) return GB_EINVAL;
  return launch_gather_rows(points, idx, out, b, c, n, m, as_stream(stream), "gb_gather");
}

extern "C" int gb_gather_grad(const float *grad_out, const int32_t *idx, float *grad_points, int b,
                              int c, int n, int m, void *stream) {
  if (b < 0 || c < 0 || n < 1 || m < 0 || !grad_out || !idx || !grad_points) return GB_EINVAL;
  return launch_scatter_rows(grad_out, idx, grad_points, b, c, n, m, 0, as_stream(stream), "gb_gather_grad");
}

extern "C" int gb_group(const float *points, const int32_t *idx, float *out, int b, int c, int n,
                        int m, int nsample, void *stream) {
  if (b < 0 || c < 0 || n < 1 || m < 0 || nsample < 0 || !points || !idx || !out) return GB_EINVAL;
  return launch_gather_rows(points, idx, out, b, c, n, (long long)m * nsample, as_stream(stream), "gb_group");
}

extern "C" int gb_group_grad(const float *grad_out, const int32_t *idx, float *grad_points, int b,
                             int c, int n, int m, int nsample, void *stream) {
  if (b < 0 || c < 0 || n < 1 || m < 0 || nsample < 0 || !grad_out || !idx || !grad_points) return GB_EINVAL;
  return launch_scatter_rows(grad_out, idx, grad_points, b, c, n, (long long)m * nsample, nsample,
                             as_stream(stream), "gb_group_grad");
}

extern "C" int gb_three_nn(const float *unknown, const float *known, float *dist2, int32_t *idx,
                           int b, int n, int m, void *stream) {
  if (b < 0 || n < 0 || m < 0 || !unknown || !known || !dist2 || !idx) return GB_EINVAL;
  if (b == 0 || n == 0) return GB_OK;
  if (b > 65535 || (long long)n * 3 > 0x7fffffffLL || (long long)m * 3 > 0x7fffffffLL) return GB_ERANGE;
  hipLaunchKernelGGL(three_nn_kernel, dim3(ceil_div(n, TPB), b), dim3(TPB), 0, as_stream(stream),
                     unknown, known, dist2, idx, n, m);
  return check_launch("gb_three_nn");
}

// The inverse-distance weights PointnetFPModule forms from three_nn's output (pointnet2_modules.py:260-263 + the sqrt of
// pointnet2_utils.py:84): d = sqrt(d2); r = 1 / (d + 1e-8); w = r / ((r0 + r1) + r2) - five torch launches as one pass,
// same operations in the same order (correctly rounded sqrt and divisions, no contraction).
namespace gb {
__global__ void interp_weights_kernel(const float *__restrict__ dist2, float *__restrict__ weight, long long rows) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const float r0 = 1.0f / (sqrtf(dist2[i * 3]) + 1e-8f);
  const float r1 = 1.0f / (sqrtf(dist2[i * 3 + 1]) + 1e-8f);
  const float r2 = 1.0f / (sqrtf(dist2[i * 3 + 2]) + 1e-8f);
  const float norm = (r0 + r1) + r2;
  weight[i * 3] = r0 / norm;
  weight[i * 3 + 1] = r1 / norm;
  weight[i * 3 + 2] = r2 / norm;
}
}  // namespace gb

extern "C" int gb_interp_weights(const float *dist2, float *weight, long long rows, void *stream) {
  using namespace gb;
  if (rows < 0 || !dist2 || !weight) return GB_EINVAL;
  if (rows == 0) return GB_OK;
  if ((rows + 255) / 256 > 0x7fffffffLL) return GB_ERANGE;
  hipLaunchKernelGGL(interp_weights_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, as_stream(stream), dist2,
                     weight, rows);
  return check_launch("gb_interp_weights");
}

extern "C" int gb_three_interpolate(const float *points, const int32_t *idx, const float *weight,
                                    float *out, int b, int c, int m, int n, void *stream) {
  if (b < 0 || c < 0 || m < 1 || n < 0 || !points || !idx || !weight || !out) return GB_EINVAL;
  if (b == 0 || c == 0 || n == 0) return GB_OK;
  if (b > 65535 || ceil_div(c, CCHUNK) > 65535) return GB_ERANGE;
  hipLaunchKernelGGL(three_interpolate_kernel, dim3(ceil_div(n, TPB), ceil_div(c, CCHUNK), b),
                     dim3(TPB), 0, as_stream(stream), points, idx, weight, out, c, m, n);
  return check_launch("gb_three_interpolate");
}

extern "C" int gb_three_interpolate_grad(const float *grad_out, const int32_t *idx,
                                         const float *weight, float *grad_points, int b, int c,
                                         int n, int m, void *stream) {
  if (b < 0 || c < 0 || m < 1 || n < 0 || !grad_out || !idx || !weight || !grad_points) return GB_EINVAL;
  if (b == 0 || c == 0 || n == 0) return GB_OK;
  if (b > 65535 || ceil_div(c, CCHUNK) > 65535) return GB_ERANGE;
  hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(ceil_div(n, TPB), ceil_div(c, CCHUNK), b),
                     dim3(TPB), 0, as_stream(stream), grad_out, idx, weight, grad_points, c, n, m);
  return check_launch("gb_three_interpolate_grad");
}

extern "C" int gb_knn(const float *ref, const float *query, int64_t *idx, int b, int dim, int nref, int nq, int k,
                      void *stream) {
  if (k == 1) return gb_knn1(ref, query, idx, b, dim, nref, nq, stream);
  if (b < 0 || dim < 1 || dim > KNN_MAXDIM || nref < 1 || nq < 0 || k < 1 || k > KNN_MAXK || k > nref || !ref || !query ||
      !idx)
    return GB_EINVAL;
  if (b == 0 || nq == 0) return GB_OK;
  if (b > 65535) return GB_ERANGE;
  const dim3 grid(ceil_div(nq, TPB), b);
  if (k <= 4) hipLaunchKernelGGL(knnk_kernel<4>, grid, dim3(TPB), 0, as_stream(stream), ref, query, idx, dim, nref, nq, k);
  else if (k <= 8) hipLaunchKernelGGL(knnk_kernel<8>, grid, dim3(TPB), 0, as_stream(stream), ref, query, idx, dim, nref, nq, k);
  else hipLaunchKernelGGL(knnk_kernel<16>, grid, dim3(TPB), 0, as_stream(stream), ref, query, idx, dim, nref, nq, k);
  return check_launch("gb_knn");
}

extern "C" int gb_knn1(const float *ref, const float *query, int64_t *idx, int b, int dim, int nref,
                       int nq, void *stream) {
  if (b < 0 || dim < 1 || dim > KNN_MAXDIM || nref < 1 || nq < 0 || !ref || !query || !idx) return GB_EINVAL;
  if (b == 0 || nq == 0) return GB_OK;
  if (b > 65535) return GB_ERANGE;
  if (dim == 3) {
    hipLaunchKernelGGL(knn1_dim3_kernel, dim3(ceil_div(nq, (TPB / 64) * KNN_QW), b), dim3(TPB), 0, as_stream(stream),
                       ref, query, idx, nref, nq);
    return check_launch("gb_knn1");
  }
  hipLaunchKernelGGL(knn1_kernel, dim3(ceil_div(nq, TPB), b), dim3(TPB), 0, as_stream(stream), ref,
                     query, idx, dim, nref, nq);
  return check_launch("gb_knn1");
}

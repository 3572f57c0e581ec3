// LocalAggregation (reference drp.py:32-67: ball-query group -> [dp, fj] -> 1x1 conv -> BN -> ReLU -> max over
// the ns neighbours), evaluated WITHOUT the grouped (B*m*ns, 3+C) tensor.
//
// The 1x1 convolution is linear and the grouping is a gather, so they commute:
//     y[p, c] = sum_k W[c, k] [dp_p, f_idx(p)]_k  =  G[idx(p), c] + dp_p . Wx[c]        G = f Wf^T  (B*n rows)
// with Wx = W[:, :3], Wf = W[:, 3:].  G is ns times smaller than the grouped tensor (4 MB: L2 resident), and
// everything BatchNorm and the backward pass need from the P = B*m*ns rows reduces to per-POINT sums
// (cnt_i = references of point i, D_i = sum of their dp) and 12 moments of dp:
//     sum_p y      = sum_i cnt_i G_i + S.Wx                       sum_p y^2 = sum_i (cnt_i G_i^2 + 2 G_i D_i.Wx) + Wx^T M Wx
//     dG[i, c]     = a_c (Sg[i, c] - cnt_i m1_c - m2_c rstd_c (cnt_i (G[i, c] - mean_c) + D_i.Wx[c]))
//     dWx[c, j]    = a_c (T[c, j] - m1_c S_j - m2_c rstd_c (U[c, j] - mean_c S_j + sum_j' Wx[c, j'] M[j', j]))
// where Sg / T are the max-pool-routed output gradients scattered to the arg-max rows only (R*C values, not P*C),
// m1 = dbeta/P, m2 = dgamma/P, U[c, j] = sum_i G[i, c] D_i[j].  The only pass over P*C values left is the
// max-pool itself, a gather from L2.  Same function as conv -> BN -> ReLU -> max on the grouped tensor, in a
// different (fp32-rounding-level) summation order.

#include "gb_common.h"

namespace gb {

constexpr int LA_TPB = 256;
constexpr int LA_MAX_NS = 64;

// dp of one grouped row: (xyz[b, id] - centre[b, j]) (* scale when mode == 1), as gb_group_concat_cl forms it
__device__ __forceinline__ void la_dp(const float *__restrict__ xyz, const float *__restrict__ centres, int bi, int n,
                                      long long grp, int id, int mode, float scale, float d[3]) {
  const float *p = xyz + ((size_t)bi * n + id) * 3;
  const float *q = centres + grp * 3;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    d[t] = p[t] - q[t];
    if (mode == 1) d[t] = d[t] * scale;
  }
}

// per-point reference counts and dp sums, and the 12 dp moments  mom = [S(3), M(3x3)]  (fp64)
__global__ __launch_bounds__(LA_TPB) void la_point_stats_kernel(const float *__restrict__ xyz,
                                                                 const float *__restrict__ centres,
                                                                 const int32_t *__restrict__ idx, int n, int m, int ns,
                                                                 int mode, float scale, long long total_rows,
                                                                 float *__restrict__ cnt, float *__restrict__ dsum,
                                                                 double *__restrict__ mom) {
  float s[3] = {0.f, 0.f, 0.f}, mm[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const long long p = ((long long)blockIdx.x * 8 + u) * LA_TPB + threadIdx.x;
    if (p < total_rows) {
      const long long grp = p / ns;
      const int bi = (int)(grp / m);
      const int id = idx[p];
      float d[3];
      la_dp(xyz, centres, bi, n, grp, id, mode, scale, d);
      const size_t pt = (size_t)bi * n + id;
      atomicAdd(cnt + pt, 1.f);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        atomicAdd(dsum + pt * 3 + t, d[t]);
        s[t] += d[t];
#pragma unroll
        for (int q = 0; q < 3; ++q) mm[3 * t + q] += d[t] * d[q];
      }
    }
  }
  // the 12 moments are same-address fp64 atomics (they serialise): combine the workgroup's waves in LDS first
  __shared__ double part[LA_TPB / 64][12];
  double v[12];
#pragma unroll
  for (int i = 0; i < 3; ++i) v[i] = (double)s[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) v[3 + i] = (double)mm[i];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v[i] += __shfl_xor(v[i], off);
  }
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int i = 0; i < 12; ++i) part[threadIdx.x >> 6][i] = v[i];
  __syncthreads();
  if (threadIdx.x < 12) {
    double t = 0.0;
    for (int w = 0; w < LA_TPB / 64; ++w) t += part[w][threadIdx.x];
    atomicAdd(mom + threadIdx.x, t);
  }
}

// LDS-accumulating form for n <= 4096 points per cloud: a workgroup owns a slice of one cloud's rows and adds into an
// (n x 4) image of [cnt, dsum] in LDS (ds_add_f32), then flushes the touched entries with coalesced global atomics -
// 4 scattered global atomics per row (2 M of them at the first level, 263 us) become ~n*4 per workgroup.
__global__ __launch_bounds__(LA_TPB) void la_point_stats_lds_kernel(const float *__restrict__ xyz,
                                                                     const float *__restrict__ centres,
                                                                     const int32_t *__restrict__ idx, int n, int m,
                                                                     int ns, int mode, float scale, int rows_per_block,
                                                                     float *__restrict__ cnt, float *__restrict__ dsum,
                                                                     double *__restrict__ mom) {
  extern __shared__ float img[];  // [n][4] = cnt, dsum0..2
  __shared__ double part[LA_TPB / 64][12];
  const int bi = blockIdx.y;
  for (int i = threadIdx.x; i < n * 4; i += LA_TPB) img[i] = 0.f;
  __syncthreads();
  const long long rows = (long long)m * ns;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  float s[3] = {0.f, 0.f, 0.f}, mm[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  double v[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) v[i] = 0.0;
  int since = 0;
  for (long long p = r0 + threadIdx.x; p < r1; p += LA_TPB) {
    const long long grp = (long long)bi * m + p / ns;
    const int id = idx[(size_t)bi * rows + p];
    float d[3];
    la_dp(xyz, centres, bi, n, grp, id, mode, scale, d);
    atomicAdd(&img[id * 4], 1.f);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      atomicAdd(&img[id * 4 + 1 + t], d[t]);
      s[t] += d[t];
#pragma unroll
      for (int q = 0; q < 3; ++q) mm[3 * t + q] += d[t] * d[q];
    }
    if (++since == 8) {  // fp32 partials of at most 8 rows, then fp64 (as the direct kernel)
#pragma unroll
      for (int i = 0; i < 3; ++i) { v[i] += (double)s[i]; s[i] = 0.f; }
#pragma unroll
      for (int i = 0; i < 9; ++i) { v[3 + i] += (double)mm[i]; mm[i] = 0.f; }
      since = 0;
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) v[i] += (double)s[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) v[3 + i] += (double)mm[i];
  __syncthreads();
  for (int i = threadIdx.x; i < n * 4; i += LA_TPB) {
    const float a = img[i];
    if (a != 0.f) {
      const int pt = i >> 2, f = i & 3;
      if (f == 0) atomicAdd(cnt + (size_t)bi * n + pt, a);
      else atomicAdd(dsum + ((size_t)bi * n + pt) * 3 + (f - 1), a);
    }
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v[i] += __shfl_xor(v[i], off);
  }
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int i = 0; i < 12; ++i) part[threadIdx.x >> 6][i] = v[i];
  __syncthreads();
  if (threadIdx.x < 12) {
    double t = 0.0;
    for (int w = 0; w < LA_TPB / 64; ++w) t += part[w][threadIdx.x];
    atomicAdd(mom + threadIdx.x, t);
  }
}

// column sums over the B*n points:  stats = [sum_p y, sum_p y^2](C),  u = [U_0, U_1, U_2](C)
constexpr int LA_RB = 32;
__global__ __launch_bounds__(LA_TPB) void la_col_stats_kernel(const float *__restrict__ G, const float *__restrict__ cnt,
                                                               const float *__restrict__ dsum,
                                                               const float *__restrict__ wx,
                                                               const double *__restrict__ mom, long long rows, int C,
                                                               double *__restrict__ stats, double *__restrict__ u) {
  const long long r0 = (long long)blockIdx.x * LA_RB;
  long long r1 = r0 + LA_RB;
  if (r1 > rows) r1 = rows;
  for (int c = threadIdx.x; c < C; c += LA_TPB) {
    const double w0 = wx[c * 3], w1 = wx[c * 3 + 1], w2 = wx[c * 3 + 2];
    double a1 = 0.0, a2 = 0.0, u0 = 0.0, u1 = 0.0, u2 = 0.0;
    for (long long r = r0; r < r1; ++r) {
      const double g = G[r * C + c], cn = cnt[r];
      const double d0 = dsum[r * 3], d1 = dsum[r * 3 + 1], d2 = dsum[r * 3 + 2];
      a1 += cn * g;
      a2 += cn * g * g + 2.0 * g * (d0 * w0 + d1 * w1 + d2 * w2);
      u0 += g * d0;
      u1 += g * d1;
      u2 += g * d2;
    }
    if (blockIdx.x == 0) {  // the terms that do not depend on G
      const double w[3] = {w0, w1, w2};
      double q = 0.0;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        a1 += mom[i] * w[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) q += w[i] * mom[3 + 3 * i + j] * w[j];
      }
      a2 += q;
    }
    atomicAdd(stats + c, a1);
    atomicAdd(stats + C + c, a2);
    atomicAdd(u + c, u0);
    atomicAdd(u + C + c, u1);
    atomicAdd(u + 2 * C + c, u2);
  }
}

// out[r, c] = max_k relu(a_c y + b_c), y = G[idx(r,k), c] + dp(r,k).Wx[c];  arg = first k attaining it.
// A workgroup stages (id, dp) of its groups' samples in LDS once; a thread owns 4 columns of one group and
// streams that group's ns rows of G (16-byte gathers that hit L2), 8 in flight.
__global__ __launch_bounds__(LA_TPB) void la_pool_kernel(const float *__restrict__ G, const float *__restrict__ xyz,
                                                          const float *__restrict__ centres,
                                                          const int32_t *__restrict__ idx, const float *__restrict__ wx,
                                                          const float *__restrict__ ab, float *__restrict__ out,
                                                          int32_t *__restrict__ arg, int n, int m, int ns, int C,
                                                          int mode, float scale, long long R) {
  extern __shared__ float4 nb[];  // [groups per block][ns] = (id bits, dp0, dp1, dp2)
  const int tpg = C / 4, gpb = LA_TPB / tpg;
  const long long g0 = (long long)blockIdx.x * gpb;
  for (int i = threadIdx.x; i < gpb * ns; i += LA_TPB) {
    const long long r = g0 + i / ns;
    if (r < R) {
      const int bi = (int)(r / m);
      const int id = idx[r * ns + i % ns];
      float d[3];
      la_dp(xyz, centres, bi, n, r, id, mode, scale, d);
      nb[i] = make_float4(__int_as_float(id), d[0], d[1], d[2]);
    }
  }
  __syncthreads();
  const int gl = threadIdx.x / tpg;
  const long long r = g0 + gl;
  if (gl >= gpb || r >= R) return;
  const int c = (threadIdx.x % tpg) * 4;
  const float *Gb = G + (size_t)(r / m) * n * C + c;
  float a[4], b[4], w[4][3], best[4];
  int bk[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    a[t] = ab[c + t];
    b[t] = ab[C + c + t];
    w[t][0] = wx[(c + t) * 3];
    w[t][1] = wx[(c + t) * 3 + 1];
    w[t][2] = wx[(c + t) * 3 + 2];
    best[t] = -INFINITY;
    bk[t] = 0;
  }
  const float4 *mine = nb + gl * ns;
  for (int k0 = 0; k0 < ns; k0 += 8) {
    float4 gv[8], q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k0 + u < ns) {
        q[u] = mine[k0 + u];
        gv[u] = *reinterpret_cast<const float4 *>(Gb + (size_t)__float_as_int(q[u].x) * C);
      }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k0 + u < ns) {
        const float g4[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float y = ((g4[t] + q[u].y * w[t][0]) + q[u].z * w[t][1]) + q[u].w * w[t][2];
          float o = a[t] * y + b[t];
          o = o > 0.f ? o : 0.f;
          if (o > best[t]) { best[t] = o; bk[t] = k0 + u; }
        }
      }
  }
  *reinterpret_cast<float4 *>(out + r * C + c) = make_float4(best[0], best[1], best[2], best[3]);
  *reinterpret_cast<int4 *>(arg + r * C + c) = make_int4(bk[0], bk[1], bk[2], bk[3]);
}

// backward of the pool + ReLU: routes dout to the arg-max rows.  sg[idx, c] += g (float atomics, R*C of them),
// red = fp64 [5][C] += column sums of g, g*xhat, g*dp0, g*dp1, g*dp2
// A thread owns a column for LA_GB / groups rows (groups = 256 / C row groups when C < 256: no idle threads).  Every
// element is a chain of four dependent reads (out / dout / arg -> idx -> xyz, G -> the atomics); the rows are therefore
// taken LA_U at a time with each level of the chain issued for all of them before the next is touched.
// What the launch costs is the scatter itself (round 5, by compiling parts out: 55 - 60 us with it, 10 - 14 us without,
// 11 us when every row is sent to its own point): a lane's atomic goes to the winner point's row of sg, a different
// cache line for almost every lane, and the chip retires ~11 G atomic line transactions per second whether a line
// carries one float or 32 (the dense atomics of the split wgrads run at 1.3 TB/s for the same reason).  Spreading the
// fp64 column sums over slot rows (same-address contention) changed nothing and was removed again.
constexpr int LA_GB = 16;
constexpr int LA_U = 8;
__global__ __launch_bounds__(LA_TPB) void la_pool_bwd_kernel(const float *__restrict__ dout, const float *__restrict__ out,
                                                              const int32_t *__restrict__ arg,
                                                              const float *__restrict__ G, const float *__restrict__ xyz,
                                                              const float *__restrict__ centres,
                                                              const int32_t *__restrict__ idx,
                                                              const float *__restrict__ wx, const float *__restrict__ ab,
                                                              float *__restrict__ sg, double *__restrict__ red, int n,
                                                              int m, int ns, int C, int mode, float scale, long long R) {
  const int lanes_c = C < LA_TPB ? C : LA_TPB;
  int groups = LA_TPB / lanes_c;   // 1, 2, 4 .. row groups of this workgroup
  if (groups > LA_GB) groups = LA_GB;
  while (LA_GB % groups != 0) --groups;
  const int grp = threadIdx.x / lanes_c;
  if (grp >= groups) return;
  const int per = LA_GB / groups;
  const long long ra = (long long)blockIdx.x * LA_GB + (long long)grp * per;
  long long rb = ra + per;
  if (rb > R) rb = R;
  if (ra >= rb) return;
  for (int c = threadIdx.x % lanes_c; c < C; c += lanes_c) {
    const float w0 = wx[c * 3], w1 = wx[c * 3 + 1], w2 = wx[c * 3 + 2];
    const float mean = ab[2 * C + c], rstd = ab[3 * C + c];
    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (long long r = ra; r < rb; r += LA_U) {
      long long rr[LA_U];
      float g[LA_U];
      int ar[LA_U], id[LA_U], bi[LA_U];
#pragma unroll
      for (int u = 0; u < LA_U; ++u) {
        rr[u] = r + u < rb ? r + u : rb - 1;   // rows past the end: the last row again, with g = 0
        const float o = out[rr[u] * C + c];
        const float dv = dout[rr[u] * C + c];
        ar[u] = arg[rr[u] * C + c];
        g[u] = (r + u < rb && o > 0.f) ? dv : 0.f;
      }
#pragma unroll
      for (int u = 0; u < LA_U; ++u) {
        id[u] = idx[rr[u] * ns + ar[u]];
        bi[u] = (int)(rr[u] / m);
      }
      float d[LA_U][3], gv[LA_U];
#pragma unroll
      for (int u = 0; u < LA_U; ++u) {
        la_dp(xyz, centres, bi[u], n, rr[u], id[u], mode, scale, d[u]);
        gv[u] = G[((size_t)bi[u] * n + id[u]) * C + c];
      }
#pragma unroll
      for (int u = 0; u < LA_U; ++u) {
        if (g[u] != 0.f) {
          const float y = ((gv[u] + d[u][0] * w0) + d[u][1] * w1) + d[u][2] * w2;
          const float xhat = (y - mean) * rstd;
          atomicAdd(sg + ((size_t)bi[u] * n + id[u]) * C + c, g[u]);
          acc[0] += g[u];
          acc[1] += g[u] * xhat;
          acc[2] += g[u] * d[u][0];
          acc[3] += g[u] * d[u][1];
          acc[4] += g[u] * d[u][2];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) atomicAdd(red + (size_t)i * C + c, acc[i]);
  }
}

// The same with the scatter pre-aggregated in LDS (round 5).  The launch above is bound by its atomics: a lane's update
// goes to the winner point's row of sg - a different cache line for almost every lane - and the chip retires ~11 G
// atomic line transactions per second whether a line carries one float or 32.  Here a workgroup takes LA_GB rows that are
// NEIGHBOURS IN SPACE (`perm`: gb_fps_row_order of the stage's centres, one launch per stage): their neighbourhoods
// overlap, so their winners are few distinct points.  The winners get a slot each (LDS hash of the point ids), the
// contributions are added into acc[slot][c] by the thread that owns column c (plain read-modify-write: one thread per
// column and row group, the groups take turns), and every touched slot leaves as one DENSE row of atomics: 8 lines per
// point instead of one line per (row, column).  Slots beyond the table's capacity fall back to the direct scatter.
constexpr int LA_AG_ACC_MAX = 32768;   // acc [umax][C] floats: 64 KB (two workgroups per CU) or 128 KB, the host's choice
template <int ROWS>   // rows of a workgroup: 16, or 32 for narrow layers (two row groups: as many rows per thread)
__global__ __launch_bounds__(LA_TPB) void la_pool_bwd_agg_kernel(const float *__restrict__ dout, const float *__restrict__ out,
                                                                  const int32_t *__restrict__ arg,
                                                                  const float *__restrict__ G, const float *__restrict__ xyz,
                                                                  const float *__restrict__ centres,
                                                                  const int32_t *__restrict__ idx,
                                                                  const float *__restrict__ wx, const float *__restrict__ ab,
                                                                  const int32_t *__restrict__ perm,
                                                                  float *__restrict__ sg, double *__restrict__ red, int n,
                                                                  int m, int ns, int C, int mode, float scale,
                                                                  int acc_floats) {
  extern __shared__ float la_lds[];
  constexpr int LA_AG_HS = ROWS * 64;   // hash slots: >= ROWS * LA_MAX_NS entries can be distinct
  constexpr int HS_SHIFT = ROWS == 32 ? 21 : 22;   // 32 - log2(LA_AG_HS)
  static_assert(ROWS == 16 || ROWS == 32, "hash shift");
  const int umax = acc_floats / C;
  float *acc = la_lds;                                              // [umax][C]
  int *hkey = reinterpret_cast<int *>(la_lds + acc_floats);         // [LA_AG_HS] point id or -1
  int *hval = hkey + LA_AG_HS;                                      // [LA_AG_HS] slot of that point
  int *ids = hval + LA_AG_HS;                                       // [umax] point id of a slot
  int *counter = ids + umax;
  const int t = threadIdx.x;
  for (int i = t; i < LA_AG_HS; i += LA_TPB) hkey[i] = -1;
  if (t == 0) *counter = 0;
  const int lanes_c = C < LA_TPB ? C : LA_TPB;   // (C <= LA_TPB here: the host checks)
  int groups = LA_TPB / lanes_c;
  if (groups > ROWS / LA_U) groups = ROWS / LA_U;   // (a thread takes whole batches of LA_U rows)
  while (ROWS % groups != 0) --groups;
  const int grp = t / lanes_c, c = t % lanes_c;
  const bool worker = grp < groups;
  constexpr int RMAX = 16;      // rows of one thread at most
  const int per = ROWS / groups;   // (<= RMAX: the host pairs ROWS = 32 with two or more row groups)
  // the rows of this workgroup: ROWS consecutive entries of the cloud's spatial order (m % ROWS == 0: one cloud)
  const long long e0 = (long long)blockIdx.x * ROWS;
  const int bi = (int)(e0 / m);
  float gsave[RMAX];
  int idsave[RMAX];
#pragma unroll
  for (int i = 0; i < RMAX; ++i) { gsave[i] = 0.f; idsave[i] = 0; }
  __syncthreads();
  if (worker) {
    const float w0 = wx[c * 3], w1 = wx[c * 3 + 1], w2 = wx[c * 3 + 2];
    const float mean = ab[2 * C + c], rstd = ab[3 * C + c];
    double accr[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i0 = 0; i0 < RMAX; i0 += LA_U) {
      if (i0 < per) {
        long long rr[LA_U];
        float g[LA_U];
        int ar[LA_U], id[LA_U];
#pragma unroll
        for (int u = 0; u < LA_U; ++u) {
          const bool live = i0 + u < per;
          rr[u] = (long long)bi * m + perm[e0 + grp * per + (live ? i0 + u : 0)];
          const float o = out[rr[u] * C + c];
          const float dv = dout[rr[u] * C + c];
          ar[u] = arg[rr[u] * C + c];
          g[u] = (live && o > 0.f) ? dv : 0.f;
        }
#pragma unroll
        for (int u = 0; u < LA_U; ++u) id[u] = idx[rr[u] * ns + ar[u]];
        float d[LA_U][3], gv[LA_U];
#pragma unroll
        for (int u = 0; u < LA_U; ++u) {
          la_dp(xyz, centres, bi, n, rr[u], id[u], mode, scale, d[u]);
          gv[u] = G[((size_t)bi * n + id[u]) * C + c];
        }
#pragma unroll
        for (int u = 0; u < LA_U; ++u) {
          gsave[i0 + u] = g[u];
          idsave[i0 + u] = id[u];
          if (g[u] != 0.f) {
            const float y = ((gv[u] + d[u][0] * w0) + d[u][1] * w1) + d[u][2] * w2;
            const float xhat = (y - mean) * rstd;
            accr[0] += g[u];
            accr[1] += g[u] * xhat;
            accr[2] += g[u] * d[u][0];
            accr[3] += g[u] * d[u][1];
            accr[4] += g[u] * d[u][2];
            // claim a hash slot for the winner (keys only: the slot numbers are dealt once all keys are in)
            unsigned hsl = ((unsigned)id[u] * 2654435761u) >> HS_SHIFT;
            while (true) {
              const int old = atomicCAS(hkey + hsl, -1, id[u]);
              if (old == -1 || old == id[u]) break;
              hsl = (hsl + 1) & (LA_AG_HS - 1);
            }
          }
        }
      }
    }
    // the column sums: one atomic per column and WORKGROUP (same-address fp64 atomics serialise, ~47 ns each: with the
    // scatter out of the way 1024 contributions per address were the first stage's whole launch) - the row groups meet in
    // LDS first (acc is not in use yet)
    double *gsum = reinterpret_cast<double *>(acc);   // [groups][5][C]
#pragma unroll
    for (int i = 0; i < 5; ++i) gsum[((size_t)grp * 5 + i) * C + c] = accr[i];
  }
  __syncthreads();
  if (worker && grp == 0) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      double v = 0.0;
      for (int gi = 0; gi < groups; ++gi) v += reinterpret_cast<double *>(acc)[((size_t)gi * 5 + i) * C + c];
      atomicAdd(red + (size_t)i * C + c, v);
    }
  }
  for (int i = t; i < LA_AG_HS; i += LA_TPB)
    if (hkey[i] != -1) {
      const int tk = atomicAdd(counter, 1);
      hval[i] = tk;
      if (tk < umax) ids[tk] = hkey[i];
    }
  __syncthreads();
  const int used = *counter < umax ? *counter : umax;
  for (int i = t; i < used * C; i += LA_TPB) acc[i] = 0.f;
  __syncthreads();
  for (int turn = 0; turn < groups; ++turn) {
    if (worker && grp == turn) {
#pragma unroll
      for (int i = 0; i < RMAX; ++i) {
        if (i < per && gsave[i] != 0.f) {
          unsigned hsl = ((unsigned)idsave[i] * 2654435761u) >> HS_SHIFT;
          while (hkey[hsl] != idsave[i]) hsl = (hsl + 1) & (LA_AG_HS - 1);
          const int tk = hval[hsl];
          if (tk < umax) acc[tk * C + c] += gsave[i];
          else atomicAdd(sg + ((size_t)bi * n + idsave[i]) * C + c, gsave[i]);   // (table full: the direct scatter)
        }
      }
    }
    __syncthreads();
  }
  for (int i = t; i < used * C; i += LA_TPB) {
    const float v = acc[i];
    if (v != 0.f) atomicAdd(sg + ((size_t)bi * n + ids[i / C]) * C + (i % C), v);
  }
}

// dG[i, c] = a (sg - cnt m1 - m2 rstd (cnt (G - mean) + D.Wx))      (training: m1 = dbeta/P, m2 = dgamma/P; eval: 0)
__global__ __launch_bounds__(LA_TPB) void la_point_grad_kernel(const float *__restrict__ sg, const float *__restrict__ G,
                                                                const float *__restrict__ cnt,
                                                                const float *__restrict__ dsum,
                                                                const float *__restrict__ wx, const float *__restrict__ ab,
                                                                const double *__restrict__ red, double invP,
                                                                long long rows, int C, int training,
                                                                float *__restrict__ dG) {
  const long long e = (long long)blockIdx.x * LA_TPB + threadIdx.x;
  if (e >= rows * C) return;
  const long long i = e / C;
  const int c = (int)(e % C);
  const float a = ab[c];
  float v = sg[e];
  if (training) {
    const float m1 = (float)(red[c] * invP), m2 = (float)(red[C + c] * invP);
    const float cn = cnt[i];
    const float dw = (dsum[i * 3] * wx[c * 3] + dsum[i * 3 + 1] * wx[c * 3 + 1]) + dsum[i * 3 + 2] * wx[c * 3 + 2];
    v = v - cn * m1 - (m2 * ab[3 * C + c]) * (cn * (G[e] - ab[2 * C + c]) + dw);
  }
  dG[e] = a * v;
}

// dWx[c, j] = a (T_j - m1 S_j - m2 rstd (U_j - mean S_j + sum_j' Wx[c, j'] M[j', j]))
__global__ void la_wx_grad_kernel(const double *__restrict__ red, const double *__restrict__ u,
                                  const double *__restrict__ mom, const float *__restrict__ wx,
                                  const float *__restrict__ ab, double invP, int C, int training,
                                  float *__restrict__ dwx, float *__restrict__ dbeta, float *__restrict__ dgamma,
                                  int slots, int ldw) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  // red: `slots` rows of [5][C] partial sums (the dgrad epilogue's slot rows), added here in slot order
  double rs[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int sl = 0; sl < slots; ++sl)
#pragma unroll
    for (int j = 0; j < 5; ++j) rs[j] += red[((size_t)sl * 5 + j) * C + c];
  if (dbeta) {  // gb_la_wx_grad_g: the BatchNorm parameter gradients are the first two sums, converted
    dbeta[c] = (float)rs[0];
    dgamma[c] = (float)rs[1];
  }
  const double a = ab[c], mean = ab[2 * C + c], rstd = ab[3 * C + c];
  const double m1 = training ? rs[0] * invP : 0.0, m2 = training ? rs[1] * invP : 0.0;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    double wm = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q) wm += (double)wx[c * 3 + q] * mom[3 + 3 * q + j];
    const double v = rs[2 + j] - m1 * mom[j] - m2 * rstd * (u[(size_t)j * C + c] - mean * mom[j] + wm);
    dwx[(size_t)c * ldw + j] = (float)(a * v);   // ldw = 3, or the pitch of a joined (C, 3 + Cf) gradient
  }
}

}  // namespace gb

using namespace gb;

static bool la_geom_ok(int b, int n, int m, int ns) { return b >= 0 && n >= 1 && m >= 0 && ns >= 1; }

extern "C" int gb_la_point_stats(const float *xyz, const float *centres, const int32_t *idx, int b, int n, int m,
                                 int ns, int mode, float scale, float *cnt, float *dsum, double *mom, void *stream) {
  if (!la_geom_ok(b, n, m, ns) || !xyz || !centres || !idx || !cnt || !dsum || !mom || (mode != 0 && mode != 1))
    return GB_EINVAL;
  const long long rows = (long long)b * m * ns;
  if (rows == 0) return GB_OK;
  if (n <= 4096 && b <= 65535 && rows / b >= 8192) {
    const long long brows = (long long)m * ns;
    long long slices = 64 / b;  // ~64 workgroups in all: every extra slice repeats the flush of the n x 4 image
    if (slices < 1) slices = 1;
    if (slices > brows / 2048) slices = brows / 2048;
    const long long rpb = (brows + slices - 1) / slices;
    slices = (brows + rpb - 1) / rpb;
    hipLaunchKernelGGL(la_point_stats_lds_kernel, dim3((unsigned)slices, (unsigned)b), dim3(LA_TPB),
                       (size_t)n * 4 * sizeof(float), as_stream(stream), xyz, centres, idx, n, m, ns, mode, scale, (int)rpb,
                       cnt, dsum, mom);
    return check_launch("gb_la_point_stats");
  }
  const long long blocks = (rows + 8 * LA_TPB - 1) / (8 * LA_TPB);
  if (blocks > 0x7fffffffLL) return GB_ERANGE;
  hipLaunchKernelGGL(la_point_stats_kernel, dim3((unsigned)blocks), dim3(LA_TPB), 0, as_stream(stream), xyz, centres,
                     idx, n, m, ns, mode, scale, rows, cnt, dsum, mom);
  return check_launch("gb_la_point_stats");
}

// W (N, 3 + C) <-> [Wx (N,3), Wf (N,C)]: the two column blocks of the LocalAggregation convolution weight as separate
// dense matrices (what the kernels above and the GEMMs take) in one launch, and their gradients back into one.
__global__ __launch_bounds__(LA_TPB) void la_split_join_kernel(float *__restrict__ w, float *__restrict__ wx,
                                                                float *__restrict__ wf, int N, int C, int join) {
  const long long e = (long long)blockIdx.x * LA_TPB + threadIdx.x;
  if (e >= (long long)N * (3 + C)) return;
  const int n = (int)(e / (3 + C)), c = (int)(e % (3 + C));
  float *part = c < 3 ? wx + n * 3 + c : wf + (size_t)n * C + (c - 3);
  if (join) w[e] = *part;
  else *part = w[e];
}

extern "C" int gb_la_split_w(const float *w, float *wx, float *wf, int N, int C, void *stream) {
  if (N < 1 || C < 1 || !w || !wx || !wf) return GB_EINVAL;
  const long long total = (long long)N * (3 + C);
  hipLaunchKernelGGL(la_split_join_kernel, dim3((unsigned)((total + LA_TPB - 1) / LA_TPB)), dim3(LA_TPB), 0,
                     as_stream(stream), const_cast<float *>(w), wx, wf, N, C, 0);
  return check_launch("gb_la_split_w");
}

extern "C" int gb_la_join_w(const float *wx, const float *wf, float *w, int N, int C, void *stream) {
  if (N < 1 || C < 1 || !w || !wx || !wf) return GB_EINVAL;
  const long long total = (long long)N * (3 + C);
  hipLaunchKernelGGL(la_split_join_kernel, dim3((unsigned)((total + LA_TPB - 1) / LA_TPB)), dim3(LA_TPB), 0,
                     as_stream(stream), w, const_cast<float *>(wx), const_cast<float *>(wf), N, C, 1);
  return check_launch("gb_la_join_w");
}

extern "C" int gb_la_col_stats(const float *G, const float *cnt, const float *dsum, const float *wx,
                               const double *mom, long long rows, int C, double *stats, double *u,
                               const GbBnFinalize *fin, void *stream) {
  if (rows < 0 || C < 1 || !G || !cnt || !dsum || !wx || !mom || !stats || !u) return GB_EINVAL;
  if (fin && (!fin->gamma || !fin->beta || !fin->ab || fin->P < 1 || fin->training != 1)) return GB_EINVAL;
  if (rows == 0) return fin ? GB_EINVAL : GB_OK;
  hipLaunchKernelGGL(la_col_stats_kernel, dim3((unsigned)((rows + LA_RB - 1) / LA_RB)), dim3(LA_TPB), 0,
                     as_stream(stream), G, cnt, dsum, wx, mom, rows, C, stats, u);
  const int rc = check_launch("gb_la_col_stats");
  if (rc != GB_OK || !fin) return rc;
  return gb_bn_finalize(stats, 1, fin->P, C, fin->gamma, fin->beta, fin->eps, fin->momentum, fin->running_mean,
                        fin->running_var, fin->ab, 1, stream);
}

extern "C" int gb_la_pool(const float *G, const float *xyz, const float *centres, const int32_t *idx, const float *wx,
                          const float *ab, float *out, int32_t *arg, int b, int n, int m, int ns, int C, int mode,
                          float scale, void *stream) {
  if (!la_geom_ok(b, n, m, ns) || !G || !xyz || !centres || !idx || !wx || !ab || !out || !arg) return GB_EINVAL;
  if (C < 16 || C % 4 != 0 || C / 4 > LA_TPB || ns > LA_MAX_NS || (mode != 0 && mode != 1)) return GB_EINVAL;
  if ((reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(arg)) % 16 != 0)
    return GB_EINVAL;
  const long long R = (long long)b * m;
  if (R == 0) return GB_OK;
  const int gpb = LA_TPB / (C / 4);
  hipLaunchKernelGGL(la_pool_kernel, dim3((unsigned)((R + gpb - 1) / gpb)), dim3(LA_TPB),
                     (size_t)gpb * ns * sizeof(float4), as_stream(stream), G, xyz, centres, idx, wx, ab, out, arg, n, m,
                     ns, C, mode, scale, R);
  return check_launch("gb_la_pool");
}

extern "C" int gb_la_pool_bwd(const float *dout, const float *out, const int32_t *arg, const float *G,
                              const float *xyz, const float *centres, const int32_t *idx, const float *wx,
                              const float *ab, float *sg, double *red, int b, int n, int m, int ns, int C, int mode,
                              float scale, void *stream) {
  if (!la_geom_ok(b, n, m, ns) || !dout || !out || !arg || !G || !xyz || !centres || !idx || !wx || !ab || !sg || !red)
    return GB_EINVAL;
  if (C < 1 || (mode != 0 && mode != 1)) return GB_EINVAL;
  const long long R = (long long)b * m;
  if (R == 0) return GB_OK;
  hipLaunchKernelGGL(la_pool_bwd_kernel, dim3((unsigned)((R + LA_GB - 1) / LA_GB)), dim3(LA_TPB), 0, as_stream(stream),
                     dout, out, arg, G, xyz, centres, idx, wx, ab, sg, red, n, m, ns, C, mode, scale, R);
  return check_launch("gb_la_pool_bwd");
}

extern "C" int gb_la_pool_bwd_perm(const float *dout, const float *out, const int32_t *arg, const float *G,
                                   const float *xyz, const float *centres, const int32_t *idx, const float *wx,
                                   const float *ab, const int32_t *perm, float *sg, double *red, int b, int n, int m,
                                   int ns, int C, int mode, float scale, void *stream) {
  const int rows_wg = (C <= LA_TPB / 2) ? 32 : 16;   // narrow layers: two row groups, 32 rows - half as many workgroups
  if (!perm || C > LA_TPB || LA_TPB % C != 0 || m % rows_wg != 0 || C < 16)   // (what the LDS form needs; else the direct scatter)
    return gb_la_pool_bwd(dout, out, arg, G, xyz, centres, idx, wx, ab, sg, red, b, n, m, ns, C, mode, scale, stream);
  if (!la_geom_ok(b, n, m, ns) || !dout || !out || !arg || !G || !xyz || !centres || !idx || !wx || !ab || !sg || !red)
    return GB_EINVAL;
  if (mode != 0 && mode != 1) return GB_EINVAL;
  const long long R = (long long)b * m;
  if (R == 0) return GB_OK;
  static std::atomic<unsigned long long> attr_set16{0}, attr_set32{0};
  allow_dynamic_lds(la_pool_bwd_agg_kernel<16>, 160 * 1024, attr_set16);
  allow_dynamic_lds(la_pool_bwd_agg_kernel<32>, 160 * 1024, attr_set32);
  // slots for the distinct winners of 16 rows: ns = 64 neighbourhoods (the first stage) need more than 128 of them
  const int acc_floats = (ns > 32 || C <= 64) ? LA_AG_ACC_MAX : LA_AG_ACC_MAX / 2;
  const size_t lds = (size_t)acc_floats * sizeof(float) + (2 * rows_wg * 64 + acc_floats / C + 4) * sizeof(int);
  if (rows_wg == 32)
    hipLaunchKernelGGL(la_pool_bwd_agg_kernel<32>, dim3((unsigned)(R / 32)), dim3(LA_TPB), lds, as_stream(stream), dout, out,
                       arg, G, xyz, centres, idx, wx, ab, perm, sg, red, n, m, ns, C, mode, scale, acc_floats);
  else
    hipLaunchKernelGGL(la_pool_bwd_agg_kernel<16>, dim3((unsigned)(R / 16)), dim3(LA_TPB), lds, as_stream(stream), dout, out,
                       arg, G, xyz, centres, idx, wx, ab, perm, sg, red, n, m, ns, C, mode, scale, acc_floats);
  return check_launch("gb_la_pool_bwd_perm");
}

extern "C" int gb_la_point_grad(const float *sg, const float *G, const float *cnt, const float *dsum, const float *wx,
                                const float *ab, const double *red, long long P, long long rows, int C, int training,
                                float *dG, void *stream) {
  if (rows < 0 || C < 1 || P < 1 || !sg || !G || !cnt || !dsum || !wx || !ab || !red || !dG) return GB_EINVAL;
  if (rows == 0) return GB_OK;
  const long long blocks = (rows * C + LA_TPB - 1) / LA_TPB;
  if (blocks > 0x7fffffffLL) return GB_ERANGE;
  hipLaunchKernelGGL(la_point_grad_kernel, dim3((unsigned)blocks), dim3(LA_TPB), 0, as_stream(stream), sg, G, cnt, dsum,
                     wx, ab, red, 1.0 / (double)P, rows, C, training, dG);
  return check_launch("gb_la_point_grad");
}

extern "C" int gb_la_wx_grad(const double *red, const double *u, const double *mom, const float *wx, const float *ab,
                             long long P, int C, int training, float *dwx, void *stream) {
  if (C < 1 || P < 1 || !red || !u || !mom || !wx || !ab || !dwx) return GB_EINVAL;
  hipLaunchKernelGGL(la_wx_grad_kernel, dim3((C + 127) / 128), dim3(128), 0, as_stream(stream), red, u, mom, wx, ab,
                     1.0 / (double)P, C, training, dwx, nullptr, nullptr, 1, 3);
  return check_launch("gb_la_wx_grad");
}

// gb_la_wx_grad that also writes the BatchNorm parameter gradients dbeta = red[0:C], dgamma = red[C:2C] in fp32 (what a
// one-slot gb_bn_bwd_reduce launch would do); red may be `slots` rows of [5][C] partial sums (added in slot order)
extern "C" int gb_la_wx_grad_g(const double *red, int slots, const double *u, const double *mom, const float *wx,
                               const float *ab, long long P, int C, int training, float *dwx, float *dbeta,
                               float *dgamma, void *stream) {
  if (C < 1 || P < 1 || slots < 1 || !red || !u || !mom || !wx || !ab || !dwx || !dbeta || !dgamma) return GB_EINVAL;
  hipLaunchKernelGGL(la_wx_grad_kernel, dim3((C + 127) / 128), dim3(128), 0, as_stream(stream), red, u, mom, wx, ab,
                     1.0 / (double)P, C, training, dwx, dbeta, dgamma, slots, 3);
  return check_launch("gb_la_wx_grad_g");
}

// gb_la_wx_grad_g writing dwx into a wider matrix: row c of the gradient starts at dwx + c * ldw (ldw >= 3) - the xyz
// columns of an aggregation conv's joined (C, 3 + Cf) weight gradient, whose feature columns a (grouped) weight-gradient
// product adds into; no separate join or strided copy launch (round 6)
extern "C" int gb_la_wx_grad_gs(const double *red, int slots, const double *u, const double *mom, const float *wx,
                                const float *ab, long long P, int C, int training, float *dwx, int ldw, float *dbeta,
                                float *dgamma, void *stream) {
  if (C < 1 || P < 1 || slots < 1 || ldw < 3 || !red || !u || !mom || !wx || !ab || !dwx || !dbeta || !dgamma) return GB_EINVAL;
  hipLaunchKernelGGL(la_wx_grad_kernel, dim3((C + 127) / 128), dim3(128), 0, as_stream(stream), red, u, mom, wx, ab,
                     1.0 / (double)P, C, training, dwx, dbeta, dgamma, slots, ldw);
  return check_launch("gb_la_wx_grad_gs");
}

// Row de-duplication for the cylinder crops of GraspPoseStage2 (reference modules.py:99-124: CloudCrop /
// GraspWidthGrouping run ONE SharedMLP over the points of D nested cylinders per seed - same radius, same rotation,
// hmax = 0.01..0.04).  A point that lies in several of a seed's cylinders yields the same MLP input row (and so the
// same output) in each of them: on the bench clouds only 32-38 % of the D*ns rows of a seed are distinct.  These
// kernels build, per seed, the list of DISTINCT points with a multiplicity (how many of the seed's D*ns slots hold
// the point - BatchNorm statistics and gradients weight the row by it) and a membership mask (which of the D
// cylinders contain it - the max-pool of cylinder d runs over the rows with bit d).  The MLP then runs on the
// distinct rows only; csrc/mlp_cl.hip's *_members kernels do the pooling.  Same function, ~2.7x fewer rows.
#include "gb_common.h"

namespace gb {

constexpr int CU_MAXW = 256;  // D*ns slots per seed handled by one workgroup

// One workgroup (CU_MAXW threads) per seed (blockIdx.x) of each of the nr query sets (blockIdx.y: the radii of stage 2).
// idx: (nr, D, R, ns) int32.  sorted / meta: (nr, R, W) with W = D*ns, the distinct ids of a seed compacted to the front
// in increasing order; meta = (multiplicity << 8) | member bits; count (nr, R).
__global__ __launch_bounds__(CU_MAXW) void cyl_unique_kernel(const int32_t *__restrict__ idx, int D, long long R, int ns,
                                                             int32_t *__restrict__ sorted, int32_t *__restrict__ meta,
                                                             int32_t *__restrict__ count) {
  __shared__ unsigned key[CU_MAXW];  // (id << 3) | depth ; 0xFFFFFFFF = unused slot
  __shared__ int wsum[CU_MAXW / 64];
  const long long r = blockIdx.x;
  const int t = threadIdx.x, W = D * ns;
  idx += (size_t)blockIdx.y * D * R * ns;
  sorted += (size_t)blockIdx.y * R * W;
  meta += (size_t)blockIdx.y * R * W;
  count += (size_t)blockIdx.y * R;
  unsigned k = 0xFFFFFFFFu;
  if (t < W) {
    const int d = t / ns, s = t % ns;
    k = ((unsigned)idx[((size_t)d * R + r) * ns + s] << 3) | (unsigned)d;
  }
  key[t] = k;
  __syncthreads();
  // bitonic sort of CU_MAXW keys
  for (int size = 2; size <= CU_MAXW; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int partner = t ^ stride;
      if (partner > t) {
        const unsigned a = key[t], b = key[partner];
        const bool up = (t & size) == 0;
        if ((a > b) == up) { key[t] = b; key[partner] = a; }
      }
      __syncthreads();
    }
  const unsigned mine = key[t];
  const bool valid = mine != 0xFFFFFFFFu;
  const unsigned id = mine >> 3;
  const bool head = valid && (t == 0 || (key[t - 1] >> 3) != id);
  int mult = 0;
  unsigned bits = 0;
  if (head)
    for (int u = t; u < CU_MAXW && key[u] != 0xFFFFFFFFu && (key[u] >> 3) == id; ++u) {
      ++mult;
      bits |= 1u << (key[u] & 7u);
    }
  // position among the heads
  const unsigned long long hm = __builtin_amdgcn_ballot_w64(head);
  const int lane = t & 63, wave = t >> 6;
  if (lane == 0) wsum[wave] = __builtin_popcountll(hm);
  __syncthreads();
  int pos = prefix_popc(hm);
  for (int w = 0; w < wave; ++w) pos += wsum[w];
  if (head) {
    sorted[r * W + pos] = (int32_t)id;
    meta[r * W + pos] = (mult << 8) | (int32_t)bits;
  }
  if (t == 0) {
    int total = 0;
    for (int w = 0; w < CU_MAXW / 64; ++w) total += wsum[w];
    count[r] = total;
  }
}

// off (nr, R) int64 = exclusive prefix sums of count (nr, R) along R, total (nr) = the sums: one workgroup per query set.
__global__ __launch_bounds__(1024) void cyl_scan_kernel(const int32_t *__restrict__ count, long long R,
                                                        int64_t *__restrict__ off, long long *__restrict__ total) {
  __shared__ long long wtot[16];
  __shared__ long long carry;
  count += (size_t)blockIdx.x * R;
  off += (size_t)blockIdx.x * R;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) carry = 0;
  __syncthreads();
  for (long long r0 = 0; r0 < R; r0 += 1024) {
    const long long r = r0 + t;
    const long long c = r < R ? count[r] : 0;
    long long incl = c;   // inclusive scan within the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const long long o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    long long base = carry;
    for (int w = 0; w < wave; ++w) base += wtot[w];
    if (r < R) off[r] = base + incl - c;
    __syncthreads();
    if (t == 1023) carry = base + incl;
    __syncthreads();
  }
  if (t == 0) total[blockIdx.x] = carry;
}

// x0 (P_u,3) = (xyz[b, id] - centre[seed]) rotated by the seed's matrix (gb_group_concat_cl mode 2 arithmetic),
// row_w (P_u) = multiplicity (also as uint16 in row_w16, for the GEMM epilogues), row_mem (P_u) = member bits;
// row index = off[seed] + u.  Query set blockIdx.y writes at row stride `cap` of every output; the last seed's workgroup
// also zeroes row_w16 / row_key on [P_u, P_u rounded up to 32): the GEMM epilogues read whole 32-row tiles.
__global__ __launch_bounds__(CU_MAXW) void cyl_rows_kernel(const float *__restrict__ xyz, const float *__restrict__ centres,
                                                           const float *__restrict__ rot,
                                                           const int32_t *__restrict__ sorted,
                                                           const int32_t *__restrict__ meta,
                                                           const int32_t *__restrict__ count,
                                                           const int64_t *__restrict__ off, int n, int m, int W,
                                                           long long R, long long cap, float *__restrict__ x0,
                                                           float *__restrict__ row_w, uint16_t *__restrict__ row_w16,
                                                           int32_t *__restrict__ row_mem,
                                                           int32_t *__restrict__ row_key) {
  const long long r = blockIdx.x;
  const size_t set = blockIdx.y;
  sorted += set * R * W;
  meta += set * R * W;
  count += set * R;
  off += set * R;
  x0 += set * cap * 3;
  row_w += set * cap;
  row_w16 += set * cap;
  row_mem += set * cap;
  if (row_key) row_key += set * cap;
  const int u = threadIdx.x;
  const int cnt = count[r];
  if (r == R - 1 && u < 32) {
    const long long pu = off[r] + cnt, row = pu + u;
    if (row < (pu + 31) / 32 * 32 && row < cap) {
      row_w16[row] = 0;
      if (row_key) row_key[row] = 0;
    }
  }
  if (u >= cnt) return;
  const long long row = off[r] + u;
  if (row >= cap) return;   // (a capacity below the row total: the caller's error, never a write out of bounds)
  const int bi = (int)(r / m);
  const int id = sorted[r * W + u];
  const int mt = meta[r * W + u];
  const float *p = xyz + ((size_t)bi * n + id) * 3;
  const float *q = centres + r * 3;
  const float *rt = rot + r * 9;
  const float x = p[0] - q[0], y = p[1] - q[1], z = p[2] - q[2];
#pragma unroll
  for (int col = 0; col < 3; ++col) x0[row * 3 + col] = ((x * rt[col]) + (y * rt[3 + col])) + (z * rt[6 + col]);
  row_w[row] = (float)(mt >> 8);
  row_w16[row] = (uint16_t)(mt >> 8);
  row_mem[row] = mt & 0xFF;
  // (seed << 13) | (multiplicity << 4) | member bits: everything the pooled GEMM epilogue (gemm_rs.hip, RS_STATS_POOL_V)
  // needs to know about a row, in one word (multiplicity <= 256 slots, D <= 4 crops, < 2^18 seeds: host-checked)
  if (row_key) row_key[row] = (int32_t)((r << 13) | ((long long)(mt >> 8) << 4) | (mt & 0xF));
}

}  // namespace gb

using namespace gb;

extern "C" int gb_cyl_unique(const int32_t *idx, int nr, int D, long long R, int ns, int32_t *sorted, int32_t *meta,
                             int32_t *count, void *stream) {
  if (nr < 1 || nr > 65535 || D < 1 || D > 8 || R < 0 || ns < 1 || D * ns > CU_MAXW || !idx || !sorted || !meta || !count)
    return GB_EINVAL;
  if (R == 0) return GB_OK;
  if (R > 0x7fffffffLL) return GB_ERANGE;
  hipLaunchKernelGGL(cyl_unique_kernel, dim3((unsigned)R, (unsigned)nr), dim3(CU_MAXW), 0, as_stream(stream), idx, D, R, ns,
                     sorted, meta, count);
  return check_launch("gb_cyl_unique");
}

extern "C" int gb_cyl_scan(const int32_t *count, int nr, long long R, int64_t *off, long long *total, void *stream) {
  if (nr < 1 || R < 0 || !count || !off || !total) return GB_EINVAL;
  hipLaunchKernelGGL(cyl_scan_kernel, dim3((unsigned)nr), dim3(1024), 0, as_stream(stream), count, R, off, total);
  return check_launch("gb_cyl_scan");
}

extern "C" int gb_cyl_rows(const float *xyz, const float *centres, const float *rot, const int32_t *sorted,
                           const int32_t *meta, const int32_t *count, const int64_t *off, int nr, int b, int n, int m,
                           int W, long long cap, float *x0, float *row_w, uint16_t *row_w16, int32_t *row_mem,
                           int32_t *row_key, void *stream) {
  if (nr < 1 || nr > 65535 || b < 0 || n < 1 || m < 0 || W < 1 || W > CU_MAXW || cap < 32 || cap % 32 || !xyz || !centres ||
      !rot || !sorted || !meta || !count || !off || !x0 || !row_w || !row_w16 || !row_mem)
    return GB_EINVAL;
  const long long R = (long long)b * m;
  if (R == 0) return GB_OK;
  if (R > 0x7fffffffLL || (row_key && R >= (1 << 18))) return GB_ERANGE;
  hipLaunchKernelGGL(cyl_rows_kernel, dim3((unsigned)R, (unsigned)nr), dim3(CU_MAXW), 0, as_stream(stream), xyz, centres,
                     rot, sorted, meta, count, off, n, m, W, R, cap, x0, row_w, row_w16, row_mem, row_key);
  return check_launch("gb_cyl_rows");
}

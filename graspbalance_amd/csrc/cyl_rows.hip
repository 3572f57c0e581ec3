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

// One workgroup (CU_MAXW threads) per seed.  idx: (D, R, ns) int32.  sorted / meta: (R, W) with W = D*ns, the
// distinct ids of a seed compacted to the front in increasing order; meta = (multiplicity << 8) | member bits.
__global__ __launch_bounds__(CU_MAXW) void cyl_unique_kernel(const int32_t *__restrict__ idx, int D, long long R, int ns,
                                                             int32_t *__restrict__ sorted, int32_t *__restrict__ meta,
                                                             int32_t *__restrict__ count) {
  __shared__ unsigned key[CU_MAXW];  // (id << 3) | depth ; 0xFFFFFFFF = unused slot
  __shared__ int wsum[CU_MAXW / 64];
  const long long r = blockIdx.x;
  const int t = threadIdx.x, W = D * ns;
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

// x0 (P_u,3) = (xyz[b, id] - centre[seed]) rotated by the seed's matrix (gb_group_concat_cl mode 2 arithmetic),
// row_w (P_u) = multiplicity (also as uint16 in row_w16, for the GEMM epilogues), row_mem (P_u) = member bits;
// row index = off[seed] + u.
__global__ __launch_bounds__(CU_MAXW) void cyl_rows_kernel(const float *__restrict__ xyz, const float *__restrict__ centres,
                                                           const float *__restrict__ rot,
                                                           const int32_t *__restrict__ sorted,
                                                           const int32_t *__restrict__ meta,
                                                           const int32_t *__restrict__ count,
                                                           const int64_t *__restrict__ off, int n, int m, int W,
                                                           float *__restrict__ x0, float *__restrict__ row_w,
                                                           uint16_t *__restrict__ row_w16,
                                                           int32_t *__restrict__ row_mem,
                                                           int32_t *__restrict__ row_key) {
  const long long r = blockIdx.x;
  const int u = threadIdx.x;
  if (u >= count[r]) return;
  const long long row = off[r] + u;
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
  // (seed << 13) | (multiplicity << 4) | member bits: everything the pooled GEMM epilogue (gemm_rs.hip, RS_STATS_POOL)
  // needs to know about a row, in one word (multiplicity <= 256 slots, D <= 4 crops, < 2^18 seeds: host-checked)
  if (row_key) row_key[row] = (int32_t)((r << 13) | ((long long)(mt >> 8) << 4) | (mt & 0xF));
}

}  // namespace gb

using namespace gb;

extern "C" int gb_cyl_unique(const int32_t *idx, int D, long long R, int ns, int32_t *sorted, int32_t *meta,
                             int32_t *count, void *stream) {
  if (D < 1 || D > 8 || R < 0 || ns < 1 || D * ns > CU_MAXW || !idx || !sorted || !meta || !count) return GB_EINVAL;
  if (R == 0) return GB_OK;
  if (R > 0x7fffffffLL) return GB_ERANGE;
  hipLaunchKernelGGL(cyl_unique_kernel, dim3((unsigned)R), dim3(CU_MAXW), 0, as_stream(stream), idx, D, R, ns, sorted,
                     meta, count);
  return check_launch("gb_cyl_unique");
}

extern "C" int gb_cyl_rows(const float *xyz, const float *centres, const float *rot, const int32_t *sorted,
                           const int32_t *meta, const int32_t *count, const int64_t *off, int b, int n, int m, int W,
                           float *x0, float *row_w, uint16_t *row_w16, int32_t *row_mem, int32_t *row_key,
                           void *stream) {
  if (b < 0 || n < 1 || m < 0 || W < 1 || W > CU_MAXW || !xyz || !centres || !rot || !sorted || !meta || !count || !off ||
      !x0 || !row_w || !row_w16 || !row_mem)
    return GB_EINVAL;
  const long long R = (long long)b * m;
  if (R == 0) return GB_OK;
  if (R > 0x7fffffffLL || (row_key && R >= (1 << 18))) return GB_ERANGE;
  hipLaunchKernelGGL(cyl_rows_kernel, dim3((unsigned)R), dim3(CU_MAXW), 0, as_stream(stream), xyz, centres, rot, sorted,
                     meta, count, off, n, m, W, x0, row_w, row_w16, row_mem, row_key);
  return check_launch("gb_cyl_rows");
}

// Ball query and cylinder query for gfx950 — replace query_ball_point_kernel
// (PointNet/_ext_src/src/ball_query_gpu.cu:9-54), ball_query_kernel_fast
// (pointnet2_batch/src/ball_query_gpu.cu:10-58) and query_cylinder_point_kernel
// (PointNet/_ext_src/src/cylinder_query_gpu.cu:20-101).
//
// The reference gives one THREAD to a centre and scans the cloud serially (b x 512 threads in all).
// Here one WAVE owns CPW centres and the 64 lanes test 64 consecutive candidates at once:
//   * candidates are read with one coalesced 12-byte load per lane (768 contiguous bytes per wave
//     instruction), prefetched one tile ahead; centre data is wave-uniform (SGPRs);
//   * v_cmp gives the 64-bit hit mask; hits are appended in index order at cnt + popcount(lower
//     lanes), so the output equals the serial scan's; cnt / first-hit are scalars per centre;
//   * a centre stops being tested once it has nsample hits and the wave leaves the scan when all
//     of its centres are full (the reference's `cnt < nsample` loop condition);
//   * the row tail is filled with the first hit (the reference pre-fills on the first hit), an
//     empty row with 0 (the reference relies on torch::zeros) — every output element is written.
// Built with -ffp-contract=off: d2 = ((dx*dx)+(dy*dy))+(dz*dz), rotations as written in the .cu.
#include "gb_common.h"

namespace gb {

constexpr int QWAVES = 4;  // waves per workgroup

template <int CPW, bool SCAN>
__global__ __launch_bounds__(QWAVES * 64) void ball_query_kernel(
    const float *__restrict__ new_xyz, const float *__restrict__ xyz, int32_t *__restrict__ idx,
    int32_t *__restrict__ scanned, int n, int m, float radius2, int nsample) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bi = blockIdx.y;
  const int c0 = (blockIdx.x * QWAVES + wave) * CPW;  // first centre of this wave
  if (c0 >= m) return;
  const f3 *pts = reinterpret_cast<const f3 *>(xyz + (size_t)bi * n * 3);
  const float *ctr = new_xyz + ((size_t)bi * m + c0) * 3;
  int32_t *rows = idx + ((size_t)bi * m + c0) * nsample;

  float cx[CPW], cy[CPW], cz[CPW];
  int cnt[CPW], first[CPW], scan[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const bool live = c0 + c < m;
    cx[c] = live ? ctr[c * 3 + 0] : 0.f;
    cy[c] = live ? ctr[c * 3 + 1] : 0.f;
    cz[c] = live ? ctr[c * 3 + 2] : 0.f;
    cnt[c] = live ? 0 : nsample;  // dead centres are "full"
    first[c] = 0;
    scan[c] = n;
  }

  f3 cur = {0.f, 0.f, 0.f};
  if (lane < n) cur = pts[lane];
  for (int base = 0; base < n; base += 64) {
    bool all_full = true;
#pragma unroll
    for (int c = 0; c < CPW; ++c) all_full = all_full && cnt[c] >= nsample;
    if (all_full) break;
    const int k = base + lane;
    f3 nxt = {0.f, 0.f, 0.f};
    if (k + 64 < n) nxt = pts[k + 64];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      if (cnt[c] < nsample) {  // wave-uniform
        const float dx = cx[c] - cur.x, dy = cy[c] - cur.y, dz = cz[c] - cur.z;
        const float d2 = ((dx * dx) + (dy * dy)) + (dz * dz);
        const bool hit = (d2 < radius2) && (k < n);
        const unsigned long long mask = __ballot(hit);
        if (mask != 0ull) {
          const int pos = cnt[c] + prefix_popc(mask);
          if (hit && pos < nsample) rows[c * nsample + pos] = k;
          if (cnt[c] == 0) first[c] = base + (int)__builtin_ctzll(mask);
          const int total = cnt[c] + (int)__builtin_popcountll(mask);
          if (SCAN && total >= nsample) {
            // lane holding the nsample-th hit: the serial scan stops right after it
            const unsigned long long last = __ballot(hit && pos == nsample - 1);
            scan[c] = base + (int)__builtin_ctzll(last) + 1;
          }
          cnt[c] = total;
        }
      }
    }
    cur = nxt;
  }
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    if (c0 + c < m) {
      const int filled = cnt[c] < nsample ? cnt[c] : nsample;
      const int pad = cnt[c] > 0 ? first[c] : 0;
      for (int l = filled + lane; l < nsample; l += 64) rows[c * nsample + l] = pad;
      if (SCAN && lane == 0) scanned[(size_t)bi * m + c0 + c] = scan[c];
    }
  }
}

// NQ = number of (radius, hmax) predicate sets evaluated per pass (1 for the plain query)
struct CylParams {
  float r2[4];
  float hmax[4];
  float hmin;
  int nr, nh;
};

template <int NR, int NH, bool SCAN>
__global__ __launch_bounds__(QWAVES * 64) void cylinder_query_kernel(
    const float *__restrict__ new_xyz, const float *__restrict__ xyz, const float *__restrict__ rot,
    int32_t *__restrict__ idx, int32_t *__restrict__ scanned, int b, int n, int m, CylParams prm,
    int nsample) {
  constexpr int NQ = NR * NH;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bi = blockIdx.y;
  const int j = blockIdx.x * QWAVES + wave;  // one centre per wave
  if (j >= m) return;
  const f3 *pts = reinterpret_cast<const f3 *>(xyz + (size_t)bi * n * 3);
  const float *ctr = new_xyz + ((size_t)bi * m + j) * 3;
  const float *r = rot + ((size_t)bi * m + j) * 9;
  const float cx = ctr[0], cy = ctr[1], cz = ctr[2];
  const float r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5], r6 = r[6],
              r7 = r[7], r8 = r[8];
  const size_t qstride = (size_t)b * m * nsample;  // elements between two queries' outputs
  int32_t *row = idx + ((size_t)bi * m + j) * nsample;

  int cnt[NQ], first[NQ], scan[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) { cnt[q] = 0; first[q] = 0; scan[q] = n; }

  f3 cur = {0.f, 0.f, 0.f};
  if (lane < n) cur = pts[lane];
  for (int base = 0; base < n; base += 64) {
    bool all_full = true;
#pragma unroll
    for (int q = 0; q < NQ; ++q) all_full = all_full && cnt[q] >= nsample;
    if (all_full) break;
    const int k = base + lane;
    f3 nxt = {0.f, 0.f, 0.f};
    if (k + 64 < n) nxt = pts[k + 64];
    const float x = cur.x - cx, y = cur.y - cy, z = cur.z - cz;
    const float x_rot = ((r0 * x) + (r3 * y)) + (r6 * z);
    const float y_rot = ((r1 * x) + (r4 * y)) + (r7 * z);
    const float z_rot = ((r2 * x) + (r5 * y)) + (r8 * z);
    const float d2 = (y_rot * y_rot) + (z_rot * z_rot);
    const bool in_h0 = (x_rot > prm.hmin) && (k < n);
#pragma unroll
    for (int ir = 0; ir < NR; ++ir) {
      const bool in_r = in_h0 && (d2 < prm.r2[ir]);
#pragma unroll
      for (int ih = 0; ih < NH; ++ih) {
        const int q = ir * NH + ih;
        if (cnt[q] < nsample) {
          const bool hit = in_r && (x_rot < prm.hmax[ih]);
          const unsigned long long mask = __ballot(hit);
          if (mask != 0ull) {
            const int pos = cnt[q] + prefix_popc(mask);
            if (hit && pos < nsample) row[q * qstride + pos] = k;
            if (cnt[q] == 0) first[q] = base + (int)__builtin_ctzll(mask);
            const int total = cnt[q] + (int)__builtin_popcountll(mask);
            if (SCAN && total >= nsample) {
              const unsigned long long last = __ballot(hit && pos == nsample - 1);
              scan[q] = base + (int)__builtin_ctzll(last) + 1;
            }
            cnt[q] = total;
          }
        }
      }
    }
    cur = nxt;
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int filled = cnt[q] < nsample ? cnt[q] : nsample;
    const int pad = cnt[q] > 0 ? first[q] : 0;
    for (int l = filled + lane; l < nsample; l += 64) row[q * qstride + l] = pad;
    if (SCAN && lane == 0) scanned[(size_t)bi * m + j] = scan[q];  // SCAN only with NQ == 1
  }
}

}  // namespace gb

extern "C" int gb_ball_query(const float *new_xyz, const float *xyz, int32_t *idx, int32_t *scanned,
                             int b, int n, int m, float radius, int nsample, void *stream) {
  using namespace gb;
  if (b < 0 || n < 1 || m < 0 || nsample < 1 || !new_xyz || !xyz || !idx) return GB_EINVAL;
  if ((long long)n * 3 > 0x7fffffffLL || (long long)m * nsample > 0x7fffffffLL) return GB_ERANGE;
  if (b == 0 || m == 0) return GB_OK;
  if (b > 65535) return GB_ERANGE;
  const float radius2 = radius * radius;  // ball_query_gpu.cu:23
  hipStream_t s = as_stream(stream);
  // more centres per wave amortise the candidate loads when there are enough centres to fill the chip
  const long long waves4 = (long long)b * ceil_div(m, 4);
  if (waves4 >= 4096) {
    dim3 grid(ceil_div(m, QWAVES * 4), b);
    if (scanned)
      hipLaunchKernelGGL((ball_query_kernel<4, true>), grid, dim3(QWAVES * 64), 0, s, new_xyz, xyz,
                         idx, scanned, n, m, radius2, nsample);
    else
      hipLaunchKernelGGL((ball_query_kernel<4, false>), grid, dim3(QWAVES * 64), 0, s, new_xyz, xyz,
                         idx, scanned, n, m, radius2, nsample);
  } else {
    dim3 grid(ceil_div(m, QWAVES), b);
    if (scanned)
      hipLaunchKernelGGL((ball_query_kernel<1, true>), grid, dim3(QWAVES * 64), 0, s, new_xyz, xyz,
                         idx, scanned, n, m, radius2, nsample);
    else
      hipLaunchKernelGGL((ball_query_kernel<1, false>), grid, dim3(QWAVES * 64), 0, s, new_xyz, xyz,
                         idx, scanned, n, m, radius2, nsample);
  }
  return check_launch("gb_ball_query");
}

extern "C" int gb_cylinder_query(const float *new_xyz, const float *xyz, const float *rot,
                                 int32_t *idx, int32_t *scanned, int b, int n, int m, float radius,
                                 float hmin, float hmax, int nsample, void *stream) {
  using namespace gb;
  if (b < 0 || n < 1 || m < 0 || nsample < 1 || !new_xyz || !xyz || !rot || !idx) return GB_EINVAL;
  if ((long long)n * 3 > 0x7fffffffLL || (long long)m * nsample > 0x7fffffffLL) return GB_ERANGE;
  if (b == 0 || m == 0) return GB_OK;
  if (b > 65535) return GB_ERANGE;
  CylParams prm = {};
  prm.r2[0] = radius * radius;  // cylinder_query_gpu.cu:38
  prm.hmax[0] = hmax;
  prm.hmin = hmin;
  prm.nr = prm.nh = 1;
  dim3 grid(ceil_div(m, QWAVES), b);
  hipStream_t s = as_stream(stream);
  if (scanned)
    hipLaunchKernelGGL((cylinder_query_kernel<1, 1, true>), grid, dim3(QWAVES * 64), 0, s, new_xyz,
                       xyz, rot, idx, scanned, b, n, m, prm, nsample);
  else
    hipLaunchKernelGGL((cylinder_query_kernel<1, 1, false>), grid, dim3(QWAVES * 64), 0, s, new_xyz,
                       xyz, rot, idx, scanned, b, n, m, prm, nsample);
  return check_launch("gb_cylinder_query");
}

extern "C" int gb_cylinder_query_multi(const float *new_xyz, const float *xyz, const float *rot,
                                       int32_t *idx, int b, int n, int m, const float *radii, int nr,
                                       float hmin, const float *hmaxs, int nh, int nsample,
                                       void *stream) {
  using namespace gb;
  if (b < 0 || n < 1 || m < 0 || nsample < 1 || !new_xyz || !xyz || !rot || !idx || !radii || !hmaxs)
    return GB_EINVAL;
  if (nr < 1 || nr > 4 || nh < 1 || nh > 4) return GB_EINVAL;
  if ((long long)n * 3 > 0x7fffffffLL || (long long)m * nsample > 0x7fffffffLL) return GB_ERANGE;
  if (b == 0 || m == 0) return GB_OK;
  if (b > 65535) return GB_ERANGE;
  CylParams prm = {};
  for (int i = 0; i < nr; ++i) prm.r2[i] = radii[i] * radii[i];
  for (int i = 0; i < nh; ++i) prm.hmax[i] = hmaxs[i];
  prm.hmin = hmin;
  prm.nr = nr;
  prm.nh = nh;
  dim3 grid(ceil_div(m, QWAVES), b);
  hipStream_t s = as_stream(stream);
#define GB_LAUNCH(NR, NH)                                                                         \
  if (nr == NR && nh == NH) {                                                                     \
    hipLaunchKernelGGL((cylinder_query_kernel<NR, NH, false>), grid, dim3(QWAVES * 64), 0, s,     \
                       new_xyz, xyz, rot, idx, (int32_t *)nullptr, b, n, m, prm, nsample);        \
    return check_launch("gb_cylinder_query_multi");                                               \
  }
  GB_LAUNCH(1, 1) GB_LAUNCH(1, 2) GB_LAUNCH(1, 3) GB_LAUNCH(1, 4)
  GB_LAUNCH(2, 1) GB_LAUNCH(2, 2) GB_LAUNCH(2, 3) GB_LAUNCH(2, 4)
  GB_LAUNCH(3, 1) GB_LAUNCH(3, 2) GB_LAUNCH(3, 3) GB_LAUNCH(3, 4)
  GB_LAUNCH(4, 1) GB_LAUNCH(4, 2) GB_LAUNCH(4, 3) GB_LAUNCH(4, 4)
#undef GB_LAUNCH
  return GB_EINVAL;
}

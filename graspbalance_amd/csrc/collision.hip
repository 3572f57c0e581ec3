// Model-free collision check of grasp candidates against the scene cloud (collision_detector.py:6-64; SURVEY.md
// section 8 row f4): the reference forms a dense (G, M, 3) float64 tensor of scene points in every gripper frame and
// eight (G, M) boolean masks; here a workgroup streams the M scene points once per grasp, tests the gripper's finger /
// bottom / approach / inner boxes in registers and leaves six integer counts per grasp.  fp64 like the reference's
// numpy arithmetic, same evaluation order (-ffp-contract=off: no fused multiply-add), integer outputs.
//
// gb_voxel_mean is the averaging step of the voxel down-sampling the detector applies to the scene first
// (open3d.geometry.PointCloud.voxel_down_sample, :11-14): the points of a voxel are summed in fp64 in their
// original order and divided by their number.
#include "gb_common.h"

namespace gb {

constexpr int CD_TPB = 256;
constexpr int CD_SPLIT_POINTS = 8192;  // scene points per workgroup: G x ceil(M / 8192) workgroups

// thr[g] = {h/2, d, d - fl, -(w/2 + fw), -w/2, w/2 + fw, w/2, d - fl - fw, d - fl - fw - approach}
// counts[g] = {left, right, bottom, shifting, global, inner}  (caller-zeroed when M > CD_SPLIT_POINTS)
__global__ __launch_bounds__(CD_TPB) void collision_counts_kernel(const double *__restrict__ scene,
                                                                  const double *__restrict__ trans,
                                                                  const double *__restrict__ rot,
                                                                  const double *__restrict__ thr,
                                                                  int32_t *__restrict__ counts, long long M) {
  __shared__ int s_cnt[CD_TPB / 64][6];
  const int g = blockIdx.x;
  const double tx = trans[g * 3 + 0], ty = trans[g * 3 + 1], tz = trans[g * 3 + 2];
  double r[9], t[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { r[i] = rot[g * 9 + i]; t[i] = thr[g * 9 + i]; }
  const long long p0 = (long long)blockIdx.y * CD_SPLIT_POINTS;
  long long p1 = p0 + CD_SPLIT_POINTS;
  if (p1 > M) p1 = M;
  int c[6] = {0, 0, 0, 0, 0, 0};
  for (long long p = p0 + threadIdx.x; p < p1; p += CD_TPB) {
    const double dx = scene[p * 3 + 0] - tx, dy = scene[p * 3 + 1] - ty, dz = scene[p * 3 + 2] - tz;
    // targets = (scene - T) @ R   (:23-24): column j = dx R[0][j] + dy R[1][j] + dz R[2][j], left to right
    const double x = (dx * r[0] + dy * r[3]) + dz * r[6];
    const double y = (dx * r[1] + dy * r[4]) + dz * r[7];
    const double z = (dx * r[2] + dy * r[5]) + dz * r[8];
    const bool m1 = (z > -t[0]) && (z < t[0]);   // between the finger planes (:26)
    const bool m2 = (x > t[2]) && (x < t[1]);    // along the fingers (:27)
    const bool m3 = y > t[3], m4 = y < t[4];     // left finger (:28-29)
    const bool m5 = y < t[5], m6 = y > t[6];     // right finger (:30-31)
    const bool m7 = (x <= t[2]) && (x > t[7]);   // gripper bottom (:32-33)
    const bool m8 = (x <= t[7]) && (x > t[8]);   // approach volume (:34-35)
    const bool left = m1 && m2 && m3 && m4, right = m1 && m2 && m5 && m6;
    const bool bottom = m1 && m3 && m5 && m7, shifting = m1 && m3 && m5 && m8;
    c[0] += left; c[1] += right; c[2] += bottom; c[3] += shifting;
    c[4] += (left || right || bottom || shifting);
    c[5] += (m1 && m2 && !m4 && !m6);            // between the fingers (:50)
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c[i] += __shfl_xor(c[i], off);
  }
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int i = 0; i < 6; ++i) s_cnt[threadIdx.x >> 6][i] = c[i];
  __syncthreads();
  if (threadIdx.x < 6) {
    int s = 0;
    for (int w = 0; w < CD_TPB / 64; ++w) s += s_cnt[w][threadIdx.x];
    if (gridDim.y == 1) counts[g * 6 + threadIdx.x] = s;
    else if (s) atomicAdd(counts + g * 6 + threadIdx.x, s);
  }
}

// out[v] = (sum of pts[seg[v] .. seg[v+1]) in order) / count, per coordinate
__global__ __launch_bounds__(CD_TPB) void voxel_mean_kernel(const double *__restrict__ pts,
                                                            const int64_t *__restrict__ seg, double *__restrict__ out,
                                                            long long V) {
  const long long v = (long long)blockIdx.x * CD_TPB + threadIdx.x;
  if (v >= V) return;
  const long long b = seg[v], e = seg[v + 1];
  double sx = 0.0, sy = 0.0, sz = 0.0;
  for (long long p = b; p < e; ++p) { sx += pts[p * 3]; sy += pts[p * 3 + 1]; sz += pts[p * 3 + 2]; }
  const double n = (double)(e - b);
  out[v * 3] = sx / n; out[v * 3 + 1] = sy / n; out[v * 3 + 2] = sz / n;
}

}  // namespace gb

using namespace gb;

extern "C" int gb_collision_counts(const double *scene, const double *trans, const double *rot, const double *thr,
                                   int32_t *counts, int G, long long M, void *stream) {
  if (G < 0 || M < 0) return GB_EINVAL;
  if (G == 0) return GB_OK;
  if (!counts || !trans || !rot || !thr || (M > 0 && !scene)) return GB_EINVAL;
  const long long chunks = M > 0 ? (M + CD_SPLIT_POINTS - 1) / CD_SPLIT_POINTS : 1;
  if (chunks > 65535) return GB_ERANGE;
  if (chunks > 1 &&
      hipMemsetAsync(counts, 0, (size_t)G * 6 * sizeof(int32_t), as_stream(stream)) != hipSuccess)
    return GB_ELAUNCH;
  hipLaunchKernelGGL(collision_counts_kernel, dim3((unsigned)G, (unsigned)chunks), dim3(CD_TPB), 0, as_stream(stream),
                     scene, trans, rot, thr, counts, M);
  return check_launch("gb_collision_counts");
}

extern "C" int gb_voxel_mean(const double *pts_sorted, const int64_t *seg_start, double *out, long long V,
                             void *stream) {
  if (V < 0 || (V > 0 && (!pts_sorted || !seg_start || !out))) return GB_EINVAL;
  if (V == 0) return GB_OK;
  const long long blocks = (V + CD_TPB - 1) / CD_TPB;
  if (blocks > 0x7fffffffLL) return GB_ERANGE;
  hipLaunchKernelGGL(voxel_mean_kernel, dim3((unsigned)blocks), dim3(CD_TPB), 0, as_stream(stream), pts_sorted,
                     seg_start, out, V);
  return check_launch("gb_voxel_mean");
}

// Host data path of the reference moved to the GPU: depth image -> camera-frame cloud
// (data_utils.py:14-25 create_point_cloud_from_depth_image), workspace mask from the foreground's bounding box in
// the table frame (data_utils.py:52-72 get_workspace_mask, graspnet_dataset.py:118-124) and the ordered compaction
// `cloud[mask]` (graspnet_dataset.py:125-127).  One pass per stage over the H*W pixels: HBM-bound, a 1280 x 720 frame
// is 0.9 M pixels.  The arithmetic follows numpy's: everything in float64 in the reference's operation order
// ((u - cx) * z / fx), the cloud is rounded to float32 only on output (`.astype(np.float32)`, :136).
#include "gb_common.h"

namespace gb {

constexpr int FR_TPB = 256;

struct FrameCam {
  double fx, fy, cx, cy, scale;
  double t[12];  // optional 3x4 transform (rows) into the frame the workspace box is taken in
  int has_t;
};

__device__ __forceinline__ void frame_point(const FrameCam &c, int u, int v, double d, double p[3]) {
  const double z = d / c.scale;
  p[0] = (double)((double)u - c.cx) * z / c.fx;
  p[1] = (double)((double)v - c.cy) * z / c.fy;
  p[2] = z;
}

__device__ __forceinline__ void frame_transform(const FrameCam &c, const double p[3], double q[3]) {
  if (!c.has_t) { q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; return; }
#pragma unroll
  for (int r = 0; r < 3; ++r)  // np.dot(T, [x y z 1]^T): one rounding per operation, in index order
    q[r] = ((c.t[4 * r] * p[0] + c.t[4 * r + 1] * p[1]) + c.t[4 * r + 2] * p[2]) + c.t[4 * r + 3];
}

// order-preserving map of a double onto uint64 so that min / max can use integer atomics
__device__ __forceinline__ unsigned long long ord64(double x) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

// pass 1: cloud (H*W,3) fp32 and the foreground (seg > 0) bounding box of the transformed points: box[0..2] = min,
// box[3..5] = max as ord64 keys (caller-initialised to ~0 / 0)
template <typename DepthT>
__global__ __launch_bounds__(FR_TPB) void frame_cloud_kernel(const DepthT *__restrict__ depth,
                                                             const int32_t *__restrict__ seg, FrameCam cam, int H, int W,
                                                             float *__restrict__ cloud,
                                                             unsigned long long *__restrict__ box) {
  __shared__ unsigned long long s_box[6];
  if (threadIdx.x < 6) s_box[threadIdx.x] = threadIdx.x < 3 ? ~0ull : 0ull;
  __syncthreads();
  const long long i = (long long)blockIdx.x * FR_TPB + threadIdx.x;
  if (i < (long long)H * W) {
    const int v = (int)(i / W), u = (int)(i % W);
    double p[3], q[3];
    frame_point(cam, u, v, (double)depth[i], p);
    if (cloud) {
      cloud[3 * i] = (float)p[0];
      cloud[3 * i + 1] = (float)p[1];
      cloud[3 * i + 2] = (float)p[2];
    }
    if (box && seg[i] > 0) {
      frame_transform(cam, p, q);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        atomicMin(&s_box[a], ord64(q[a]));
        atomicMax(&s_box[3 + a], ord64(q[a]));
      }
    }
  }
  __syncthreads();
  if (box && threadIdx.x < 6) {
    if (threadIdx.x < 3) { if (s_box[threadIdx.x] != ~0ull) atomicMin(&box[threadIdx.x], s_box[threadIdx.x]); }
    else if (s_box[threadIdx.x] != 0ull) atomicMax(&box[threadIdx.x], s_box[threadIdx.x]);
  }
}

__device__ __forceinline__ double unord64(unsigned long long k) {
  const unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

// pass 2: mask = depth > 0 & strictly inside the box widened by `outlier` (no box: depth > 0 only); per-workgroup
// counts for the ordered compaction
template <typename DepthT>
__global__ __launch_bounds__(FR_TPB) void frame_mask_kernel(const DepthT *__restrict__ depth, FrameCam cam, int H, int W,
                                                            const unsigned long long *__restrict__ box, double outlier,
                                                            uint8_t *__restrict__ mask, int32_t *__restrict__ counts) {
  __shared__ int s_cnt[FR_TPB / 64];
  const long long i = (long long)blockIdx.x * FR_TPB + threadIdx.x;
  bool keep = false;
  if (i < (long long)H * W) {
    const double d = (double)depth[i];
    keep = d > 0.0;
    if (box) {
      double p[3], q[3];
      frame_point(cam, (int)(i % W), (int)(i / W), d, p);
      frame_transform(cam, p, q);
      bool in = true;
#pragma unroll
      for (int a = 0; a < 3; ++a)
        in = in && (q[a] > unord64(box[a]) - outlier) && (q[a] < unord64(box[3 + a]) + outlier);
      if (mask) mask[i] = in ? 1 : 0;   // the reference's workspace_mask alone ...
      keep = keep && in;                // ... and combined with depth > 0 for the compaction
    } else if (mask) {
      mask[i] = keep ? 1 : 0;
    }
  }
  const unsigned long long b = __ballot(keep);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < FR_TPB / 64; ++w) t += s_cnt[w];
    counts[blockIdx.x] = t;
  }
}

// pass 3: out_idx[offset(block) + rank] = pixel index of every kept pixel, in pixel order (== np.nonzero(mask))
template <typename DepthT>
__global__ __launch_bounds__(FR_TPB) void frame_compact_kernel(const DepthT *__restrict__ depth, FrameCam cam, int H, int W,
                                                               const unsigned long long *__restrict__ box, double outlier,
                                                               const int64_t *__restrict__ offsets,
                                                               int32_t *__restrict__ out_idx) {
  __shared__ int s_cnt[FR_TPB / 64];
  const long long i = (long long)blockIdx.x * FR_TPB + threadIdx.x;
  bool keep = false;
  if (i < (long long)H * W) {
    const double d = (double)depth[i];
    keep = d > 0.0;
    if (box && keep) {
      double p[3], q[3];
      frame_point(cam, (int)(i % W), (int)(i / W), d, p);
      frame_transform(cam, p, q);
#pragma unroll
      for (int a = 0; a < 3; ++a)
        keep = keep && (q[a] > unord64(box[a]) - outlier) && (q[a] < unord64(box[3 + a]) + outlier);
    }
  }
  const unsigned long long b = __ballot(keep);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) s_cnt[wave] = __popcll(b);
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += s_cnt[w];
  if (keep) out_idx[offsets[blockIdx.x] + base + prefix_popc(b)] = (int32_t)i;
}

}  // namespace gb

using namespace gb;

static bool frame_cam(const double *cam5, const double *trans12, FrameCam *c) {
  if (!cam5 || cam5[0] == 0.0 || cam5[1] == 0.0 || cam5[4] == 0.0) return false;
  c->fx = cam5[0]; c->fy = cam5[1]; c->cx = cam5[2]; c->cy = cam5[3]; c->scale = cam5[4];
  c->has_t = trans12 != nullptr;
  for (int i = 0; i < 12; ++i) c->t[i] = trans12 ? trans12[i] : 0.0;
  return true;
}

#define GB_FRAME_DISPATCH(KERNEL, ...)                                                                       \
  do {                                                                                                       \
    if (depth_is_u16)                                                                                        \
      hipLaunchKernelGGL((KERNEL<uint16_t>), grid, dim3(FR_TPB), 0, as_stream(stream),                      \
                         reinterpret_cast<const uint16_t *>(depth), __VA_ARGS__);                           \
    else                                                                                                     \
      hipLaunchKernelGGL((KERNEL<float>), grid, dim3(FR_TPB), 0, as_stream(stream),                         \
                         reinterpret_cast<const float *>(depth), __VA_ARGS__);                              \
  } while (0)

extern "C" int gb_frame_cloud(const void *depth, int depth_is_u16, const int32_t *seg, const double *cam5,
                              const double *trans12, int H, int W, float *cloud, unsigned long long *box,
                              void *stream) {
  FrameCam c;
  if (!depth || H < 1 || W < 1 || (!cloud && !box) || (box && !seg) || !frame_cam(cam5, trans12, &c)) return GB_EINVAL;
  const dim3 grid((unsigned)(((long long)H * W + FR_TPB - 1) / FR_TPB));
  GB_FRAME_DISPATCH(frame_cloud_kernel, seg, c, H, W, cloud, box);
  return check_launch("gb_frame_cloud");
}

extern "C" int gb_frame_mask(const void *depth, int depth_is_u16, const double *cam5, const double *trans12, int H, int W,
                             const unsigned long long *box, double outlier, uint8_t *mask, int32_t *counts,
                             void *stream) {
  FrameCam c;
  if (!depth || H < 1 || W < 1 || !counts || !frame_cam(cam5, trans12, &c)) return GB_EINVAL;
  const dim3 grid((unsigned)(((long long)H * W + FR_TPB - 1) / FR_TPB));
  GB_FRAME_DISPATCH(frame_mask_kernel, c, H, W, box, outlier, mask, counts);
  return check_launch("gb_frame_mask");
}

extern "C" int gb_frame_compact(const void *depth, int depth_is_u16, const double *cam5, const double *trans12, int H,
                                int W, const unsigned long long *box, double outlier, const int64_t *offsets,
                                int32_t *out_idx, void *stream) {
  FrameCam c;
  if (!depth || H < 1 || W < 1 || !offsets || !out_idx || !frame_cam(cam5, trans12, &c)) return GB_EINVAL;
  const dim3 grid((unsigned)(((long long)H * W + FR_TPB - 1) / FR_TPB));
  GB_FRAME_DISPATCH(frame_compact_kernel, c, H, W, box, outlier, offsets, out_idx);
  return check_launch("gb_frame_compact");
}

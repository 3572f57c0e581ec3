/*
 * graspbal.h — C-ABI of libgraspbal_hip.so, the MI355X (gfx950) implementation of the
 * GraspBalance point-cloud hot path.
 *
 * Every entry point replaces one raw-pointer launcher of the reference's two CUDA
 * extensions (paths relative to the reference tree):
 *   PN-ext = PointNet/_ext_src        (python module pointnet2._ext,        bindings.cpp:12-26)
 *   PB-ext = pointnet2_batch/src      (python module pointnet2_batch_cuda,  pointnet2_api.cpp:10-24)
 *   KNN    = KNN/Pytorch_CUDA_KNN     (python module KNN._C,                vision.cpp:3-5)
 *
 * Conventions
 *   - all pointers are DEVICE pointers (hipMalloc / torch CUDA storage), contiguous, fp32 data,
 *     int32 indices (int64, 1-based for gb_knn) — the reference's dtype contract
 *     (_ext_src/include/utils.h:10-30);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); launches are
 *     asynchronous, the library never synchronises, allocates (workspaces are caller-provided), or
 *     keeps mutable global state (per-call options travel in GbGemmOpts), so it is re-entrant from
 *     autograd worker threads like the reference launchers;
 *   - return value: GB_OK (0) or a negative GB_E* code; the library never calls exit() and never
 *     throws (the reference does fprintf+exit(-1), cuda_utils.h:38-47);
 *   - outputs are fully written by the kernels unless stated otherwise ("accumulates into" means
 *     the caller zero-fills first, exactly like torch::zeros in the reference wrappers);
 *   - distances use the no-FMA evaluation order ((dx*dx)+(dy*dy))+(dz*dz); the library is built
 *     with -ffp-contract=off so results are bit-identical to the CPU oracle in oracle/.
 */
#ifndef GRASPBAL_H
#define GRASPBAL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GB_ABI_VERSION 7

enum {
  GB_OK = 0,
  GB_EINVAL = -1,   /* bad dimension / null pointer / unsupported flag combination */
  GB_ELAUNCH = -2,  /* hipLaunchKernel reported an error (hipGetLastError text via gb_last_error) */
  GB_ERANGE = -3    /* a dimension exceeds what the kernels index with int32 */
};

/* ---- gb_fps flags -------------------------------------------------------------------------- */
/* skip points with x*x+y*y+z*z <= 1e-3 (PN-ext sampling_gpu.cu:105-106); PB-ext has no skip.   */
#define GB_FPS_SKIP_NEAR_ORIGIN 0x1u
/* tie-break among exactly equal maxima:
 *   GB_FPS_TIE_LOWEST : lowest point index wins (the reference's torch fallback,
 *                       TrainModel/pointnet2_util.py:41)
 *   GB_FPS_TIE_TREE512: winner of the reference's strided scan + shared-memory tree with
 *                       block_size = min(2^floor(log2 n), 512)   (PN-ext sampling_gpu.cu:64-178)
 *   GB_FPS_TIE_TREE1024: same with the PB-ext cap of 1024         (pointnet2_batch sampling_gpu.cu:65-181)
 * All three coincide when no two candidates tie exactly.                                        */
#define GB_FPS_TIE_LOWEST   0x00u
#define GB_FPS_TIE_TREE512  0x10u
#define GB_FPS_TIE_TREE1024 0x20u
#define GB_FPS_TIE_MASK     0x30u
/* gb_fps_pruned only - how a register-resident cloud (n <= 20480) is spread over the waves of one CU.  Same outputs
 * for every layout; AUTO lets the library choose.  W4 / W8 / W12 / W16: that many waves, the rows of a wave in
 * run-time indexed register vectors, the winner's coordinates carried through the reduction (fps.hip, round 5).
 * R4 is the round-4 kernel (16 waves, one branch per row, the winner's coordinates re-read from global memory every
 * iteration), kept for A/B measurements.                                                                            */
#define GB_FPS_LAYOUT_AUTO  0x000u
#define GB_FPS_LAYOUT_W4    0x100u
#define GB_FPS_LAYOUT_W8    0x200u
#define GB_FPS_LAYOUT_W12   0x300u
#define GB_FPS_LAYOUT_W16   0x400u
#define GB_FPS_LAYOUT_R4    0x500u
#define GB_FPS_LAYOUT_MASK  0x700u

int gb_abi_version(void);
/* last launch error text of the calling thread ("" if none) */
const char *gb_last_error(void);

/* Furthest point sampling, start index 0.
 * replaces furthest_point_sampling_kernel_wrapper (PN-ext sampling_gpu.cu:180-234) and
 *          furthest_point_sampling_kernel_launcher (PB-ext sampling_gpu.cu:183-220).
 * xyz (b,n,3) f32; idx (b,m) i32 out.
 * temp (b,n) f32 running min-distance: may be NULL (the library then starts from 1e10 like
 * sampling.cpp:78-80); if non-NULL it is read as the initial state and the final state is written
 * back (PB-ext caller-allocated contract, subsample.py:76-79).                                   */
int gb_fps(const float *xyz, float *temp, int32_t *idx, int b, int n, int m, unsigned flags,
           void *stream);
/* Pruned FPS for large clouds: same outputs as gb_fps, bit for bit (same samples, same order, same tie rules),
 * with the per-iteration min-distance update skipped for the wavefronts whose points are all farther from the
 * new sample than their current min-distance.  `perm` (b,n) int32 is any permutation of each cloud's indices;
 * a spatially coherent one (sort by gb_fps_morton_keys) is what makes the skipping effective.  n <= 65536; for
 * n > 20480 the cloud no longer fits one CU's registers and `scratch` (b*n*4 floats, 16-byte aligned) receives a
 * sorted (x,y,z,key) copy that the touched rows are re-read from; temp (optional) as in gb_fps.                                                                          */
int gb_fps_pruned(const float *xyz, const int32_t *perm, float *temp, int32_t *idx, int b, int n, int m,
                  unsigned flags, float *scratch, void *stream);
/* perm (b,n) int32: a spatially coherent permutation of each cloud's indices for gb_fps_pruned - counting sort by the
 * Morton code of a 32^3 grid cell, one launch; the order inside a cell is not deterministic (gb_fps_pruned's output
 * does not depend on the permutation).                                                                  */
int gb_fps_cell_order(const float *xyz, int32_t *perm, int b, int n, void *stream);
/* perm (b,n) int32: the permutation gb_fps_pruned skips most with (round 5) - two levels of equal-count splits along the
 * locally widest axis, cut at multiples of 64 points, so that a ROW of 64 consecutive points is a short piece of a thin
 * slab (5.4 of 313 rows touched per sample on a 20 000-point table-top scene; 11.6 in gb_fps_cell_order's order).  One
 * launch; n > 24576 falls back to gb_fps_cell_order.  Not deterministic inside a bin; gb_fps_pruned's output does not
 * depend on the permutation.                                                                                 */
int gb_fps_row_order(const float *xyz, int32_t *perm, int b, int n, void *stream);
/* The same with a caller's workspace `ws` (b*n int32) for the level-1 order: clouds of 24577 .. 65536 points get the compact
 * rows too (without `ws`, or beyond 65536 points: gb_fps_cell_order's order).  `ws` may be the `scratch` the following
 * gb_fps_pruned call takes (it is dead by then).                                                                 */
int gb_fps_row_order_ws(const float *xyz, int32_t *perm, int32_t *ws, int b, int n, void *stream);
/* keys (b,n) int32: 30-bit Morton code of each point within its cloud's bounding box.                   */
int gb_fps_morton_keys(const float *xyz, int32_t *keys, int b, int n, void *stream);
/* Segmented FPS - the per-object sampling loop of ObjectBalanceSampling (TrainModel/modules.py:178-221, one
 * furthest_point_sample(points[seg == j], share_j) call per object per cloud) as ONE launch, a workgroup per
 * segment.  xyz (T,3): the segments' points packed back to back; seg_off, out_off (S+1) int32 device arrays:
 * segment s = points [seg_off[s], seg_off[s+1]), its out_off[s+1]-out_off[s] samples are written to
 * idx[out_off[s]..) as indices WITHIN the segment, exactly what gb_fps returns for that segment as a cloud of its
 * own (same flags, same tie rule for its size).  temp (T) floats: workspace.  max_n: largest segment size.       */
int gb_fps_segments(const float *xyz, const int32_t *seg_off, const int32_t *out_off, float *temp, int32_t *idx,
                    int S, int max_n, unsigned flags, void *stream);
/* gb_fps preceded by a parallel, exact check of the hypothesis "the samples are 0..m-1" (true when the input is
 * itself in farthest-point order, e.g. the centres of the previous set-abstraction level, and no exact tie is broken
 * differently): clouds that pass skip the m-1 sequential iterations, the others run them.  Same outputs as gb_fps
 * in every case.  scratch_T (b,m), scratch_temp (b,n) floats, ok (b) int32: workspace.  n <= 24576, m <= n.        */
int gb_fps_guarded(const float *xyz, float *temp, int32_t *idx, int b, int n, int m, unsigned flags, float *scratch_T,
                   float *scratch_temp, int32_t *ok, void *stream);

/* out[b,c,j] = points[b,c,idx[b,j]]  — gather_points_kernel_wrapper (PN sampling_gpu.cu:27-35),
 * gather_points_kernel_launcher_fast (PB sampling_gpu.cu:21-33). points (b,c,n), idx (b,m).      */
int gb_gather(const float *points, const int32_t *idx, float *out, int b, int c, int n, int m,
              void *stream);
/* grad_points[b,c,idx[b,j]] += grad_out[b,c,j]; accumulates into grad_points (b,c,n).
 * gather_points_grad_kernel_wrapper (PN sampling_gpu.cu:54-62), PB sampling_gpu.cu:50-62.        */
int gb_gather_grad(const float *grad_out, const int32_t *idx, float *grad_points, int b, int c,
                   int n, int m, void *stream);

/* Ball query: for each centre the first `nsample` points in index order with d2 < radius^2
 * (strict), row padded with the first hit, all-zero row when there is no hit.
 * query_ball_point_kernel_wrapper (PN ball_query_gpu.cu:46-54),
 * ball_query_kernel_launcher_fast (PB ball_query_gpu.cu:45-58).
 * new_xyz (b,m,3), xyz (b,n,3), idx (b,m,nsample) out (every element written).
 * scanned (b,m) i32 may be NULL; if given it receives, per centre, the number of points the
 * reference's serial scan would have visited (min(n, 1 + position of the nsample-th hit)) — the
 * algorithmic pair count used for roofline accounting.                                           */
int gb_ball_query(const float *new_xyz, const float *xyz, int32_t *idx, int32_t *scanned, int b,
                  int n, int m, float radius, int nsample, void *stream);

/* Cylinder query (PN cylinder_query_gpu.cu:20-78): p = xyz - centre, rotated by the centre's
 * row-major 3x3 `rot` as p^T R; accept if y'^2+z'^2 < radius^2 and hmin < x' < hmax.
 * rot (b,m,9). Same ordering / padding / zero-row rules as gb_ball_query.                       */
int gb_cylinder_query(const float *new_xyz, const float *xyz, const float *rot, int32_t *idx,
                      int32_t *scanned, int b, int n, int m, float radius, float hmin, float hmax,
                      int nsample, void *stream);

/* The nr x nh cylinder queries of GraspPoseStage2 (TrainModel/graspbalance.py:84-87,
 * modules.py:99-101) in one pass over xyz: query (ir,ih) uses radii[ir], hmin, hmaxs[ih] and
 * writes idx[(ir*nh+ih)][b][m][nsample]. radii / hmaxs are HOST arrays (nr, nh <= 4).
 * Results are bit-identical to nr*nh separate gb_cylinder_query calls.                          */
int gb_cylinder_query_multi(const float *new_xyz, const float *xyz, const float *rot,
                            int32_t *idx, int b, int n, int m, const float *radii, int nr,
                            float hmin, const float *hmaxs, int nh, int nsample, void *stream);

/* out[b,c,j,k] = points[b,c,idx[b,j,k]] — group_points_kernel_wrapper (PN group_points_gpu.cu:46-55),
 * PB group_points_gpu.cu:57-70. points (b,c,n), idx (b,m,ns), out (b,c,m,ns).                    */
int gb_group(const float *points, const int32_t *idx, float *out, int b, int c, int n, int m,
             int nsample, void *stream);
/* grad_points[b,c,idx[b,j,k]] += grad_out[b,c,j,k]; accumulates into grad_points (b,c,n).
 * PN group_points_gpu.cu:92-101, PB group_points_gpu.cu:24-37.                                   */
int gb_group_grad(const float *grad_out, const int32_t *idx, float *grad_points, int b, int c,
                  int n, int m, int nsample, void *stream);

/* Three nearest `known` points of every `unknown` point: squared distances ascending + indices;
 * ties keep the lower index (strict <, PN interpolate_gpu.cu:14-64, PB interpolate_gpu.cu:16-59).
 * unknown (b,n,3), known (b,m,3), dist2 (b,n,3) f32, idx (b,n,3) i32. With m < 3 the missing
 * slots are (+inf, 0) like the reference's 1e40 -> float conversion.                            */
int gb_three_nn(const float *unknown, const float *known, float *dist2, int32_t *idx, int b, int n,
                int m, void *stream);
/* The inverse-distance interpolation weights of PointnetFPModule (reference PointNet/pointnet2_modules.py:260-263 with the
 * sqrt of pointnet2_utils.py:84) from gb_three_nn's SQUARED distances: dist2, weight (rows, 3) f32;
 * w_k = r_k / ((r_0 + r_1) + r_2), r_k = 1 / (sqrt(dist2_k) + 1e-8).                                                  */
int gb_interp_weights(const float *dist2, float *weight, long long rows, void *stream);
/* out[b,c,j] = sum_t points[b,c,idx[b,j,t]] * weight[b,j,t], evaluated (p1*w1 + p2*w2) + p3*w3.
 * PN interpolate_gpu.cu:77-116, PB :84-117. points (b,c,m), idx/weight (b,n,3), out (b,c,n).      */
int gb_three_interpolate(const float *points, const int32_t *idx, const float *weight, float *out,
                         int b, int c, int m, int n, void *stream);
/* grad_points[b,c,idx[b,j,t]] += grad_out[b,c,j]*weight[b,j,t]; accumulates into (b,c,m).
 * PN interpolate_gpu.cu:121-159, PB :127-168.                                                    */
int gb_three_interpolate_grad(const float *grad_out, const int32_t *idx, const float *weight,
                              float *grad_points, int b, int c, int n, int m, void *stream);

/* Brute-force 1-NN in `dim` dimensions (KNN/Pytorch_CUDA_KNN/knn.h:11-59 with k = 1, the only k
 * the reference calls: label_generation.py:58,84). ref (b,dim,nref), query (b,dim,nq) f32,
 * idx (b,1,nq) int64, 1-BASED; the lowest reference index wins ties (stable sort of knn_cpu.cpp). */
int gb_knn1(const float *ref, const float *query, int64_t *idx, int b, int dim, int nref, int nq,
            void *stream);
/* ... and for any k <= 16 (knn.h:11-59, cuda/knn.cu:113-176, cpu/knn_cpu.cpp:4-55): idx (b,k,nq) int64, 1-based, row i
 * = the (i+1)-th nearest reference of each query; equal distances in ascending index order (the reference's stable
 * insertion / bubble sorts).  k = 1 runs gb_knn1's kernels.  k > nref, k > 16 or dim > 8: GB_EINVAL.              */
int gb_knn(const float *ref, const float *query, int64_t *idx, int b, int dim, int nref, int nq, int k, void *stream);

/* Grasp-label gather for the training-time label matching (label_generation.py:60-99):
 *   out[r, v, :] = srcs[obj[r]][pt[r], view_inds[obj[r], v], :]     (W floats per (point, view))
 * srcs: HOST array of nsrc (<= 128) device pointers, one (Np_o, V, W) tensor per object - the table travels in the
 * kernel arguments, so fresh label tensors every step cost neither a host-to-device copy nor a lookup cache;
 * obj/pt (R) int32; view_inds (n_objects, V) int64; out (R, V, W), may be NULL when only out_max / out_col are wanted.  Composes the reference's two index_selects
 * (views then seeds) so only the kept rows are copied.  out_max (optional; one float, caller-initialised to
 * -inf) receives the maximum of everything gathered, NaN if any (the `.max()` of label_generation.py:113).
 * out_col (optional, (R, V, W / col_stride)): a contiguous copy of the columns w with w % col_stride == col_off -
 * the widths offsets[..., 2] that gb_label_finish needs, so it reads a third of the offsets tensor's bytes. */
int gb_label_gather(const float *const *srcs, int nsrc, const int32_t *obj, const int32_t *pt,
                    const int64_t *view_inds, float *out, float *out_max, float *out_col, int col_stride,
                    int col_off, int R, int V, int W, void *stream);
/* The "lean" label matching of a TRAINING step: of the (B,Ns,V,A,D[,3]) label / offset / tolerance tensors that
 * label_generation.py:60-116 builds for every seed and all V template views (1.2 GB at B = 4), the step consumes the
 * per-view maxima (view labels), the rows of ONE view per seed (the view the network picked, :138-157) and one width
 * per seed (loss.py:29-42).  gb_label_gather with out = NULL gives the labels' maximum in one read pass;
 *   gb_label_scores     : view_scores / view_arg (R*V) exactly as gb_label_finish, the labels and widths read straight
 *                         from the objects' label (Np,V,ad) and offset (Np,V,ad,3) tensors (two host pointer tables)
 *   gb_label_gather_view: out (R,W) = srcs[obj[r]][pt[r], view_inds[obj[r], row_view[r]], :]                      */
int gb_label_scores(const float *const *label_srcs, const float *const *offset_srcs, int nsrc, const int32_t *obj,
                    const int32_t *pt, const int64_t *view_inds, const float *u_max, float max_width,
                    float *view_scores, int32_t *view_arg, int R, int V, int ad, void *stream);
int gb_label_gather_view(const float *const *srcs, int nsrc, const int32_t *obj, const int32_t *pt,
                         const int64_t *view_inds, const int64_t *row_view, float *out, int R, int V, int W,
                         void *stream);
/* The same three entries with the pointer tables in DEVICE memory (nsrc pointers each, every tensor 16-byte aligned): the
 * launch reads the tensors' addresses from the table when it runs, not when it is enqueued, so a captured / replayed step
 * follows a table the caller rewrites per batch (1 KB) instead of needing the label tensors copied to fixed addresses
 * (2.8 GB per batch at B = 4): graspbalance_amd/train.py, _StaticBatch.                                             */
int gb_label_gather_dt(const float *const *srcs_dev, int nsrc, const int32_t *obj, const int32_t *pt,
                       const int64_t *view_inds, float *out, float *out_max, float *out_col, int col_stride,
                       int col_off, int R, int V, int W, void *stream);
int gb_label_scores_dt(const float *const *label_srcs_dev, const float *const *offset_srcs_dev, int nsrc,
                       const int32_t *obj, const int32_t *pt, const int64_t *view_inds, const float *u_max,
                       float max_width, float *view_scores, int32_t *view_arg, int R, int V, int ad, void *stream);
int gb_label_gather_view_dt(const float *const *srcs_dev, int nsrc, const int32_t *obj, const int32_t *pt,
                            const int64_t *view_inds, const int64_t *row_view, float *out, int R, int V, int W,
                            void *stream);
/* ---- host data path on the GPU (reference data_utils.py:14-72, graspnet_dataset.py:110-136) ----
 * A depth frame becomes the network's input cloud without leaving the device: three passes over the H*W pixels.
 * depth: (H,W) uint16 (depth_is_u16 = 1) or float32; cam5 = HOST [fx, fy, cx, cy, scale] doubles (CameraInfo);
 * trans12 = HOST 3x4 row-major transform (cam0_wrt_table . camera_pose, graspnet_dataset.py:121) or NULL.
 * Arithmetic in float64 in numpy's operation order; the cloud is rounded to float32 on output (:136).
 *   gb_frame_cloud : cloud (H*W,3) f32 (may be NULL) = create_point_cloud_from_depth_image (data_utils.py:14-25);
 *                    box (6 x uint64, caller-initialised to {~0,~0,~0,0,0,0}; may be NULL) = order-preserving keys of
 *                    the min / max of the TRANSFORMED points with seg > 0 (get_workspace_mask :61-63; seg (H,W) int32)
 *   gb_frame_mask  : mask (H*W) u8 (may be NULL) = workspace mask (:64-67: strictly inside the box widened by
 *                    `outlier`; box NULL: depth > 0), counts (one int32 per 256 pixels) = kept pixels per workgroup,
 *                    kept = depth > 0 & workspace mask (graspnet_dataset.py:118-124)
 *   gb_frame_compact: out_idx[offsets[wg] + rank] = pixel index of every kept pixel in pixel order
 *                    (== np.nonzero(mask): `cloud[mask]` :125 is a gather with it); offsets = exclusive scan of counts */
int gb_frame_cloud(const void *depth, int depth_is_u16, const int32_t *seg, const double *cam5,
                   const double *trans12, int H, int W, float *cloud, unsigned long long *box, void *stream);
int gb_frame_mask(const void *depth, int depth_is_u16, const double *cam5, const double *trans12, int H, int W,
                  const unsigned long long *box, double outlier, uint8_t *mask, int32_t *counts, void *stream);
int gb_frame_compact(const void *depth, int depth_is_u16, const double *cam5, const double *trans12, int H, int W,
                     const unsigned long long *box, double outlier, const int64_t *offsets, int32_t *out_idx,
                     void *stream);

/* Model-free collision check of grasp candidates (collision_detector.py:6-64; csrc/collision.hip), fp64 like the
 * reference's numpy arithmetic.
 *   gb_voxel_mean      : the averaging step of open3d's voxel_down_sample (:11-14): pts_sorted (N,3) grouped by voxel in
 *                        original point order, seg_start (V+1) first row of every voxel -> out (V,3) = per-voxel mean
 *   gb_collision_counts: scene (M,3); per grasp g: trans (G,3), rot (G,9 row-major), thr (G,9) =
 *                        {h/2, d, d-fl, -(w/2+fw), -w/2, w/2+fw, w/2, d-fl-fw, d-fl-fw-approach}  (:26-35) ->
 *                        counts (G,6) int32 = points in the {left, right, bottom, shifting, any of the four, inner}
 *                        volumes of the gripper (:37-41, :50), i.e. the row sums the reference divides by the volumes */
int gb_voxel_mean(const double *pts_sorted, const int64_t *seg_start, double *out, long long V, void *stream);
int gb_collision_counts(const double *scene, const double *trans, const double *rot, const double *thr,
                        int32_t *counts, int G, long long M, void *stream);

/* Score transform + per-view maximum of the gathered labels (reference label_generation.py:112-116):
 * out = log(*u_max / label) where label > 0 and offsets[..., 2] (width) <= max_width, else 0;
 * view_scores[row] = max over the ad = A*D grasps of the row.  labels/out (rows, ad), offsets (rows, ad, 3),
 * u_max: device scalar (the maximum of `labels`); ad % 4 == 0; 16-byte aligned tensors.  widths (optional,
 * (rows, ad)): offsets[..., 2] as its own contiguous tensor (gb_label_gather's out_col) - read instead of `offsets`,
 * which may then be NULL.  view_arg (optional, (rows) int32): the position in [0, ad) of that maximum, the first
 * one - with view_scores it gives the arg-max over all views of a seed that loss.py:31 takes, without another
 * pass over the tensor.                                                                                    */
int gb_label_finish(const float *labels, const float *offsets, const float *widths, const float *u_max,
                    float max_width, float *out, float *view_scores, int32_t *view_arg, long long rows, int ad,
                    void *stream);

/* ---- channel-last fused pieces of the SharedMLP (1x1 conv + BatchNorm + ReLU + max over nsample) ----
 * No reference launcher corresponds one-to-one: these replace the torch passes the reference runs
 * around its cuBLAS/cuDNN GEMMs — QueryAndGroup's cat / sub / div (pointnet2_utils.py:178-192),
 * CylinderQueryAndGroup's matmul (:281-284), BatchNorm2d + ReLU (pytorch_utils.py:33-59,96-110),
 * F.max_pool2d (pointnet2_modules.py:165-169) and their autograd mirrors.
 * Activations are position-major rows act[p][c], p = (b*m + j)*nsample + k.                        */

/* X0[p] = [ g(xyz[b,idx[p]] - new_xyz[b,j]) , feat[b,idx[p],:] ],  out (b*m*ns, 3+c).
 * feat (b,n,c) channel-last (may be NULL when c == 0).  mode 0: g = identity; 1: g = * scale
 * (normalize_xyz, scale = 1/radius); 2: g = rotation by rot (b,m,9), p^T R.                       */
int gb_group_concat_cl(const float *xyz, const float *new_xyz, const int32_t *idx, const float *feat,
                       const float *rot, float *out, int b, int n, int m, int ns, int c, int mode,
                       float scale, void *stream);
/* dfeat[b, idx[p], :] += dx0[p, 3:]   (accumulates into dfeat (b,n,c)) */
int gb_group_concat_cl_grad(const float *dx0, const int32_t *idx, float *dfeat, int b, int n, int m,
                            int ns, int c, void *stream);
/* `segments` independent copies in one launch: table = DEVICE array of {const float *src; long long dst_element;
 * long long count (<= 8192)} (24 bytes each, 8-byte aligned); dst[dst_element .. + count) = src[0 .. count).  What a
 * flat-buffer optimizer needs to gather its parameters' gradients (the reference's torch.optim.Adam walks the
 * parameters one by one, train.py:94); source and destination ranges must not overlap.                       */
int gb_copy_segments(const void *table, int segments, float *dst, void *stream);
/* Feature propagation's front end on channel-last rows (reference pointnet2_modules.py:402-435: three_interpolate ->
 * torch.cat with the skip features -> SharedMLP): out (b*n, c2 + c1), row j = [ (known[i0]*w0 + known[i1]*w1) +
 * known[i2]*w2 , skip[j] ] with known (b,m,c2), skip (b,n,c1) (NULL when c1 == 0) channel-last, idx / weight (b,n,3) as
 * gb_three_nn / the reference's weights: the values of gb_three_interpolate, written once where the MLP reads them.
 * _grad: dknown (b,m,c2) += the interpolation's transpose (caller-zeroed; a row of c2 consecutive floats per (j, k)),
 * dskip (b,n,c1) = dx0[:, c2:]; either may be NULL.                                                              */
int gb_interp_concat_cl(const float *known, const int32_t *idx, const float *weight, const float *skip, float *out, int b,
                        int n, int m, int c2, int c1, void *stream);
int gb_interp_concat_cl_grad(const float *dx0, const int32_t *idx, const float *weight, float *dknown, float *dskip, int b,
                             int n, int m, int c2, int c1, void *stream);
/* Per-call options of the gb_gemm_* entry points (NULL = all defaults).  The library keeps NO state between calls:
 * what used to be process-wide switches travels with every call, so two threads (autograd workers, two trainers)
 * may use different settings concurrently.
 *   precision   : GB_PREC_F32 (default; exact fp32 MFMA, the 1e-5-parity configurations) or GB_PREC_BF16 (BASELINE
 *                 configs[4], "mixed bf16 MLP / fp32 geometry" - what torch autocast around pytorch_utils.py:61-113 would
 *                 give the reference): both operands are rounded to bf16 on their way into the matrix cores;
 *                 accumulation, BatchNorm statistics and all tensors in memory stay fp32; reductions shorter than 16
 *                 (the xyz-only first layers) stay fp32.
 *   reserved_cus: size the persistent row-streaming grids for (CUs - reserved_cus) compute units, 0..128: for callers
 *                 that keep a long one-workgroup-per-cloud kernel (furthest-point sampling of the NEXT batch) resident on
 *                 a side stream.
 *   scratch     : caller-owned device workspace (16-byte aligned) of scratch_bytes bytes, used by gb_gemm_fwd /
 *                 gb_gemm_dgrad for products with few output tiles and a long reduction: the reduction is split over
 *                 workgroups, every chunk stores its partial product there and the chunks are added in chunk order
 *                 (bit-reproducible).  GB_GEMM_SCRATCH_BYTES always suffices; with NULL / too small a workspace the
 *                 product simply runs unsplit (same result to fp32 rounding, slower for those shapes).  The workspace
 *                 is only touched by kernels on `stream`: calls on one stream may share one.  The reference's
 *                 wrappers allocate, its launchers never do (ball_query.cpp:13-37) - same split here.           */
#define GB_PREC_F32 0
#define GB_PREC_BF16 1
/* GB_PREC_F32_SPLIT3 (round 5): an fp32 mode.  The tall products (gb_gemm_fwd, gb_gemm_fwd_pool, gb_gemm_dgrad on the
 * row-streaming kernel, gb_gemm_wgrad / gb_gemm_wgrad_gen3 on the register-direct kernel: from 65 536 rows / with a
 * device-side row count) run through the bf16 matrix cores as a three-way split:
 * every operand is cut EXACTLY into three 8-bit slices of its 24-bit mantissa and the six products of weight >= 2^-16 are
 * accumulated in fp32 - the error against fp64 is that of the fp32 MFMA (what is dropped is <= 2^-23 relative), the results
 * are not bit-identical to GB_PREC_F32's.  Every other product, and every shape no split instantiation fits, runs exactly
 * as under GB_PREC_F32.
 * Range (round 6, tests/test_gemm_gpu.py::test_split_products_on_extreme_operands): any FINITE operands of magnitude
 * >= 2^-110 give the fp32 bound whatever their spread (columns 120 binades apart: measured).  Below that the lower slices
 * fall into bf16's denormal range, which the matrix cores flush: the product is still right to 2^-15 at 2^-115 and to
 * 2^-8 at 2^-126 (1e-38; fp32 MFMA keeps its 5e-7 there).  NON-FINITE operands: an output that is +-inf or NaN under
 * GB_PREC_F32 is non-finite here too and vice versa, but it may be NaN where the fp32 product says +-inf - the slices of
 * inf are (inf, NaN, NaN), and an exact fix-up would need every output row / column with a non-finite operand recomputed
 * on the fp32 path; one BatchNorm later both are NaN in the reference's network and in this one.  Finite outputs next to
 * them meet the usual bound. */
#define GB_PREC_F32_SPLIT3 2
#define GB_GEMM_SCRATCH_BYTES (320ull * 64 * 128 * 4)
/*   rows_dev    : optional DEVICE pointer (8-byte aligned) to the actual row count of this call, 0 <= *rows_dev <= P.  The
 *                 P argument is then only the CAPACITY the caller sized its buffers (and the library its grid) for: the
 *                 kernels read the count themselves, so a caller whose row count is produced on the device (the
 *                 distinct rows of the cylinder crops, gb_cyl_unique / gb_cyl_rows) never has to read it back - no
 *                 host synchronisation, and the whole step can be captured in a HIP graph.  Rows [*rows_dev, P) are
 *                 neither read nor written.  Honoured by gb_gemm_fwd_gen3, gb_gemm_fwd_pool, gb_gemm_wgrad,
 *                 gb_gemm_wgrad_gen3, gb_gemm_dgrad_first_gen3 and by gb_gemm_fwd / gb_gemm_dgrad / gb_gemm_dgrad_first
 *                 on the row-streaming kernel (ask gb_gemm_uses_rs); every other case returns GB_EINVAL rather than
 *                 ignore it.                                                                                       */
/*   flags       : GB_GEMM_NO_RING - few-row fp32 products (gb_gemm_fwd / _dgrad / _wgrad) skip the LDS-DMA ring kernel
 *                 (csrc/gemm_ring.hip) and run on the register-staged tiles of csrc/gemm_cl.hip: an A/B and fallback
 *                 switch, results equal to fp32 rounding (a different summation order).                            */
/*                 GB_GEMM_NO_PAIR - gb_gemm_dgrad_wgrad issues its two products as two launches even where one
 *                 launch could carry both (A/B switch; same results up to the order of wgrad's fp32 atomics).        */
/*                 GB_GEMM_NO_DIRECT - tall fp32 weight gradients (gb_gemm_wgrad / gb_gemm_wgrad_gen3 from 65 536 rows or
 *                 with rows_dev) skip the register-direct kernel (csrc/gemm_wg.hip) and run on the LDS tiles of
 *                 csrc/gemm_cl.hip (A/B and fallback switch; same results up to the order of the fp32 additions).    */
#define GB_GEMM_NO_RING 1
#define GB_GEMM_NO_PAIR 2
#define GB_GEMM_NO_DIRECT 4
typedef struct GbGemmOpts {
  int precision;
  int reserved_cus;
  void *scratch;
  unsigned long long scratch_bytes;
  const long long *rows_dev;
  int flags;
} GbGemmOpts;

/* The arguments of gb_bn_finalize as a struct: entry points that produce BatchNorm sums take an optional pointer to
 * one and then finish the layer themselves (ab table, running statistics) with a gb_bn_finalize launch from the same
 * call - one host transition per layer instead of two.  training must be 1 (the evaluation-mode table does not depend
 * on a kernel's sums: call gb_bn_finalize).                                                                         */
typedef struct GbBnFinalize {
  const float *gamma, *beta;
  float *running_mean, *running_var; /* may be NULL */
  float *ab;                         /* [a, b, mean, rstd](C) */
  long long P;                       /* rows the sums stand for */
  float eps, momentum;
  int training;
} GbBnFinalize;

/* stats[0:C] += column sums of y (P,C), stats[C:2C] += column sums of y*y; fp64, caller zeroes.  fin: optional */
int gb_col_stats(const float *y, long long P, int C, double *stats, const GbBnFinalize *fin, void *stream);
/* The closing pass of a split forward product: y (P,C) = the chunk-ordered sum of `chunks` partial products (P*C floats
 * each, back to back in part), the column sums of y and y^2 into stats [2C] (caller-zeroed) in the same sweep and, with
 * fin, the BatchNorm finalisation.  C % 4 == 0, part / y 16-byte aligned (else GB_EINVAL). */
int gb_split_col_stats(const float *part, int chunks, float *y, long long P, int C, double *stats,
                       const GbBnFinalize *fin, void *stream);
/* ... and of a split dgrad product: dz = the chunk-ordered sum, and the BatchNorm-backward sums dstats [2C]
 * (caller-zeroed: sum g, sum g*xhat with g = dz*[a*y+b > 0]) of the ReLU layer with pre-BN output y and table ab. */
int gb_split_bn_bwd_stats(const float *part, int chunks, float *dz, const float *y, const float *ab, long long P, int C,
                          double *dstats, void *stream);
/* ab[0:C] = a = gamma*rstd, ab[C:2C] = b = beta - mean*a, ab[2C:3C] = mean, ab[3C:4C] = rstd.
 * `stats` is [slots][2C] (slot rows are summed; gb_col_stats fills one row, gb_gemm_fwd spreads its
 * epilogue atomics over `stat_slots` rows).
 * training: batch statistics from `stats` (biased variance) and running_* updated with `momentum`
 * (unbiased variance), as nn.BatchNorm does; otherwise the running statistics are used.           */
int gb_bn_finalize(const double *stats, int slots, long long P, int C, const float *gamma, const float *beta,
                   float eps, float momentum, float *running_mean, float *running_var, float *ab,
                   int training, void *stream);
/* z = act(a*y + b (+ residual)), act = ReLU when relu != 0 */
int gb_affine_act(const float *y, const float *ab, const float *residual, float *z, long long P, int C,
                  int relu, void *stream);
/* out[r,c] = max_k relu(a*y[r*ns+k,c] + b), arg[r,c] = first k attaining it;  R = P/ns rows */
int gb_affine_relu_maxpool(const float *y, const float *ab, float *out, int32_t *arg, long long R, int ns,
                           int C, void *stream);
/* BatchNorm(+ReLU)(+residual) backward. dstats[0:C] += dbeta, dstats[C:2C] += dgamma (fp64, zeroed by
 * the caller); then dy = a*(dA - dbeta/P - xhat*dgamma/P) (training) or a*dA (eval); dres (optional)
 * receives dA = dout*[z>0].  The *_pool forms take the (R,C) gradient of the max-pooled output.
 * dbeta / dgamma (optional, both or none): gb_bn_bwd_reduce runs from the same call (one host transition less). */
int gb_bn_bwd_stats(const float *dout, const float *y, const float *ab, const float *residual, long long P,
                    int C, int relu, double *dstats, float *dbeta, float *dgamma, void *stream);
int gb_bn_bwd_apply(const float *dout, const float *y, const float *ab, const float *residual,
                    const double *dstats, long long P, int C, int relu, int training, float *dy, float *dres,
                    void *stream);
/* gb_bn_bwd_apply that also writes the layer's parameter gradients dbeta (C), dgamma (C) in fp32 from the same fp64 sums
 * (dstats must then be the totals, i.e. come from ONE slot row): what a separate gb_bn_bwd_reduce launch would do. */
int gb_bn_bwd_apply_g(const float *dout, const float *y, const float *ab, const float *residual, const double *dstats,
                      long long P, int C, int relu, int training, float *dy, float *dres, float *dbeta, float *dgamma,
                      void *stream);
int gb_bn_bwd_stats_pool(const float *dout, const float *out, const int32_t *arg, const float *y,
                         const float *ab, long long R, int ns, int C, double *dstats, float *dbeta, float *dgamma,
                         void *stream);
int gb_bn_bwd_apply_pool(const float *dout, const float *out, const int32_t *arg, const float *y,
                         const float *ab, const double *dstats, long long R, int ns, int C, int training,
                         float *dy, void *stream);

/* Sum the [slots][2C] fp64 partial BatchNorm-backward sums a GEMM epilogue (gb_gemm_dgrad) or
 * gb_bn_bwd_stats left: dstats (optional, fp64 [2C]) = the totals in the form gb_bn_bwd_apply reads,
 * dbeta / dgamma (fp32 [C]) = the BatchNorm parameter gradients (torch: batch_norm_backward's
 * grad_bias / grad_weight, reached by the reference through nn.BatchNorm2d, pytorch_utils.py:74-78). */
int gb_bn_bwd_reduce(const double *dst, int slots, int C, double *dstats, float *dbeta, float *dgamma,
                     void *stream);

/* ---- distinct rows of the nested cylinder crops (csrc/cyl_rows.hip, *_members kernels in csrc/mlp_cl.hip) -----
 * Reference: TrainModel/modules.py:99-124 (CloudCrop / GraspWidthGrouping: D CylinderQueryAndGroup crops per seed,
 * hmax = 0.01..0.04, stacked and sent through ONE SharedMLP, then max_pool2d over the samples of each crop).  A point
 * lying in several of a seed's cylinders gives identical rows; the MLP runs on the distinct rows, weighted by their
 * multiplicity in the BatchNorm sums, and each crop's max runs over its members.                              */
/* nr query sets at once (the radii of stage 2): idx (nr, D, R, ns) int32 (R = b*m seeds).  Per set and seed: sorted
 * (nr, R, D*ns) = its distinct point ids, ascending, compacted to the front; meta (nr, R, D*ns) = (multiplicity << 8) |
 * member bits (bit d: in crop d); count (nr, R).  D*ns <= 256.                                                     */
int gb_cyl_unique(const int32_t *idx, int nr, int D, long long R, int ns, int32_t *sorted, int32_t *meta, int32_t *count,
                  void *stream);
/* off (nr, R) int64 = exclusive prefix sums of count (nr, R) along R; total (nr) int64 = the row count P_u of each set,
 * ON THE DEVICE: what GbGemmOpts.rows_dev / the rows_dev arguments take, so that no caller has to read it back.   */
int gb_cyl_scan(const int32_t *count, int nr, long long R, int64_t *off, long long *total, void *stream);
/* Rows off[r] .. off[r]+count[r]-1 of seed r, for each of the nr sets at row stride `cap` (the caller's row capacity per
 * set, a multiple of 32, >= the set's total; rows beyond it are dropped, never written out of bounds):
 * x0 (nr, cap, 3) = (xyz[b,id] - centre[r]) rotated by rot[r] (3x3) as gb_group_concat_cl mode 2, row_w (nr, cap) =
 * multiplicity (row_w16: the same as uint16, for gb_gemm_fwd_w), row_mem = member bits, row_key (optional) = (seed << 13)
 * | (multiplicity << 4) | member bits for gb_gemm_fwd_pool (D <= 4).  row_w16 / row_key are zero on [P_u, P_u rounded up
 * to 32).  W = D*ns is the row pitch of sorted / meta.                                                              */
int gb_cyl_rows(const float *xyz, const float *centres, const float *rot, const int32_t *sorted, const int32_t *meta,
                const int32_t *count, const int64_t *off, int nr, int b, int n, int m, int W, long long cap, float *x0,
                float *row_w, uint16_t *row_w16, int32_t *row_mem, int32_t *row_key, void *stream);
/* The LAST layer of a crop stack without ever storing its output (reference modules.py:104-124: the SharedMLP's final
 * conv + BatchNorm + ReLU, then max_pool2d over each crop; pointnet2_modules.py:176-188 has the same shape).
 * gb_gemm_fwd_pool: Y = f(X) W^T is formed tile by tile; what leaves the kernel are the weighted BatchNorm sums
 * (as gb_gemm_fwd_w) and, per (32-row tile t, seed r with rows in t, crop d, column c), the extreme of sign(gamma_c)*y
 * over the seed's member rows in the tile: pairs[((t + r)*D + d)*N + c] - a seed's rows are contiguous, so slot
 * t + r is unique and is written with plain stores.  relu(a*y + b) is monotone in y with the sign of a = gamma*rstd,
 * so gb_pool_pairs finishes the pooling once the statistics are known: out ((R*D), N) = the crop's max of
 * relu(a*y + b), ystar = the y attaining it (0 / 0 for a crop or a seed without rows).  row_key (P rounded up to 32,
 * zero tail; 16-byte aligned) from gb_cyl_rows, seed ids < seeds; pairs: pairs_elems >= ((P + 31) / 32 + seeds) * D * N
 * floats (else GB_ERANGE).  N in {64, 128, 160, 256}, D <= 4, P >= 16384, K % 4 == 0: otherwise GB_EINVAL (ask
 * gb_gemm_uses_rs(P, K, N, 0, 3, has_aff)).  stats as in gb_gemm_fwd (required).  y (optional, (P,N)): Y is stored as
 * well, for a caller whose backward wants it: gb_bn_bwd_apply_members_v finds the arg-max rows in it by value
 * (y == ystar).  y = NULL: a forward-only caller (inference) - the layer's output is neither stored nor traceable.  */
int gb_gemm_fwd_pool(const float *x, const float *w, const float *aff, const int32_t *row_key, const float *gamma,
                     float *pairs, long long pairs_elems, long long seeds, float *y, double *stats, int stat_slots,
                     long long P, int K, int N, int D, const GbBnFinalize *fin, const GbGemmOpts *opts, void *stream);
int gb_pool_pairs(const float *pairs, const int64_t *off, const int32_t *cnt, const float *ab, const float *gamma,
                  float *out, float *ystar, long long R, int D, int C, void *stream);
/* gb_gemm_fwd whose BatchNorm sums weight row p by row_w16[p] (uint16 multiplicities; the array must extend,
 * zero-filled, to the next multiple of 32 rows).                                                              */
int gb_gemm_fwd_w(const float *x, const float *w, const float *aff, const uint16_t *row_w16, float *y, double *stats,
                  int stat_slots, long long P, int K, int N, const GbBnFinalize *fin, const GbGemmOpts *opts,
                  void *stream);
/* out ((R*D), C) [row r*D + d] = max over the rows of seed r with member bit d of relu(a*y + b); arg = absolute
 * row index of the maximum.  D <= 4; C % 4 == 0.                                                               */
int gb_affine_relu_maxpool_members(const float *y, const float *ab, const int32_t *row_mem, const int64_t *off,
                                   const int32_t *cnt, float *out, int32_t *arg, long long R, int D, int C,
                                   void *stream);
/* Backward of BatchNorm + ReLU + member max-pool onto the distinct rows: dy[u] = a*(g_u - w_u*dbeta/P - xhat_u*
 * w_u*dgamma/P), g_u = sum over the crops whose arg-max is u of dout*[out > 0]; dstats = [dbeta, dgamma] sums from
 * gb_bn_bwd_stats_pool(..., ns = 0: arg is an absolute row index); P_total = rows of the original batch.        */
int gb_bn_bwd_apply_members(const float *dout, const float *out, const int32_t *arg, const float *y, const float *ab,
                            const double *dstats, const float *row_w, const int64_t *off, const int32_t *cnt,
                            long long R, int D, int C, long long P_total, int training, float *dy, void *stream);
/* ... with the arg-max rows found by value: crop d's gradient goes to the FIRST row of the seed (in row order) that is a
 * member of d (row_mem bit d) and whose y equals ystar[(r*D + d), c] - what gb_pool_pairs leaves after a values-only
 * gb_gemm_fwd_pool.                                                                                              */
int gb_bn_bwd_apply_members_v(const float *dout, const float *out, const float *ystar, const float *y, const float *ab,
                              const double *dstats, const float *row_w, const int32_t *row_mem, const int64_t *off,
                              const int32_t *cnt, long long R, int D, int C, long long P_total, int training, float *dy,
                              void *stream);
/* gb_bn_bwd_apply (ReLU, no residual) for rows with multiplicities: dy = a*(dA*[z>0] - w*dbeta/P - xhat*w*dgamma/P).
 * rows_dev (optional, device, 8-byte aligned): the actual row count when `rows` is the caller's capacity
 * (GbGemmOpts.rows_dev has the story).                                                                            */
int gb_bn_bwd_apply_w(const float *dout, const float *y, const float *ab, const double *dstats, const float *row_w,
                      long long rows, long long P_total, int C, int training, float *dy, const long long *rows_dev,
                      void *stream);

/* ---- LocalAggregation without the grouped tensor (csrc/local_agg.hip) ---------------------------------
 * Reference: TrainModel/drp.py:32-67 (LocalAggregation.forward :62 = QueryAndGroup -> [dp, fj] ->
 * create_convblock2d (1x1 conv, BatchNorm2d, ReLU) -> max over the ns neighbours), reached from
 * InvResMLP.forward drp.py:109-117.  The 1x1 conv commutes with the gather:
 *   y[p,c] = G[idx(p),c] + dp_p . Wx[c],   G = f Wf^T on the B*n points,  Wx = W[:, :3], Wf = W[:, 3:],
 * so only per-point sums of the grouping are needed besides G (see the header of local_agg.hip).
 * Geometry arguments as gb_group_concat_cl: xyz (b,n,3), centres (b,m,3), idx (b,m,ns) int32 into n,
 * mode 0: dp = xyz[idx] - centre; mode 1: that times `scale`.                                          */
/* cnt (b*n) += number of rows referencing each point, dsum (b*n,3) += sum of their dp,
 * mom fp64 [12] += [S = sum_p dp (3), M = sum_p dp dp^T (3x3)]; all caller-zeroed.                      */
int gb_la_point_stats(const float *xyz, const float *centres, const int32_t *idx, int b, int n, int m, int ns,
                      int mode, float scale, float *cnt, float *dsum, double *mom, void *stream);
/* stats fp64 [2C] += [sum_p y, sum_p y^2] (the BatchNorm batch sums over all P = b*m*ns rows, in the form
 * gb_bn_finalize reads), u fp64 [3][C] += U[j][c] = sum_i G[i,c] D_i[j]; G (rows = b*n, C); caller-zeroed. */
/* W (N, 3+C) -> Wx (N,3), Wf (N,C) (dense copies of the two column blocks of the LocalAggregation conv weight,
 * drp.py:32-67: the xyz part and the feature part) in one launch; gb_la_join_w is the inverse (for the gradients). */
int gb_la_split_w(const float *w, float *wx, float *wf, int N, int C, void *stream);
int gb_la_join_w(const float *wx, const float *wf, float *w, int N, int C, void *stream);
int gb_la_col_stats(const float *G, const float *cnt, const float *dsum, const float *wx, const double *mom,
                    long long rows, int C, double *stats, double *u, const GbBnFinalize *fin,
                    void *stream);
/* out (b*m, C) = max_k relu(a y + b), arg = first k attaining it; ab = [a, b, mean, rstd](C) from
 * gb_bn_finalize.  C % 4 == 0, 16 <= C <= 1024, ns <= 64.                                               */
int gb_la_pool(const float *G, const float *xyz, const float *centres, const int32_t *idx, const float *wx,
               const float *ab, float *out, int32_t *arg, int b, int n, int m, int ns, int C, int mode,
               float scale, void *stream);
/* Backward of pool + ReLU: g = dout*[out > 0] goes to the arg-max row only.  sg (b*n, C) += g at that row's
 * point (caller-zeroed), red fp64 [5][C] += column sums of [g, g*xhat, g*dp0, g*dp1, g*dp2] (caller-zeroed;
 * red[0] = dbeta, red[1] = dgamma).                                                                     */
int gb_la_pool_bwd(const float *dout, const float *out, const int32_t *arg, const float *G, const float *xyz,
                   const float *centres, const int32_t *idx, const float *wx, const float *ab, float *sg,
                   double *red, int b, int n, int m, int ns, int C, int mode, float scale, void *stream);
/* gb_la_pool_bwd with the scatter pre-aggregated in LDS: perm (b,m) int32 = a spatially coherent order of each cloud's
 * centres (gb_fps_row_order(centres, perm, b, m)); a workgroup takes 16 consecutive entries, gives the few distinct winner
 * points of those rows a slot each and adds one dense row of atomics per point instead of one scattered atomic per
 * (row, column).  Same sums up to the order of the fp32 additions.  perm = NULL, m % 16 != 0 or a C that does not divide
 * 256: exactly gb_la_pool_bwd.                                                                                    */
int gb_la_pool_bwd_perm(const float *dout, const float *out, const int32_t *arg, const float *G, const float *xyz,
                        const float *centres, const int32_t *idx, const float *wx, const float *ab, const int32_t *perm,
                        float *sg, double *red, int b, int n, int m, int ns, int C, int mode, float scale, void *stream);
/* dG (rows = b*n, C): gradient of G through BatchNorm (training: batch statistics over P rows; else a*sg). */
int gb_la_point_grad(const float *sg, const float *G, const float *cnt, const float *dsum, const float *wx,
                     const float *ab, const double *red, long long P, long long rows, int C, int training,
                     float *dG, void *stream);
/* dwx (C,3): gradient of Wx.                                                                            */
int gb_la_wx_grad(const double *red, const double *u, const double *mom, const float *wx, const float *ab,
                  long long P, int C, int training, float *dwx, void *stream);
/* ... and the BatchNorm parameter gradients dbeta (C) = red[0:C], dgamma (C) = red[C:2C] in fp32 from the same launch;
 * red may be `slots` rows of [5][C] partial sums (a dgrad epilogue's slot rows), added in slot order. */
int gb_la_wx_grad_g(const double *red, int slots, const double *u, const double *mom, const float *wx, const float *ab,
                    long long P, int C, int training, float *dwx, float *dbeta, float *dgamma, void *stream);
/* ... with dwx inside a wider matrix: row c at dwx + c * ldw, ldw >= 3 (ABI v7): the xyz columns of the joined
 * (C, 3 + Cf) gradient of LocalAggregation's conv (drp.py:32-67), whose feature columns gb_gemm_wgrad_group adds into. */
int gb_la_wx_grad_gs(const double *red, int slots, const double *u, const double *mom, const float *wx, const float *ab,
                     long long P, int C, int training, float *dwx, int ldw, float *dbeta, float *dgamma, void *stream);

/* ---- fp32 MFMA GEMMs of the channel-last SharedMLP (csrc/gemm_cl.hip) — replace the cuBLAS/cuDNN
 * 1x1 convolutions the reference reaches through torch (pytorch_utils.py:61-113) ------------------- */
/* Y (P,N) = f(X (P,K)) W(N,K)^T.  aff (optional) = [a(K), b(K)]: f(x) = relu(a_k x + b_k), i.e. the
 * previous layer's BatchNorm + ReLU applied while loading.  stats (optional, fp64 [stat_slots][2N],
 * caller-zeroed) += column sums and sums of squares of Y (BatchNorm batch statistics), spread over the
 * slot rows to avoid same-address atomic contention; gb_bn_finalize sums the rows.
 * Reproducibility: Y holds the same bits on every call with the same options (gb_gemm_dgrad's dX likewise; a split
 * reduction adds its chunks in a fixed order).  gb_gemm_wgrad accumulates with fp32 atomics: last-bit differences
 * between runs.                                                                                                 */
int gb_gemm_fwd(const float *x, const float *w, const float *aff, float *y, double *stats, int stat_slots,
                long long P, int K, int N, const GbBnFinalize *fin, const GbGemmOpts *opts, void *stream);
/* dX (P,K) = dY (P,N) W(N,K), W in its natural (N,K) row-major layout.  Optional fused BatchNorm-backward
 * statistics of the previous layer (dX is the gradient of its post-ReLU output): y_prev (P,K) its pre-BN
 * output, ab_prev = [a,b,mean,rstd](K), dstats fp64 [stat_slots][2K] (caller-zeroed) += [sum dA,
 * sum dA*xhat] with dA = dX*[a*y+b > 0].  Pass NULLs / 0 to skip.  dbeta / dgamma (optional): the
 * previous layer's gb_bn_bwd_reduce runs from the same call, the slot rows' total going to dstats_total. */
int gb_gemm_dgrad(const float *dy, const float *w, float *dx, const float *y_prev, const float *ab_prev,
                  double *dstats, int stat_slots, long long P, int K, int N, double *dstats_total, float *dbeta,
                  float *dgamma, const GbGemmOpts *opts, void *stream);
/* dW (N,K) += dY (P,N)^T f(X (P,K)); accumulates (fp32 atomics, reduction over P split across
 * workgroups).  x_aff (optional) = [a(K), b(K)]: f(x) = relu(a_k x + b_k), as in gb_gemm_fwd.      */
int gb_gemm_wgrad(const float *dy, const float *x, const float *x_aff, float *dw, long long P, int K, int N,
                  const GbGemmOpts *opts, void *stream);
/* Both gradient products of ONE layer, which share dY and do not depend on each other (torch autograd issues them as
 * two cuBLAS calls behind the 1x1 convolutions of pytorch_utils.py:61-113; the reference's stacks that take this form:
 * pointnet2_modules.py:176-188, modules.py:104-124, drp.py:97-117): dX = dY W with the optional fused BatchNorm-backward
 * sums exactly as gb_gemm_dgrad, dW += dY^T f(X) exactly as gb_gemm_wgrad.  Few-row products that each fill about half
 * of the chip (64x64 tiles of the LDS-DMA ring kernel, csrc/gemm_ring.hip) leave as ONE launch whose workgroups are
 * resident together; every other shape runs gb_gemm_wgrad then gb_gemm_dgrad.  rows_dev: as the single entries.  */
int gb_gemm_dgrad_wgrad(const float *dy, const float *w, float *dx, const float *y_prev, const float *ab_prev,
                        double *dstats, int stat_slots, long long P, int K, int N, double *dstats_total, float *dbeta,
                        float *dgamma, const float *x, const float *x_aff, float *dw, const GbGemmOpts *opts,
                        void *stream);
/* Many weight gradients in ONE call (ABI v7, round 6).  A weight gradient reads its layer's stored input and the gradient
 * the layer's backward has formed and nothing depends on it until the optimizer: the reference issues each as its own
 * cuBLAS call in the middle of backward (the 1x1 convolutions of pytorch_utils.py:61-113 under drp.py:97-117,
 * pointnet2_modules.py:402-435, modules.py:49-175); a caller of this library may instead record them and hand them over
 * together.  Item i: dW_i (N, ldw >= K; caller-zeroed, accumulated with fp32 atomics) += dY_i (P,N)^T f(X_i (P,K)), x_aff
 * as gb_gemm_wgrad.  The few-row products (P <= 131 072 rows, P % 32 == 0, K and N multiples of 4, 16-byte aligned dY / X)
 * run as one grid per 63 items on the LDS-DMA ring kernel's tiles (csrc/gemm_ring.hip); every other item runs exactly
 * as its own gb_gemm_wgrad call and needs ldw == K (GB_EINVAL otherwise).  `items` is HOST memory, read during the
 * call only.  opts: precision / reserved_cus / flags as the single entry; rows_dev is not supported (GB_EINVAL).      */
typedef struct GbWgradItem {
  const float *dy, *x, *x_aff;
  float *dw;
  long long P;
  int K, N, ldw;
} GbWgradItem;
int gb_gemm_wgrad_group(const GbWgradItem *items, int count, const GbGemmOpts *opts, void *stream);
/* 1 when gb_gemm_wgrad_group runs a product of this shape (16-byte aligned operands) inside a grouped launch - the case
 * that accepts ldw != K -, 0 when it issues it as a single gb_gemm_wgrad.  Host-side introspection, no launch.       */
int gb_gemm_wgrad_groups(long long P, int K, int N, int precision, int reserved_cus, unsigned flags);
/* Which kernel gb_gemm_fwd (dgrad = 0) / gb_gemm_dgrad (dgrad = 1) launches for 16-byte aligned operands of
 * this shape: 1 = the row-streaming kernel CAN run it (csrc/gemm_rs.hip: what the generated-operand, pooled and
 * device-row-count entries require), 0 = it cannot.  The plain gb_gemm_fwd / gb_gemm_dgrad additionally prefer the tiled
 * kernels below 65 536 rows (gb_gemm_kernel_for says what actually launches).
 * Pure host-side introspection (no launch), used by bench.py to attribute timings per kernel.      */
int gb_gemm_uses_rs(long long P, int K, int N, int dgrad, int fused_stats, int has_aff);
/* ... and, for all three products (kind 0 = gb_gemm_fwd, 1 = gb_gemm_dgrad, 2 = gb_gemm_wgrad; fp32, default options):
 * 0 = register-staged LDS tiles (csrc/gemm_cl.hip), 1 = row-streaming (csrc/gemm_rs.hip), 2 = LDS-DMA ring
 * (csrc/gemm_ring.hip: the few-row products), 3 = the column-reduction wgrad for <= 4 input channels, 4 = the
 * register-direct tall wgrad (csrc/gemm_wg.hip).  kind 3 = gb_gemm_dgrad_wgrad: 2 when ONE launch of the ring kernel
 * carries both products, 0 when the call issues the two single products.                                          */
int gb_gemm_kernel_for(int kind, long long P, int K, int N, int fused_stats, int has_aff);
/* ... the same for a call that carries GbGemmOpts {precision, reserved_cus, flags} (ABI v7): the
 * register-direct wgrad cuts a product with the wave count and grid of the skeleton the PRECISION selects (fp32 MFMA:
 * 8 waves; bf16 and GB_PREC_F32_SPLIT3: 4 waves, the grid clamped by the rows) and the A/B flags veto kernels, so the
 * answer for the default options can differ from the kernel a call with other options launches.  gb_gemm_kernel_for
 * is this function at (GB_PREC_F32, 0, 0).                                                                          */
int gb_gemm_kernel_for2(int kind, long long P, int K, int N, int fused_stats, int has_aff, int precision,
                        int reserved_cus, unsigned flags);
/* dgrad into the first layer of a stack whose input x_in (P,3) has 3 channels: dZ = dY (P,N) W (N,K) is formed
 * but not stored; sums fp64 [slots][5K] (caller-zeroed) += column sums of [g, g*xhat, g*x_0, g*x_1, g*x_2] with
 * g = dZ*[a*y+b > 0], xhat = (y - mean)*rstd, y = y_prev (P,K) the layer's pre-BatchNorm output, ab_prev =
 * [a,b,mean,rstd](K).  The layer's dgamma/dbeta are the first two sums; its weight gradient follows from the sums
 * and the moments of x_in (gb_moments3) by gb_la_wx_grad with u = 0.  GB_EINVAL when the shape is not eligible:
 * ask gb_gemm_uses_rs(P, K, N, 1, 2, 0) first.                                                          */
int gb_gemm_dgrad_first(const float *dy, const float *w, const float *y_prev, const float *ab_prev, const float *x_in,
                        double *sums, int slots, long long P, int K, int N, const GbGemmOpts *opts, void *stream);
/* The xyz-only FIRST layer of a stack (3 -> K conv + BatchNorm + ReLU: pytorch_utils.py:61-113 under
 * pointnet2_modules.py:148-188 / modules.py:104-124) folded into its consumers: its output Y1 = x0 W1^T (P,K) is never
 * written or read - every consumer re-forms y1 = ((x*w0) + (y*w1)) + (z*w2) from the row's 12 bytes.
 *   gb_bn_finalize_lin3      : the layer's BatchNorm table ab = [a,b,mean,rstd](K) and running statistics from the 12
 *                              moments of x0 (gb_moments3): sum y = W1 s, sum y^2 = diag(W1 M W1^T), in fp64
 *   gb_gemm_fwd_gen3         : the second layer, Y (P,N) = relu(a1*y1 + b1) W^T (+ BatchNorm sums, row weights as
 *                              gb_gemm_fwd_w); ab1 = [a1(K), b1(K)]
 *   gb_gemm_wgrad_gen3       : its weight gradient, dW (N,K) += dY^T relu(a1*y1 + b1)
 *   gb_gemm_dgrad_first_gen3 : gb_gemm_dgrad_first with y_prev re-formed from x_in and w_in = W1
 * All three GEMM forms: GB_EINVAL when the row-streaming kernel does not take the shape (ask gb_gemm_uses_rs).      */
int gb_bn_finalize_lin3(const double *mom, const float *w, long long P, int C, const float *gamma, const float *beta,
                        float eps, float momentum, float *running_mean, float *running_var, float *ab, void *stream);
int gb_gemm_fwd_gen3(const float *x0, const float *w1, const float *ab1, const float *w, const uint16_t *row_w16,
                     float *y, double *stats, int stat_slots, long long P, int K, int N, const GbBnFinalize *fin,
                     const GbGemmOpts *opts, void *stream);
int gb_gemm_wgrad_gen3(const float *dy, const float *x0, const float *w1, const float *ab1, float *dw, long long P,
                       int K, int N, const GbGemmOpts *opts, void *stream);
int gb_gemm_dgrad_first_gen3(const float *dy, const float *w, const float *ab_prev, const float *x_in,
                             const float *w_in, double *sums, int slots, long long P, int K, int N,
                             const GbGemmOpts *opts, void *stream);
/* mom fp64 [12] (caller-zeroed) += [sum_p w_p x (3), sum_p w_p x x^T (3x3)] of x (P,3); w = row_w (P) or 1.
 * rows_dev (optional, device): the actual row count, <= P (see GbGemmOpts.rows_dev).                           */
int gb_moments3(const float *x, const float *row_w, long long P, double *mom, const long long *rows_dev, void *stream);

/* ---- the training loss (TrainModel/loss.py:44-179) ------------------------------------------------------------
 * gb_grasp_loss_fwd: every term of get_loss in three launches.  Tensors as the reference holds them: obj_score
 * (B,2,Ns), view_score / view_label (B,Ns,V), obj_label (B,Ns) int64 = objectness label of each seed, weight (B,Ns)
 * = generate_reweight_mask (:29-42), labels / offsets / tolerance = the top-view labels (B,Ns,A,D[,3]), the four
 * predictions (B,A,Ns,D).  obj_score and the predictions may be channel slices of wider tensors: batch_strides (HOST
 * array of 5: obj_score, score, angle, width, tolerance) gives their batch strides in elements, the inner dimensions
 * are dense.  D <= 8.  With view_arg != NULL the seed weights are formed in the kernel instead of read from `weight`:
 * view_arg (B,Ns,V) int32 from gb_label_finish, offsets_all (B,Ns,V,A,D,3), the prior's nb+1 ascending bin edges and nb
 * bin weights (ScalePrior of loss.py:18-42).  Workspace / results: partial (B*Ns*20), aux (B*Ns*20), den (3) floats; graspable (B,Ns) int64 =
 * `graspable_mask`; out (14) = [overall, objectness, view, score, angle, width, tolerance losses, graspable acc /
 * prec / recall, positive-view count, angle accuracy at 0 / 15 / 30 degrees].
 * gb_grasp_loss_bwd: grad_out (7) = gradients of out[0..6]; writes the dense gradients d_obj (B,2,Ns), d_view
 * (B,Ns,V), d_score / d_angle / d_width / d_tol (B,A,Ns,D) in full.                                                 */
int gb_grasp_loss_fwd(const float *obj_score, const float *view_score, const float *view_label,
                      const int64_t *obj_label, const float *weight, const float *labels, const float *offsets,
                      const float *tolerance, const float *score_pred, const float *angle_pred,
                      const float *width_pred, const float *tol_pred, const long long *batch_strides,
                      const int32_t *view_arg, const float *offsets_all, const float *edges, const float *prior_w, int nb,
                      int B, int Ns, int V, int A, int D, float thresh_bad, float thresh_good, float max_width, float max_tol,
                      float *partial, float *aux, int64_t *graspable, float *out, float *den, void *stream);
int gb_grasp_loss_bwd(const float *obj_score, const float *view_score, const float *view_label,
                      const int64_t *obj_label, const float *weight, const float *labels, const float *offsets,
                      const float *tolerance, const float *score_pred, const float *angle_pred,
                      const float *width_pred, const float *tol_pred, const long long *batch_strides,
                      const int32_t *view_arg, const float *offsets_all, const float *edges, const float *prior_w, int nb,
                      int B, int Ns, int V, int A, int D, float thresh_bad, float thresh_good, float max_width, float max_tol,
                      const float *aux, const int64_t *graspable, const float *den, const float *grad_out,
                      float *d_obj, float *d_view, float *d_score, float *d_angle, float *d_width, float *d_tol,
                      void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GRASPBAL_H */

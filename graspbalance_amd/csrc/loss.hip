// The training loss of the reference (TrainModel/loss.py: compute_robust_graspable_loss :55-78,
// compute_weighted_view_loss :80-116, compute_weighted_grasp_loss :118-179, get_loss :44-53) as three launches
// instead of ~180 element-wise / reduction launches forward and as many backward: every term is a masked mean over
// the B*Ns seeds, so one wave per seed forms the seed's contributions, a single workgroup adds them up in fp64, and
// the backward pass (one wave per seed again) writes the gradients of the six prediction tensors in full.
//   total = CE(objectness) + view MSE (scale-reweighted) + 0.2 (score Huber + angle CE + width Huber + tolerance Huber)
#include "gb_common.h"

namespace gb {

constexpr int LS_NP = 20;   // per-seed partial sums
constexpr int LS_AUX = 20;  // per-seed values kept for the backward pass: [mw, dm, lm[0..7], best[0..7], -, -]
constexpr int LS_MAXD = 8;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__device__ __forceinline__ float huber1(float e) {  // loss_utils.huber_loss with delta = 1
  const float a = fabsf(e);
  const float q = fminf(a, 1.0f);
  return 0.5f * q * q + (a - q);
}

struct LossArgs {
  const float *obj_score;   // (B,2,Ns)
  const float *view_score;  // (B,Ns,V)
  const float *view_label;  // (B,Ns,V)
  const int64_t *obj_label; // (B,Ns)   objectness label of the seed
  const float *weight;      // (B,Ns)   scale-prior weight of the seed
  const float *labels;      // (B,Ns,A,D)   top-view grasp scores
  const float *offsets;     // (B,Ns,A,D,3) top-view (angle, depth, width)
  const float *tolerance;   // (B,Ns,A,D)
  const float *score_pred, *angle_pred, *width_pred, *tol_pred;  // (B,A,Ns,D)
  // optional (view_arg != nullptr): the seed weight is formed here instead of read from `weight` - the reference's
  // generate_reweight_mask (loss.py:29-42): width of the best grasp over ALL views of the seed, binned by the prior
  const int32_t *view_arg;   // (B,Ns,V)  position in [0, A*D) of each view's best label (gb_label_finish)
  const float *offsets_all;  // (B,Ns,V,A,D,3)
  const float *edges;        // (nb+1) bin edges, ascending
  const float *prior_w;      // (nb) bin weights
  int nb;
  int B, Ns, V, A, D;
  long long bs_obj, bs_score, bs_angle, bs_width, bs_tol;  // batch strides (elements) of obj_score and the four predictions
  float thresh_bad, thresh_good, max_width, max_tol;
};

// per-depth work of one seed (lanes 0..D-1): arg-max angle of the labels, the targets, the masks
struct DepthVals {
  int best;
  float tlab, twid, ttol, lm;
};
__device__ __forceinline__ DepthVals depth_targets(const LossArgs &g, long long s, int d, bool obj, float w) {
  DepthVals r;
  const float *tl = g.labels + s * g.A * g.D;
  float bv = tl[d];
  int ba = 0;
  for (int a = 1; a < g.A; ++a) {
    const float v = tl[a * g.D + d];
    if (v > bv) { bv = v; ba = a; }  // first maximum (torch.argmax)
  }
  r.best = ba;
  r.tlab = bv;
  r.twid = g.offsets[((s * g.A + ba) * g.D + d) * 3 + 2];
  r.ttol = g.tolerance[(s * g.A + ba) * g.D + d];
  r.lm = (obj && bv > g.thresh_bad) ? w : 0.f;
  return r;
}

// one wave per seed
__global__ __launch_bounds__(256) void loss_seed_kernel(LossArgs g, float *__restrict__ partial, float *__restrict__ aux,
                                                         int64_t *__restrict__ graspable_out) {
  const int lane = threadIdx.x & 63;
  const long long s = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= (long long)g.B * g.Ns) return;
  const int b = (int)(s / g.Ns), i = (int)(s % g.Ns);
  // ---- views
  const float *vs = g.view_score + s * g.V, *vl = g.view_label + s * g.V;
  float cnt = 0.f, sq = 0.f, pos = 0.f;
  float bestv = -INFINITY;
  int besti = 0x7fffffff;
  for (int v = lane; v < g.V; v += 64) {
    const float a = vs[v], l = vl[v];
    if (l > bestv) { bestv = l; besti = v; }  // first maximum among this lane's views
    cnt += l > g.thresh_bad ? 1.f : 0.f;
    const float df = a - l;
    sq += df * df;
    pos += a >= g.thresh_good ? 1.f : 0.f;
  }
  cnt = wave_sum(cnt);
  sq = wave_sum(sq);
  pos = wave_sum(pos);
  const long long ol = g.obj_label[s];
  const long long glabel = cnt > 10.f ? ol : 0;
  float w;
  if (g.view_arg) {  // wave-uniform
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {  // arg-max over the views: largest value, then the lowest view index
      const float ov = __shfl_xor(bestv, off);
      const int oi = __shfl_xor(besti, off);
      if (ov > bestv || (ov == bestv && oi < besti)) { bestv = ov; besti = oi; }
    }
    const long long vrow = s * g.V + besti;
    const float width = g.offsets_all[(vrow * g.A * g.D + g.view_arg[vrow]) * 3 + 2];
    int nlt = 0;  // torch.bucketize(width, edges): the number of edges below it
    for (int k = 0; k <= g.nb; ++k) nlt += g.edges[k] < width ? 1 : 0;
    const bool inside = nlt >= 1 && nlt <= g.nb && width != g.edges[nlt < g.nb ? nlt : g.nb];
    w = g.prior_w[inside ? nlt - 1 : 0];
  } else {
    w = g.weight[s];
  }
  const bool vmask = glabel * ol > 0;
  const float mw = vmask ? w : 0.f;
  // ---- objectness cross entropy (2 classes)
  const float z0 = g.obj_score[b * g.bs_obj + i], z1 = g.obj_score[b * g.bs_obj + g.Ns + i];
  const float zm = fmaxf(z0, z1);
  const float lse = zm + logf(expf(z0 - zm) + expf(z1 - zm));
  const float ce = lse - (glabel == 1 ? z1 : z0);
  const int pred = z1 > z0 ? 1 : 0;
  const bool correct = pred == glabel;
  // ---- grasp terms of the top view: lane d < D owns depth d
  const bool obj = ol != 0;
  DepthVals t = {0, 0.f, 0.f, 0.f, 0.f};
  float h_score = 0.f, ce_ang = 0.f, h_wid = 0.f, h_tol = 0.f, sel = 0.f, a0 = 0.f, a15 = 0.f, a30 = 0.f;
  if (lane < g.D) {
    const int d = lane;
    t = depth_targets(g, s, d, obj, w);
    const size_t po = (size_t)i * g.D + d, pa = (size_t)g.Ns * g.D;  // element (b,a,i,d) at b*bs + a*pa + po
    const float *ang = g.angle_pred + b * g.bs_angle + po;
    h_score = huber1(g.score_pred[b * g.bs_score + t.best * pa + po] - t.tlab);
    float am = ang[0];
    int ap = 0;
    for (int a = 1; a < g.A; ++a) {
      const float z = ang[a * pa];
      if (z > am) { am = z; ap = a; }
    }
    float se = 0.f;
    for (int a = 0; a < g.A; ++a) se += expf(ang[a * pa] - am);
    ce_ang = (am + logf(se)) - ang[t.best * pa];
    h_wid = huber1((g.width_pred[b * g.bs_width + t.best * pa + po] - t.twid) / g.max_width);
    h_tol = huber1((g.tol_pred[b * g.bs_tol + t.best * pa + po] - t.ttol) / g.max_tol);
    if (t.lm != 0.f) {
      const int df = ap > t.best ? ap - t.best : t.best - ap;
      sel = 1.f;
      a0 = df == 0 ? 1.f : 0.f;
      a15 = (df <= 1 || df >= g.A - 1) ? 1.f : 0.f;
      a30 = (df <= 2 || df >= g.A - 2) ? 1.f : 0.f;
    }
  }
  float dm = lane < g.D ? t.lm : -INFINITY;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) dm = fmaxf(dm, __shfl_xor(dm, off));
  const float s_score = wave_sum(lane < g.D ? h_score * dm : 0.f);
  const float s_ang = wave_sum(ce_ang * t.lm), s_wid = wave_sum(h_wid * t.lm), s_tol = wave_sum(h_tol * t.lm);
  const float s_lm = wave_sum(t.lm), s_sel = wave_sum(sel), s_a0 = wave_sum(a0), s_a15 = wave_sum(a15), s_a30 = wave_sum(a30);
  float *ax = aux + s * LS_AUX;
  if (lane < g.D) {
    ax[2 + lane] = t.lm;
    ax[2 + LS_MAXD + lane] = (float)t.best;
  }
  if (lane == 0) {
    float *p = partial + s * LS_NP;
    p[0] = ce;
    p[1] = correct ? 1.f : 0.f;
    p[2] = pred == 1 ? 1.f : 0.f;
    p[3] = (correct && pred == 1) ? 1.f : 0.f;
    p[4] = glabel == 1 ? 1.f : 0.f;
    p[5] = (correct && glabel == 1) ? 1.f : 0.f;
    p[6] = sq * mw;
    p[7] = mw;
    p[8] = vmask ? pos : 0.f;
    p[9] = s_score;
    p[10] = dm;
    p[11] = s_ang;
    p[12] = s_lm;
    p[13] = s_wid;
    p[14] = s_tol;
    p[15] = s_sel;
    p[16] = s_a0;
    p[17] = s_a15;
    p[18] = s_a30;
    p[19] = 0.f;
    ax[0] = mw;
    ax[1] = dm;
    graspable_out[s] = glabel;
  }
}

// out[0..13]: total, objectness, view, score, angle, width, tolerance losses; graspable acc / prec / recall; positive
// view count; angle accuracy at 0 / 15 / 30 degrees.  den[0..2]: the three masked-mean denominators (for backward).
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float *__restrict__ partial, long long S, int V, int D,
                                                           float *__restrict__ out, float *__restrict__ den) {
  __shared__ double sh[LS_NP][256 / 64];
  double acc[LS_NP];
#pragma unroll
  for (int k = 0; k < LS_NP; ++k) acc[k] = 0.0;
  for (long long s = threadIdx.x; s < S; s += 256)
#pragma unroll
    for (int k = 0; k < LS_NP; ++k) acc[k] += (double)partial[s * LS_NP + k];
#pragma unroll
  for (int k = 0; k < LS_NP; ++k) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc[k] += __shfl_xor(acc[k], off);
    if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[LS_NP];
    for (int k = 0; k < LS_NP; ++k) t[k] = sh[k][0] + sh[k][1] + sh[k][2] + sh[k][3];
    const float view_den = (float)(t[7] * V) + 1e-6f;
    const float depth_den = (float)(t[10] * D) + 1e-6f;
    const float lm_den = (float)t[12] + 1e-6f;
    const float l_obj = (float)(t[0] / (double)S);
    const float l_view = (float)t[6] / view_den;
    const float l_score = (float)t[9] / depth_den;
    const float l_ang = (float)t[11] / lm_den, l_wid = (float)t[13] / lm_den, l_tol = (float)t[14] / lm_den;
    out[1] = l_obj;
    out[2] = l_view;
    out[3] = l_score;
    out[4] = l_ang;
    out[5] = l_wid;
    out[6] = l_tol;
    out[0] = l_obj + l_view + 0.2f * (((l_score + l_ang) + l_wid) + l_tol);
    out[7] = (float)(t[1] / (double)S);
    out[8] = (float)t[3] / (float)t[2];   // 0/0 = NaN for an empty selection, like the mean of an empty tensor
    out[9] = (float)t[5] / (float)t[4];
    out[10] = (float)t[8];
    out[11] = (float)t[16] / (float)t[15];
    out[12] = (float)t[17] / (float)t[15];
    out[13] = (float)t[18] / (float)t[15];
    den[0] = view_den;
    den[1] = depth_den;
    den[2] = lm_den;
  }
}

struct LossGrads {
  float *d_obj, *d_view, *d_score, *d_angle, *d_width, *d_tol;
};

// go[0..6]: upstream gradients of out[0..6]
__global__ __launch_bounds__(256) void loss_bwd_kernel(LossArgs g, const float *__restrict__ aux,
                                                        const int64_t *__restrict__ graspable,
                                                        const float *__restrict__ den, const float *__restrict__ go,
                                                        LossGrads o) {
  const int lane = threadIdx.x & 63;
  const long long S = (long long)g.B * g.Ns;
  const long long s = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= S) return;
  const int b = (int)(s / g.Ns), i = (int)(s % g.Ns);
  const float c_obj = go[0] + go[1], c_view = go[0] + go[2];
  const float c_score = 0.2f * go[0] + go[3], c_ang = 0.2f * go[0] + go[4], c_wid = 0.2f * go[0] + go[5],
              c_tol = 0.2f * go[0] + go[6];
  const float *ax = aux + s * LS_AUX;
  const float mw = ax[0], dm = ax[1];
  // views: d/dx of sum((x - y)^2 mw) / den
  const float kv = c_view * 2.f * mw / den[0];
  const float *vs = g.view_score + s * g.V, *vl = g.view_label + s * g.V;
  float *dv = o.d_view + s * g.V;
  for (int v = lane; v < g.V; v += 64) dv[v] = kv * (vs[v] - vl[v]);
  // objectness: (softmax - onehot) / S
  if (lane == 0) {
    const float z0 = g.obj_score[b * g.bs_obj + i], z1 = g.obj_score[b * g.bs_obj + g.Ns + i];
    const float zm = fmaxf(z0, z1);
    const float e0 = expf(z0 - zm), e1 = expf(z1 - zm);
    const float inv = 1.f / (e0 + e1), k = c_obj / (float)S;
    const long long gl = graspable[s];
    o.d_obj[((size_t)b * 2) * g.Ns + i] = k * (e0 * inv - (gl == 1 ? 0.f : 1.f));  // dense (B,2,Ns)
    o.d_obj[((size_t)b * 2 + 1) * g.Ns + i] = k * (e1 * inv - (gl == 1 ? 1.f : 0.f));
  }
  // grasp terms: lane = a*D + d
  const int AD = g.A * g.D;
  for (int e = lane; e < AD; e += 64) {
    const int a = e / g.D, d = e % g.D;
    const float lm = ax[2 + d];
    const int best = (int)ax[2 + LS_MAXD + d];
    const size_t po = (size_t)i * g.D + d, pa = (size_t)g.Ns * g.D;
    const size_t at = ((size_t)b * g.A + a) * pa + po;  // the gradients are dense (B,A,Ns,D)
    // angle: lm/den (softmax - onehot)
    const float *ang = g.angle_pred + b * g.bs_angle + po;
    float am = ang[0];
    for (int q = 1; q < g.A; ++q) am = fmaxf(am, ang[q * pa]);
    float se = 0.f;
    for (int q = 0; q < g.A; ++q) se += expf(ang[q * pa] - am);
    const float p = expf(ang[a * pa] - am) / se;
    o.d_angle[at] = c_ang * (lm / den[2]) * (p - (a == best ? 1.f : 0.f));
    float gs = 0.f, gw = 0.f, gt = 0.f;
    if (a == best) {
      const long long sl = (s * g.A + best) * g.D + d;
      const float es = g.score_pred[b * g.bs_score + a * pa + po] - g.labels[sl];
      gs = c_score * (dm / den[1]) * fminf(fmaxf(es, -1.f), 1.f);
      const float ew = (g.width_pred[b * g.bs_width + a * pa + po] - g.offsets[sl * 3 + 2]) / g.max_width;
      gw = c_wid * (lm / den[2]) * fminf(fmaxf(ew, -1.f), 1.f) / g.max_width;
      const float et = (g.tol_pred[b * g.bs_tol + a * pa + po] - g.tolerance[sl]) / g.max_tol;
      gt = c_tol * (lm / den[2]) * fminf(fmaxf(et, -1.f), 1.f) / g.max_tol;
    }
    o.d_score[at] = gs;
    o.d_width[at] = gw;
    o.d_tol[at] = gt;
  }
}

}  // namespace gb

using namespace gb;

static bool loss_args_ok(const LossArgs &g, bool forward) {
  if (forward && (g.view_arg ? (!g.offsets_all || !g.edges || !g.prior_w || g.nb < 1) : !g.weight)) return false;
  return g.B >= 1 && g.Ns >= 1 && g.V >= 1 && g.A >= 1 && g.D >= 1 && g.D <= LS_MAXD && g.obj_score && g.view_score &&
         g.view_label && g.obj_label && g.labels && g.offsets && g.tolerance && g.score_pred && g.angle_pred &&
         g.width_pred && g.tol_pred;
}

extern "C" int gb_grasp_loss_fwd(const float *obj_score, const float *view_score, const float *view_label,
                                 const int64_t *obj_label, const float *weight, const float *labels,
                                 const float *offsets, const float *tolerance, const float *score_pred,
                                 const float *angle_pred, const float *width_pred, const float *tol_pred,
                                 const long long *batch_strides, const int32_t *view_arg, const float *offsets_all,
                                 const float *edges, const float *prior_w, int nb, int B, int Ns, int V, int A, int D, float thresh_bad, float thresh_good, float max_width, float max_tol,
                                 float *partial, float *aux, int64_t *graspable, float *out, float *den, void *stream) {
  if (!batch_strides) return GB_EINVAL;
  const long long *bsv = batch_strides;  // HOST array [obj_score, score, angle, width, tol]
  const LossArgs g = {obj_score, view_score, view_label, obj_label, weight, labels, offsets, tolerance, score_pred,
                      angle_pred, width_pred, tol_pred, view_arg, offsets_all, edges, prior_w, nb, B, Ns, V, A, D, bsv[0], bsv[1], bsv[2], bsv[3], bsv[4], thresh_bad, thresh_good,
                      max_width, max_tol};
  if (!loss_args_ok(g, true) || !partial || !aux || !graspable || !out || !den) return GB_EINVAL;
  const long long S = (long long)B * Ns;
  hipLaunchKernelGGL(loss_seed_kernel, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, as_stream(stream), g, partial, aux,
                     graspable);
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, as_stream(stream), partial, S, V, D, out, den);
  return check_launch("gb_grasp_loss_fwd");
}

extern "C" int gb_grasp_loss_bwd(const float *obj_score, const float *view_score, const float *view_label,
                                 const int64_t *obj_label, const float *weight, const float *labels,
                                 const float *offsets, const float *tolerance, const float *score_pred,
                                 const float *angle_pred, const float *width_pred, const float *tol_pred,
                                 const long long *batch_strides, const int32_t *view_arg, const float *offsets_all,
                                 const float *edges, const float *prior_w, int nb, int B, int Ns, int V, int A, int D, float thresh_bad, float thresh_good, float max_width, float max_tol,
                                 const float *aux, const int64_t *graspable, const float *den, const float *grad_out,
                                 float *d_obj, float *d_view, float *d_score, float *d_angle, float *d_width,
                                 float *d_tol, void *stream) {
  if (!batch_strides) return GB_EINVAL;
  const long long *bsv = batch_strides;  // HOST array [obj_score, score, angle, width, tol]
  const LossArgs g = {obj_score, view_score, view_label, obj_label, weight, labels, offsets, tolerance, score_pred,
                      angle_pred, width_pred, tol_pred, view_arg, offsets_all, edges, prior_w, nb, B, Ns, V, A, D, bsv[0], bsv[1], bsv[2], bsv[3], bsv[4], thresh_bad, thresh_good,
                      max_width, max_tol};
  if (!loss_args_ok(g, false) || !aux || !graspable || !den || !grad_out || !d_obj || !d_view || !d_score || !d_angle ||
      !d_width || !d_tol)
    return GB_EINVAL;
  const long long S = (long long)B * Ns;
  const LossGrads o = {d_obj, d_view, d_score, d_angle, d_width, d_tol};
  hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, as_stream(stream), g, aux, graspable,
                     den, grad_out, o);
  return check_launch("gb_grasp_loss_bwd");
}

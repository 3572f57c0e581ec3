// Furthest point sampling for gfx950 — replaces furthest_point_sampling_kernel of both reference
// extensions (PointNet/_ext_src/src/sampling_gpu.cu:75-234, pointnet2_batch/src/sampling_gpu.cu:74-220).
//
// Design (one workgroup per cloud, the whole cloud REGISTER-resident):
//   * thread t owns points k = t + p*BLOCK (p < P); x,y,z and the running min-distance live in
//     4*P VGPRs, so an iteration touches no memory except one scalar load of the new sample's xyz;
//   * per iteration: P x (3 sub, 3 mul, 2 add, min, cmp, 2 select) VALU, then a wave64 argmax on
//     DPP (float max, then min tie-key among the lanes holding the max), one LDS slot per wave,
//     ONE barrier (slots are double-buffered), a 16-lane DPP reduce of the slots by every wave;
//   * ties are resolved through an explicit key so the result is the reference's for each of the
//     three rules in graspbal.h (lowest index; strided scan + shared-memory tree with block 512/1024):
//     tree winner = smallest bit-reversed (k mod BS), then smallest k div BS.
//     BLOCK is always a multiple of BS, so one thread's points share k mod BS and the strict '>' of
//     the in-thread scan already picks the smallest k div BS.
//   * arithmetic is the no-FMA order ((dx*dx)+(dy*dy))+(dz*dz) (file built with -ffp-contract=off).
// Clouds that do not fit the register file of one CU (n > 24576) go through fps_stream_kernel,
// which keeps the min-distances in the caller's `temp` buffer (L2-resident) instead.
#include <type_traits>

#include "gb_common.h"

namespace gb {

#define GB_FPS_KEY_SHIFT 22

struct FpsTie {
  int bs_log2;  // log2(reference block size), <0 => lowest-index rule
};

__device__ __forceinline__ unsigned fps_key(int k, int bs_log2) {
  if (bs_log2 < 0) return (unsigned)k;
  const unsigned r = (unsigned)k & ((1u << bs_log2) - 1u);
  const unsigned q = (unsigned)k >> bs_log2;
  const unsigned rev = bs_log2 == 0 ? 0u : (__brev(r) >> (32 - bs_log2));
  return (rev << GB_FPS_KEY_SHIFT) | q;
}
__device__ __forceinline__ int fps_unkey(unsigned key, int bs_log2) {
  if (bs_log2 < 0) return (int)key;
  const unsigned rev = key >> GB_FPS_KEY_SHIFT;
  const unsigned q = key & ((1u << GB_FPS_KEY_SHIFT) - 1u);
  const unsigned r = bs_log2 == 0 ? 0u : (__brev(rev) >> (32 - bs_log2));
  return (int)((q << bs_log2) | r);
}

// Block-wide argmax of (d, key): returns the winning point index (wave-uniform in every wave).
// d < 0 marks "no candidate"; if nobody has one the reference's tree returns index 0.
template <int BLOCK>
__device__ __forceinline__ int block_argmax(float d, unsigned key, int bs_log2, float *s_d,
                                            unsigned *s_key, int buf) {
  constexpr int W = BLOCK / 64;
  float wmax = wave_max_f32(d);
  unsigned wkey = wave_min_u32(d == wmax ? key : 0xFFFFFFFFu);
  if constexpr (W > 1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
      s_d[buf * 16 + wave] = wmax;
      s_key[buf * 16 + wave] = wkey;
    }
    __syncthreads();
    float dd = -2.0f;
    unsigned kk = 0xFFFFFFFFu;
    if (lane < W) {
      dd = s_d[buf * 16 + lane];
      kk = s_key[buf * 16 + lane];
    }
    // W <= 16 slots sit in row 0 of the wave; lanes >= W carry (-2, MAX) and never win
    const float rmax = row_max_f32(dd);
    const unsigned rkey = row_min_u32(dd == rmax ? kk : 0xFFFFFFFFu);
    wmax = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(rmax)));
    wkey = (unsigned)__builtin_amdgcn_readfirstlane((int)rkey);
  }
  return wmax < 0.0f ? 0 : fps_unkey(wkey, bs_log2);
}

template <int BLOCK, int P>
__global__ __launch_bounds__(BLOCK) void fps_reg_kernel(const float *__restrict__ xyz,
                                                         float *__restrict__ temp_io,
                                                         int32_t *__restrict__ idx, int n, int m,
                                                         int skip, int bs_log2, const int32_t *__restrict__ guard,
                                                         const float *__restrict__ guard_temp) {
  __shared__ float s_d[32];
  __shared__ unsigned s_key[32];
  const int tid = threadIdx.x;
  const float *pts = xyz + (size_t)blockIdx.x * n * 3;
  int32_t *out = idx + (size_t)blockIdx.x * m;
  float *tio = temp_io ? temp_io + (size_t)blockIdx.x * n : nullptr;
  if (guard && guard[blockIdx.x]) {
    // gb_fps_guarded: the prefix check proved that the samples are 0..m-1 (already written); only the running
    // min-distances remain to be handed over
    if (tio)
      for (int k = tid; k < n; k += BLOCK) tio[k] = guard_temp[(size_t)blockIdx.x * n + k];
    return;
  }

  float px[P], py[P], pz[P], pt[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int k = tid + p * BLOCK;
    float x = 0.f, y = 0.f, z = 0.f, t = -1.0f;  // t = -1: never a candidate (d2 = min(d,-1) = -1)
    if (k < n) {
      const f3 v = reinterpret_cast<const f3 *>(pts)[k];
      x = v.x; y = v.y; z = v.z;
      t = tio ? tio[k] : 1e10f;
      if (skip) {
        const float mag = ((x * x) + (y * y)) + (z * z);
        if (mag < 1e-3f) t = -1.0f;  // == (double)mag <= 1e-3 of sampling_gpu.cu:106 (1e-3f rounds up)
      }
    }
    px[p] = x; py[p] = y; pz[p] = z; pt[p] = t;
  }
  // my tie key for p = 0; p only adds to the low (k div BS) field / to k itself
  const unsigned key0 = fps_key(tid, bs_log2);
  const unsigned keystep = bs_log2 < 0 ? (unsigned)BLOCK : ((unsigned)BLOCK >> bs_log2);

  int old = 0;
  if (tid == 0) out[0] = 0;
  for (int j = 1; j < m; ++j) {
    // wave-uniform address -> scalar loads
    const float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
    // NACC independent (best, argbest) chains over p mod NACC: the compare/select recurrence of a
    // single chain is 3 dependent VALU ops per point, which 4 waves per SIMD cannot fully hide
    constexpr int NACC = P >= 4 ? 4 : 1;
    float bq[NACC];
    int bpq[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) { bq[q] = -1.0f; bpq[q] = q; }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float dx = px[p] - x1, dy = py[p] - y1, dz = pz[p] - z1;
      const float d = ((dx * dx) + (dy * dy)) + (dz * dz);
      const float d2 = __builtin_fminf(d, pt[p]);
      pt[p] = d2;
      const bool gt = d2 > bq[p % NACC];
      bq[p % NACC] = gt ? d2 : bq[p % NACC];
      bpq[p % NACC] = gt ? p : bpq[p % NACC];
    }
    float best = bq[0];
    int bestp = bpq[0];
#pragma unroll
    for (int q = 1; q < NACC; ++q) {  // the strict '>' of a single scan == larger value, then lower p
      const bool take = bq[q] > best || (bq[q] == best && bpq[q] < bestp);
      best = take ? bq[q] : best;
      bestp = take ? bpq[q] : bestp;
    }
    if (best < 0.0f) bestp = 0;
    const unsigned key = key0 + (unsigned)bestp * keystep;
    old = block_argmax<BLOCK>(best, key, bs_log2, s_d, s_key, j & 1);
    if (tid == 0) out[j] = old;
  }
  if (tio) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int k = tid + p * BLOCK;
      if (k < n && pt[p] >= 0.0f) tio[k] = pt[p];  // skipped points keep their input value
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Pruned variant for large clouds.  The update temp[k] = min(temp[k], d(k, new sample)) changes nothing for a
// point that is farther from the new sample than its current temp.  With the points visited in a spatially
// coherent order (`perm`: Morton order, gb_fps_morton_keys + a sort), the cloud is cut into "rows" of 64
// CONSECUTIVE sorted points; row r lives in register slot r / W of wave r % W (one point per lane), so the
// handful of adjacent rows a new sample touches fall into DIFFERENT waves and are processed in parallel.  Lane p
// of a wave keeps the record of the wave's p-th row: bounding box, largest temp, tie key of the point holding
// it.  Per iteration a wave forms, in lanes 0..P-1, the squared distance from the new sample to each of its
// rows' boxes (same un-fused operations as a point distance: IEEE rounding is monotonic, so box distance <=
// every point's computed distance) and updates only the rows where that bound is below the row's largest temp -
// after a few dozen samples ~10 of the 320 rows.  An updated row re-derives its maximum and tie key with two
// DPP wave reductions, and the block arg-max only looks at the row records.  The result is the SAME sample
// sequence as the full update, bit for bit, for every tie rule (keys follow the ORIGINAL indices, LDS table).
template <int BLOCK, int P>
__global__ __launch_bounds__(BLOCK) void fps_pruned_kernel(const float *__restrict__ xyz,
                                                            const int32_t *__restrict__ perm,
                                                            float *__restrict__ temp_io,
                                                            int32_t *__restrict__ idx, int n, int m, int skip,
                                                            int bs_log2) {
  static_assert(P <= 32, "row records live in lanes 0..P-1");
  constexpr int W = BLOCK / 64;
  extern __shared__ unsigned s_tie[];  // [BLOCK * P] tie key of each sorted position
  __shared__ float s_d[32];
  __shared__ unsigned s_key[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *pts = xyz + (size_t)blockIdx.x * n * 3;
  const int32_t *pm = perm + (size_t)blockIdx.x * n;
  int32_t *out = idx + (size_t)blockIdx.x * m;
  float *tio = temp_io ? temp_io + (size_t)blockIdx.x * n : nullptr;

  float px[P], py[P], pz[P], pt[P];
  // row records, valid in lane p < P; other lanes: empty box, never a candidate
  float lox = INFINITY, loy = INFINITY, loz = INFINITY, hix = -INFINITY, hiy = -INFINITY, hiz = -INFINITY;
  float rmax = -2.0f;
  unsigned rkey = 0xFFFFFFFFu;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int k = (p * W + wave) * 64 + lane;
    float x = 0.f, y = 0.f, z = 0.f, t = -INFINITY;  // -inf: never a candidate
    unsigned key = 0xFFFFFFFFu;
    if (k < n) {
      const int o = pm[k];
      const f3 v = reinterpret_cast<const f3 *>(pts)[o];
      x = v.x; y = v.y; z = v.z;
      t = tio ? tio[o] : 1e10f;
      key = fps_key(o, bs_log2);
      if (skip) {
        const float mag = ((x * x) + (y * y)) + (z * z);
        if (mag < 1e-3f) t = -INFINITY;
      }
    }
    px[p] = x; py[p] = y; pz[p] = z; pt[p] = t;
    s_tie[k] = key;
    const bool cand = t >= 0.f;
    const float bhx = wave_max_f32(cand ? x : -INFINITY), blx = -wave_max_f32(cand ? -x : -INFINITY);
    const float bhy = wave_max_f32(cand ? y : -INFINITY), bly = -wave_max_f32(cand ? -y : -INFINITY);
    const float bhz = wave_max_f32(cand ? z : -INFINITY), blz = -wave_max_f32(cand ? -z : -INFINITY);
    const bool any = __builtin_amdgcn_ballot_w64(cand) != 0ull;
    if (lane == p) {
      lox = blx; loy = bly; loz = blz; hix = bhx; hiy = bhy; hiz = bhz;
      rmax = any ? 3.0e38f : -1.0f;  // > any squared distance: forces the row's first update
      rkey = fps_key(0, bs_log2);
    }
  }
  __syncthreads();

  int old = 0;
  if (tid == 0) out[0] = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
    const float ex = fmaxf(fmaxf(lox - x1, x1 - hix), 0.f);
    const float ey = fmaxf(fmaxf(loy - y1, y1 - hiy), 0.f);
    const float ez = fmaxf(fmaxf(loz - z1, z1 - hiz), 0.f);
    const float lb = ((ex * ex) + (ey * ey)) + (ez * ez);
    const unsigned long long need = __builtin_amdgcn_ballot_w64(lb < rmax);  // bit p: my p-th row must be updated
    if (need != 0ull) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (need & (1ull << p)) {  // wave-uniform
          const unsigned kk = s_tie[(p * W + wave) * 64 + lane];
          const float dx = px[p] - x1, dy = py[p] - y1, dz = pz[p] - z1;
          const float d = ((dx * dx) + (dy * dy)) + (dz * dz);
          const float d2 = __builtin_fminf(d, pt[p]);
          pt[p] = d2;
          const float mx = wave_max_f32(d2);
          // the row's arg-max key: almost always a single lane holds the maximum -> read its key directly;
          // exact ties (duplicate points, lattice data) take the min-key reduction
          const unsigned long long eq = __builtin_amdgcn_ballot_w64(d2 == mx);
          unsigned kmin;
          if (__builtin_popcountll(eq) == 1)
            kmin = (unsigned)__builtin_amdgcn_readlane((int)kk, __builtin_ctzll(eq));
          else
            kmin = wave_min_u32(d2 == mx ? kk : 0xFFFFFFFFu);
          if (lane == p) { rmax = mx; rkey = kmin; }
        }
      }
    }
    old = block_argmax<BLOCK>(rmax, rkey, bs_log2, s_d, s_key, j & 1);
    if (tid == 0) out[j] = old;
  }
  if (tio) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int k = (p * W + wave) * 64 + lane;
      if (k < n && pt[p] >= 0.0f) tio[pm[k]] = pt[p];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Round 5: the pruned sampling with O(1) row dispatch and the winner's coordinates carried through the reduction.
// Where the time of fps_pruned_kernel<1024, 20> went (ISA + instruction counts): a wave issues one instruction per
// 4 cycles and the waves of one SIMD share its vector pipeline, so an iteration costs about
// (waves per SIMD) x (instructions a wave executes) x 2-4 cycles whether or not the wave had a row to update - with
// 16 waves every SIMD runs the box test, the 20-branch "which of my rows" chain (3 scalar instructions per row), the
// 12-step wave arg-max and the cross-wave step FOUR times (4 x ~180 instructions), then waits for the dependent
// scalar load of the winner's xyz (L2, ~290 cycles).  Here:
//   * the rows of a wave live in REGISTER VECTORS indexed at run time (s_set_gpr_idx_on: gfx9's VGPR index mode, what
//     the compiler emits for a wave-uniform subscript of an ext_vector_type value), so the needed rows are walked by
//     find-first-set with ONE copy of the update code and no per-row branch at all;
//   * W = 8 waves (two per SIMD) by default: the per-wave fixed work runs twice per SIMD instead of four times;
//   * a row record keeps the COORDINATES of its arg-max point (three readlanes when the row is updated), the wave
//     arg-max publishes (d, x, y, z, key) of its candidate in LDS, and after the one barrier every wave reads the W
//     candidates: the next sample's coordinates come out of that read - no global load in the loop;
//   * the reductions take the single-holder fast path (ballot + readlane) at both levels; exact ties fall back to
//     the min-key reduction, so the sequence is the same bit for bit (tests: test_fps_pruned_is_the_same_sequence
//     runs every layout).
template <int LO, int HI, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (LO < HI) {
    f(std::integral_constant<int, LO>{});
    static_for<LO + 1, HI>(f);
  }
}
// wave max as ONE asm block: the compiler cannot see into an asm statement and pads every separate one with its own
// hazard nops; written as a whole the six steps need only the two wait states each DPP read is owed
__device__ __forceinline__ float wave_max_f32_1(float v) {
  asm volatile(
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float row_max_f32_1(float v) {  // lanes 0..15 (one DPP row), result in every lane of the row
  asm volatile(
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}
// six independent wave maxima interleaved: no wait states needed between the DPP steps of different chains
__device__ __forceinline__ void wave_max_f32_x6(float &a, float &b, float &c, float &d, float &e, float &f) {
#define GB_S6(CTRL)                                          \
  "v_max_f32_dpp %0, %0, %0 " CTRL "\n\tv_max_f32_dpp %1, %1, %1 " CTRL "\n\tv_max_f32_dpp %2, %2, %2 " CTRL \
  "\n\tv_max_f32_dpp %3, %3, %3 " CTRL "\n\tv_max_f32_dpp %4, %4, %4 " CTRL "\n\tv_max_f32_dpp %5, %5, %5 " CTRL "\n\t"
  asm volatile("s_nop 1\n\t" GB_S6("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                   GB_S6("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                       GB_S6("row_half_mirror row_mask:0xf bank_mask:0xf") GB_S6("row_mirror row_mask:0xf bank_mask:0xf")
                           GB_S6("row_bcast:15 row_mask:0xa bank_mask:0xf")
                               GB_S6("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
#undef GB_S6
  a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
  b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
  c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 63));
  d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 63));
  e = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), 63));
  f = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f), 63));
}
// v_writelane_b32: clang has no builtin for it; the LLVM intrinsic is reachable through its name
extern "C" __device__ int gb_writelane_i32(int value, int lane, int vec) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ float rdlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ float wrlane_f(float vec, float s, int l) {
  return __int_as_float(gb_writelane_i32(__float_as_int(s), l, __float_as_int(vec)));
}

// The rows of a wave (one point per lane and row) live in register vectors of 32 or 16 rows ("chunks", at most three:
// P = C0 + C1 + C2; shorter vectors make the compiler fall back to a chain of selects): x, y, z and the running
// min-distance t of row q are element q of four vectors, so a wave-uniform RUN-TIME row number subscripts them.
// (Plain local variables, not members of an aggregate: the optimiser keeps whole-vector locals in registers; inside a
// struct it fell back to scratch memory.)
typedef float fps_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float4 fps_lds_f4;
// (a wave's rows beyond its register vectors - BIG only, rows 48..63 of clouds of more than 49 152 points - keep their
// min-distances in LDS: [row][thread])
struct FpsLdsRows {
  float *base;
  int stride;
};
template <class V> __device__ __forceinline__ float fps_row_get(const V &v, int q) { return v[q]; }
template <class V> __device__ __forceinline__ void fps_row_set(V &v, int q, float x) { v[q] = x; }
__device__ __forceinline__ float fps_row_get(const FpsLdsRows &v, int q) { return v.base[q * v.stride]; }
__device__ __forceinline__ void fps_row_set(FpsLdsRows &v, int q, float x) { v.base[q * v.stride] = x; }
template <int N> struct RowVecT { typedef float T __attribute__((ext_vector_type(N))); };
template <> struct RowVecT<0> { typedef float T __attribute__((ext_vector_type(4))); };  // unused chunk

// BIG (20 480 < n <= 65 536: the cloud no longer fits one CU's registers): only the running min-distances stay in the
// register vectors; a row's coordinates and tie keys are re-read from a sorted (x, y, z, key) copy in global memory
// (`sorted`, written by the prologue) - one 16-byte coalesced load per lane, requested one needed row AHEAD of the row
// being updated - and the tie keys need no LDS table.
template <int W, int C0, int C1, int C2, bool BIG = false, int LR = 0>
__global__ __launch_bounds__(W * 64) void fps_rows_kernel(const float *__restrict__ xyz,
                                                           const int32_t *__restrict__ perm,
                                                           float *__restrict__ temp_io, int32_t *__restrict__ idx,
                                                           int n, int m, int skip, int bs_log2,
                                                           float4 *__restrict__ sorted
#ifdef GB_FPS_STAMPS
                                                           , unsigned long long *__restrict__ stamps
#endif
) {
#ifdef GB_FPS_STAMPS  // diagnostic build only (tools/fps_stamps.sh): cycles per phase of every wave, to a buffer of its own
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_rows = 0, st_t;
#define GB_STAMP0() st_t = __builtin_amdgcn_s_memtime()
#define GB_STAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[i] += now_ - st_t; st_t = now_; }
#else
#define GB_STAMP0()
#define GB_STAMP(i)
#endif
  constexpr int PR = C0 + C1 + C2;     // rows per wave in register vectors
  constexpr int P = PR + LR;           // rows per wave (LR more with their min-distances in LDS: BIG only)
  static_assert(LR == 0 || BIG, "LDS rows: the BIG form only (its LDS is free of the tie-key table)");
  static_assert(W <= 16 && P <= 128, "W candidates sit in one DPP row; row records in two sets of 64 lanes");
  constexpr int S = (P + 63) / 64;     // record sets
  extern __shared__ unsigned s_tie[];  // [W * P * 64] tie key of each sorted position
  __shared__ float4 s_cand[2][16][2];  // {(d, x, y, z), (key, -, -, -)} of each wave's candidate, double-buffered
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float *pts = xyz + (size_t)blockIdx.x * n * 3;
  const int32_t *pm = perm + (size_t)blockIdx.x * n;
  int32_t *out = idx + (size_t)blockIdx.x * m;
  float *tio = temp_io ? temp_io + (size_t)blockIdx.x * n : nullptr;
  const float x0 = pts[0], y0 = pts[1], z0 = pts[2];  // the first sample; also what "nobody is a candidate" returns

  typename RowVecT<C0>::T x0v, y0v, z0v, t0v;
  typename RowVecT<C1>::T x1v, y1v, z1v, t1v;
  typename RowVecT<C2>::T x2v, y2v, z2v, t2v;
  float4 *srt = BIG ? sorted + (size_t)blockIdx.x * n : nullptr;
  FpsLdsRows tl = {reinterpret_cast<float *>(s_tie) + tid, W * 64};   // (LR > 0: the dynamic LDS holds [LR][threads] floats)
  // row records: lane l of set s describes row s*64 + l of this wave (= global row (s*64+l)*W + wave).  A record whose
  // row has no candidate keeps (rmax < 0, key of point 0, coordinates of point 0): when NOBODY has a candidate the
  // reductions below deliver "index 0" - the reference's result - without a special case
  float lox[S], loy[S], loz[S], hix[S], hiy[S], hiz[S], rmax[S], ax[S], ay[S], az[S];
  unsigned rkey[S];
  const unsigned key_of_0 = fps_key(0, bs_log2);
#pragma unroll
  for (int s = 0; s < S; ++s) {  // lanes without a row: empty box, never a candidate
    lox[s] = loy[s] = loz[s] = INFINITY;
    hix[s] = hiy[s] = hiz[s] = -INFINITY;
    rmax[s] = -2.0f;
    rkey[s] = key_of_0;
    ax[s] = x0; ay[s] = y0; az[s] = z0;
  }
  if (tid < 32) {  // the candidate slots of waves that do not exist never win
    s_cand[tid >> 4][tid & 15][0] = make_float4(-3.0f, x0, y0, z0);
    s_cand[tid >> 4][tid & 15][1].x = __uint_as_float(key_of_0);
  }
  // LDS byte addresses of this wave's slot / of the slot this lane reads in the pick, flipped between the two buffers
  // by one XOR per iteration (buffer j & 1; the loop starts at j = 1)
  unsigned wslot = (unsigned)(size_t)(fps_lds_f4 *)&s_cand[1][wave][0];
  unsigned rslot = (unsigned)(size_t)(fps_lds_f4 *)&s_cand[1][lane & 15][0];
  constexpr unsigned SLOT_FLIP = 16 * 2 * sizeof(float4);
  // one row of the prologue: load the row's points, note its box and "has a candidate" in the row's record lane.
  // (`store` receives the row's coordinates, min-distance and key; the register rows are walked by a compile-time
  // loop - their subscripts must be constants -, the LDS rows of the BIG form by an ordinary one)
  auto prologue_row = [&](int p, auto &&store) __attribute__((always_inline)) {
    const int k = (p * W + wave) * 64 + lane;
    float x = 0.f, y = 0.f, z = 0.f, t = -INFINITY;  // -inf: never a candidate
    unsigned key = 0xFFFFFFFFu;
    if (k < n) {
      const int o = pm[k];
      const f3 v = reinterpret_cast<const f3 *>(pts)[o];
      x = v.x; y = v.y; z = v.z;
      t = tio ? tio[o] : 1e10f;
      key = fps_key(o, bs_log2);
      if (skip) {
        const float mag = ((x * x) + (y * y)) + (z * z);
        if (mag < 1e-3f) t = -INFINITY;
      }
    }
    store(x, y, z, t, key, k);
    const bool cand = t >= 0.f;
    float bhx = cand ? x : -INFINITY, bhy = cand ? y : -INFINITY, bhz = cand ? z : -INFINITY;
    float blx = cand ? -x : -INFINITY, bly = cand ? -y : -INFINITY, blz = cand ? -z : -INFINITY;
    wave_max_f32_x6(bhx, bhy, bhz, blx, bly, blz);
    const bool any = __builtin_amdgcn_ballot_w64(cand) != 0ull;
#pragma unroll
    for (int s = 0; s < S; ++s)
      if (lane == p - s * 64) {   // (p is a constant for the register rows: one set survives)
        lox[s] = -blx; loy[s] = -bly; loz[s] = -blz; hix[s] = bhx; hiy[s] = bhy; hiz[s] = bhz;
        rmax[s] = any ? 3.0e38f : -1.0f;  // > any squared distance: forces the row's first update
      }
    __builtin_amdgcn_sched_barrier(0);  // keep the unrolled rows from piling their loads up (register pressure)
  };
  static_for<0, PR>([&](auto pc) __attribute__((always_inline)) {
    constexpr int p = decltype(pc)::value;
    prologue_row(p, [&](float x, float y, float z, float t, unsigned key, int k) __attribute__((always_inline)) {
      if constexpr (BIG) {
        if constexpr (p < C0) t0v[p] = t;
        else if constexpr (p < C0 + C1) t1v[p - C0] = t;
        else t2v[p - C0 - C1] = t;
        if (k < n) srt[k] = make_float4(x, y, z, __uint_as_float(key));
      } else {
        if constexpr (p < C0) { x0v[p] = x; y0v[p] = y; z0v[p] = z; t0v[p] = t; }
        else if constexpr (p < C0 + C1) { x1v[p - C0] = x; y1v[p - C0] = y; z1v[p - C0] = z; t1v[p - C0] = t; }
        else { x2v[p - C0 - C1] = x; y2v[p - C0 - C1] = y; z2v[p - C0 - C1] = z; t2v[p - C0 - C1] = t; }
        s_tie[k] = key;
      }
    });
  });
  if constexpr (LR > 0) {
    for (int p = PR; p < P; ++p)
      prologue_row(p, [&](float x, float y, float z, float t, unsigned key, int k) __attribute__((always_inline)) {
        fps_row_set(tl, p - PR, t);
        if (k < n) srt[k] = make_float4(x, y, z, __uint_as_float(key));
      });
  }
  __syncthreads();

  float x1 = x0, y1 = y0, z1 = z0;  // wave-uniform
  // wave 0 collects the winners' keys, one lane per iteration, and turns 64 of them into indices at a time
  unsigned won = key_of_0;           // (sample 0 is point 0)
#ifdef GB_FPS_STAMPS
  const unsigned long long st_begin = __builtin_amdgcn_s_memtime(), st_rbegin = __builtin_amdgcn_s_memrealtime();
#endif
  for (int j = 1; j < m; ++j) {
    GB_STAMP0();
    static_for<0, S>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      const float ex = fmaxf(fmaxf(lox[s] - x1, x1 - hix[s]), 0.f);
      const float ey = fmaxf(fmaxf(loy[s] - y1, y1 - hiy[s]), 0.f);
      const float ez = fmaxf(fmaxf(loz[s] - z1, z1 - hiz[s]), 0.f);
      const float lb = ((ex * ex) + (ey * ey)) + (ez * ez);
      unsigned long long need = __builtin_amdgcn_ballot_w64(lb < rmax[s]);  // bit l: row s*64+l must be updated
      GB_STAMP(0);
#ifdef GB_FPS_STAMPS
      st_rows += __builtin_popcountll(need);
#endif
      // BIG: the (x, y, z, key) of the first needed row is requested here, every further one while its predecessor is
      // being updated (the load's L2 latency - 300 to 800 cycles - would otherwise sit in front of every row)
      float4 rowq = make_float4(0.f, 0.f, 0.f, 0.f);
      auto row_load = [&](unsigned long long nd) __attribute__((always_inline)) {
        if constexpr (BIG) {
          if (nd != 0ull) {
            const int kq = ((s * 64 + __builtin_ctzll(nd)) * W + wave) * 64 + lane;
            return kq < n ? srt[kq] : make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));
          }
        }
        return make_float4(0.f, 0.f, 0.f, 0.f);
      };
      rowq = row_load(need);
      while (need != 0ull) {  // wave-uniform
        const int l = __builtin_ctzll(need);
        need &= need - 1ull;
        const int p = s * 64 + l;
        unsigned kk;
        float bx = 0.f, by = 0.f, bz = 0.f;
        if constexpr (BIG) {
          kk = __float_as_uint(rowq.w); bx = rowq.x; by = rowq.y; bz = rowq.z;
          rowq = row_load(need);   // the next needed row's, in flight under this one's update
        } else {
          kk = s_tie[(p * W + wave) * 64 + lane];
        }
        auto update = [&](auto &vx, auto &vy, auto &vz, auto &vt, int q) __attribute__((always_inline)) {
          float qx, qy, qz;
          if constexpr (BIG) { qx = bx; qy = by; qz = bz; }
          else { qx = vx[q]; qy = vy[q]; qz = vz[q]; }
          const float qt = fps_row_get(vt, q);
          const float dx = qx - x1, dy = qy - y1, dz = qz - z1;
          const float d = ((dx * dx) + (dy * dy)) + (dz * dz);
          float d2;  // == fminf(d, t) without the canonicalising v_max the builtin puts in front
          asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d), "v"(qt));
          fps_row_set(vt, q, d2);
          GB_STAMP(6);
          const float mx = wave_max_f32_1(d2);
          // the row's arg-max: almost always a single lane holds the maximum -> its key and coordinates are read
          // directly and the check "was it the only holder" comes AFTER the record is written (its scalar chain -
          // count, compare, branch - then runs beside the vector work instead of in front of it); exact ties
          // (duplicate points, lattice data) redo the record with the holder of the smallest key
          GB_STAMP(7);
          const unsigned long long eq = __builtin_amdgcn_ballot_w64(d2 == mx);
          int L = __builtin_ctzll(eq | (1ull << 63));
          GB_STAMP(8);
          auto record = [&](int LL) __attribute__((always_inline)) {
            const unsigned kwin = (unsigned)__builtin_amdgcn_readlane((int)kk, LL);
            const float cx = rdlane_f(qx, LL), cy = rdlane_f(qy, LL), cz = rdlane_f(qz, LL);
            rmax[s] = wrlane_f(rmax[s], mx, l);
            rkey[s] = (unsigned)gb_writelane_i32((int)kwin, l, (int)rkey[s]);
            ax[s] = wrlane_f(ax[s], cx, l);
            ay[s] = wrlane_f(ay[s], cy, l);
            az[s] = wrlane_f(az[s], cz, l);
          };
          record(L);
          if (__builtin_popcountll(eq) != 1) {
            const unsigned kmin = wave_min_u32(d2 == mx ? kk : 0xFFFFFFFFu);
            record(__builtin_ctzll(__builtin_amdgcn_ballot_w64(d2 == mx && kk == kmin) | (1ull << 63)));
          }
          GB_STAMP(9);
        };
        // (the comment-only asm statements differ per branch on purpose: with two chunks of the same type the optimiser
        // otherwise sinks the identical bodies into one block that addresses the vectors through a pointer phi - and a
        // register vector subscripted through memory lives in scratch)
        if ((C1 == 0 && LR == 0) || p < C0) { update(x0v, y0v, z0v, t0v, p); asm volatile("; rows chunk 0"); }
        else if (C1 != 0 && ((C2 == 0 && LR == 0) || p < C0 + C1)) { update(x1v, y1v, z1v, t1v, p - C0); asm volatile("; rows chunk 1"); }
        else if (C2 != 0 && (LR == 0 || p < PR)) { update(x2v, y2v, z2v, t2v, p - C0 - C1); asm volatile("; rows chunk 2"); }
        else if constexpr (LR != 0) { update(x0v, y0v, z0v, tl, p - PR); asm volatile("; rows in LDS"); }
      }
    });
    GB_STAMP(1);
    // the wave's candidate: arg-max over its row records
    float v = rmax[0];
#pragma unroll
    for (int s = 1; s < S; ++s) v = fmaxf(v, rmax[s]);
    const float wmax = wave_max_f32_1(v);
    GB_STAMP(10);
    // ... published by the lane that HOLDS the winning record (no readlanes: the wave is masked down to the holder and
    // it stores its own registers); almost always there is exactly one holder - exact ties between rows first find the
    // smallest key among them.  Several lanes can only remain when their records are identical (no candidate anywhere:
    // every record is "point 0"), and identical stores to one slot are harmless.
    bool holder[S];
    int holders = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      holder[s] = rmax[s] == wmax;
      holders += __builtin_popcountll(__builtin_amdgcn_ballot_w64(holder[s]));
    }
    if (holders != 1) {
      unsigned km = 0xFFFFFFFFu;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const unsigned ks = wave_min_u32(holder[s] ? rkey[s] : 0xFFFFFFFFu);
        km = ks < km ? ks : km;
      }
#pragma unroll
      for (int s = 0; s < S; ++s) holder[s] = holder[s] && rkey[s] == km;
    }
    GB_STAMP(11);
#pragma unroll
    for (int s = 0; s < S; ++s)
      if (holder[s]) {
        const fps_f32x4 rec = {rmax[s], ax[s], ay[s], az[s]};
        asm volatile("ds_write_b128 %0, %1\n\tds_write_b32 %0, %2 offset:16" ::"v"(wslot), "v"(rec), "v"(rkey[s]) : "memory");
      }
    GB_STAMP(2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the compiler does not know that the asm above wrote LDS)
    __syncthreads();
    GB_STAMP(3);
    // The pick.  (A form that stays in vector registers - the 16 slots reduced by DPP with the coordinates as payload,
    // a 64-bit compare and five selects per step - was measured: 68 instructions instead of ~30, and because EVERY wave
    // runs the pick at the same moment the SIMDs' issue slots are what it costs: 1.34 -> 1.88 ms.  Scalar round trips
    // are slow but they cost the other waves nothing.)
    fps_f32x4 c;  // the four DPP rows of the wave all hold the 16 slots
    unsigned ck;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b32 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(c), "=&v"(ck) : "v"(rslot) : "memory");
    wslot ^= SLOT_FLIP;
    rslot ^= SLOT_FLIP;
    const float dmax = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(row_max_f32_1(c.x))));
    unsigned long long e2 = __builtin_amdgcn_ballot_w64(c.x == dmax) & 0xFFFFull;
    int Lw = __builtin_ctzll(e2 | (1ull << 63));
    x1 = rdlane_f(c.y, Lw); y1 = rdlane_f(c.z, Lw); z1 = rdlane_f(c.w, Lw);
    if (__builtin_popcountll(e2) != 1) {  // (the single-holder answer first, the check for ties behind it)
      const unsigned kq = row_min_u32(c.x == dmax ? ck : 0xFFFFFFFFu);
      const unsigned kmin = (unsigned)__builtin_amdgcn_readfirstlane((int)kq);
      e2 = (__builtin_amdgcn_ballot_w64(c.x == dmax && ck == kmin) & 0xFFFFull) | (1ull << 63);
      Lw = __builtin_ctzll(e2);
      x1 = rdlane_f(c.y, Lw); y1 = rdlane_f(c.z, Lw); z1 = rdlane_f(c.w, Lw);
    }
    GB_STAMP(12);
    GB_STAMP(13);
    if (wave == 0) {
      if ((j & 63) == 0) {  // the keys of samples j-64 .. j-1 are complete
        out[j - 64 + lane] = fps_unkey(won, bs_log2);
      }
      won = (unsigned)gb_writelane_i32(__builtin_amdgcn_readlane((int)ck, Lw), j & 63, (int)won);
    }
    GB_STAMP(4);
  }
#ifdef GB_FPS_STAMPS
  if (stamps && lane == 0) {
    unsigned long long *o = stamps + ((size_t)blockIdx.x * 16 + wave) * 20;
    for (int i = 0; i < 16; ++i) o[i] = st_acc[i];
    o[16] = st_rows;
    o[17] = __builtin_amdgcn_s_memtime() - st_begin;
    o[18] = __builtin_amdgcn_s_memrealtime() - st_rbegin;
    o[19] = (unsigned long long)m;
  }
#endif
  if (wave == 0) {
    const int base = (m - 1) & ~63;  // samples base .. m-1 are still in `won`
    if (base + lane < m) out[base + lane] = fps_unkey(won, bs_log2);
  }
  if (tio) {
    static_for<0, PR>([&](auto pc) __attribute__((always_inline)) {
      constexpr int p = decltype(pc)::value;
      const int k = (p * W + wave) * 64 + lane;
      float t;
      if constexpr (p < C0) t = t0v[p];
      else if constexpr (p < C0 + C1) t = t1v[p - C0];
      else t = t2v[p - C0 - C1];
      if (k < n && t >= 0.0f) tio[pm[k]] = t;
    });
    if constexpr (LR > 0) {
      for (int p = PR; p < P; ++p) {
        const int k = (p * W + wave) * 64 + lane;
        const float t = fps_row_get(tl, p - PR);
        if (k < n && t >= 0.0f) tio[pm[k]] = t;
      }
    }
  }
}

// The same for clouds beyond one CU's register file (20 480 < n <= 65 536): only the running min-distances stay in
// registers (P <= 64 per lane); a row's coordinates and tie keys are re-read from a sorted (x, y, z, key) copy in
// global memory - one 16-byte coalesced load per lane - but only for the few rows a sample actually touches, so the
// cloud is streamed once at the start instead of once per iteration (fps_stream_kernel: 16 us per iteration at
// n = 50 000).
template <int BLOCK, int P, int PR>  // rows p < PR keep their min-distances in registers, the rest in LDS
__global__ __launch_bounds__(BLOCK) void fps_pruned_big_kernel(const float *__restrict__ xyz,
                                                                const int32_t *__restrict__ perm,
                                                                float4 *__restrict__ sorted, float *__restrict__ temp_io,
                                                                int32_t *__restrict__ idx, int n, int m, int skip,
                                                                int bs_log2) {
  static_assert(P <= 64, "row records live in lanes 0..P-1");
  constexpr int W = BLOCK / 64;
  __shared__ float s_d[32];
  __shared__ unsigned s_key[32];
  __shared__ float s_t[(P > PR ? P - PR : 1) * BLOCK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *pts = xyz + (size_t)blockIdx.x * n * 3;
  const int32_t *pm = perm + (size_t)blockIdx.x * n;
  float4 *srt = sorted + (size_t)blockIdx.x * n;
  int32_t *out = idx + (size_t)blockIdx.x * m;
  float *tio = temp_io ? temp_io + (size_t)blockIdx.x * n : nullptr;

  float pt[PR];
  float lox = INFINITY, loy = INFINITY, loz = INFINITY, hix = -INFINITY, hiy = -INFINITY, hiz = -INFINITY;
  float rmax = -2.0f;
  unsigned rkey = 0xFFFFFFFFu;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int k = (p * W + wave) * 64 + lane;
    float x = 0.f, y = 0.f, z = 0.f, t = -INFINITY;
    if (k < n) {
      const int o = pm[k];
      const f3 v = reinterpret_cast<const f3 *>(pts)[o];
      x = v.x; y = v.y; z = v.z;
      t = tio ? tio[o] : 1e10f;
      if (skip) {
        const float mag = ((x * x) + (y * y)) + (z * z);
        if (mag < 1e-3f) t = -INFINITY;
      }
      srt[k] = make_float4(x, y, z, __uint_as_float(fps_key(o, bs_log2)));
    }
    if (p < PR) pt[p] = t; else s_t[(p - PR) * BLOCK + tid] = t;
    const bool cand = t >= 0.f;
    const float bhx = wave_max_f32(cand ? x : -INFINITY), blx = -wave_max_f32(cand ? -x : -INFINITY);
    const float bhy = wave_max_f32(cand ? y : -INFINITY), bly = -wave_max_f32(cand ? -y : -INFINITY);
    const float bhz = wave_max_f32(cand ? z : -INFINITY), blz = -wave_max_f32(cand ? -z : -INFINITY);
    const bool any = __builtin_amdgcn_ballot_w64(cand) != 0ull;
    if (lane == p) {
      lox = blx; loy = bly; loz = blz; hix = bhx; hiy = bhy; hiz = bhz;
      rmax = any ? 3.0e38f : -1.0f;
      rkey = fps_key(0, bs_log2);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the unrolled rows from piling their loads up (register pressure)
  }

  int old = 0;
  if (tid == 0) out[0] = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
    const float ex = fmaxf(fmaxf(lox - x1, x1 - hix), 0.f);
    const float ey = fmaxf(fmaxf(loy - y1, y1 - hiy), 0.f);
    const float ez = fmaxf(fmaxf(loz - z1, z1 - hiz), 0.f);
    const float lb = ((ex * ex) + (ey * ey)) + (ez * ez);
    const unsigned long long need = __builtin_amdgcn_ballot_w64(lb < rmax);
    if (need != 0ull) {
      int k0 = wave * 64 + lane;
      asm volatile("" : "+v"(k0));  // keep the P row addresses from being hoisted out of the loop (2 VGPRs each)
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (need & (1ull << p)) {  // wave-uniform
          const int k = p * W * 64 + k0;
          const float4 q = k < n ? srt[k] : make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));
          const unsigned kk = __float_as_uint(q.w);
          const float dx = q.x - x1, dy = q.y - y1, dz = q.z - z1;
          const float d = ((dx * dx) + (dy * dy)) + (dz * dz);
          const float d2 = __builtin_fminf(d, p < PR ? pt[p] : s_t[(p - PR) * BLOCK + tid]);
          if (p < PR) pt[p] = d2; else s_t[(p - PR) * BLOCK + tid] = d2;
          const float mx = wave_max_f32(d2);
          const unsigned long long eq = __builtin_amdgcn_ballot_w64(d2 == mx);
          unsigned kmin;
          if (__builtin_popcountll(eq) == 1)
            kmin = (unsigned)__builtin_amdgcn_readlane((int)kk, __builtin_ctzll(eq));
          else
            kmin = wave_min_u32(d2 == mx ? kk : 0xFFFFFFFFu);
          if (lane == p) { rmax = mx; rkey = kmin; }
        }
      }
    }
    old = block_argmax<BLOCK>(rmax, rkey, bs_log2, s_d, s_key, j & 1);
    if (tid == 0) out[j] = old;
  }
  if (tio) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int k = (p * W + wave) * 64 + lane;
      const float t = p < PR ? pt[p] : s_t[(p - PR) * BLOCK + tid];
      if (k < n && t >= 0.0f) tio[pm[k]] = t;
    }
  }
}

// 30-bit Morton keys of each cloud's points over the cloud's bounding box (10 bits per axis); one workgroup per
// cloud.  Any permutation gives the same FPS result; this one makes consecutive points spatially close.
__device__ __forceinline__ unsigned spread10(unsigned v) {
  v &= 0x3FFu;
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
__global__ __launch_bounds__(1024) void fps_morton_kernel(const float *__restrict__ xyz, int32_t *__restrict__ keys,
                                                           int n) {
  __shared__ float s_lo[3][16], s_hi[3][16];
  const float *pts = xyz + (size_t)blockIdx.x * n * 3;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = threadIdx.x; k < n; k += 1024)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = pts[k * 3 + a];
      lo[a] = fminf(lo[a], v);
      hi[a] = fmaxf(hi[a], v);
    }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float h = wave_max_f32(hi[a]), l = -wave_max_f32(-lo[a]);
    if ((threadIdx.x & 63) == 0) { s_hi[a][threadIdx.x >> 6] = h; s_lo[a][threadIdx.x >> 6] = l; }
  }
  __syncthreads();
  float scale[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float h = s_hi[a][0], l = s_lo[a][0];
    for (int w = 1; w < 16; ++w) { h = fmaxf(h, s_hi[a][w]); l = fminf(l, s_lo[a][w]); }
    lo[a] = l;
    scale[a] = h > l ? 1023.0f / (h - l) : 0.f;
  }
  for (int k = threadIdx.x; k < n; k += 1024) {
    unsigned q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float f = (pts[k * 3 + a] - lo[a]) * scale[a];
      q[a] = f >= 1023.f ? 1023u : (f > 0.f ? (unsigned)f : 0u);  // NaN -> 0
    }
    keys[(size_t)blockIdx.x * n + k] = (int32_t)(spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2));
  }
}

// A spatially coherent visiting order without a sort: counting sort of the points by the 15-bit Morton code of their
// 32 x 32 x 32 grid cell (one workgroup per cloud: histogram, scan and scatter all in LDS; the order inside a cell is
// whatever the atomics make it).  gb_fps_pruned returns the same samples for ANY permutation, so all that matters
// is that 64 consecutive points are neighbours - a cell of the table-top scenes is ~2 cm, a row of 64 points spans a
// few of them - and this is one 15 us launch where keys + a full 30-bit device sort were 130 us and three launches.
constexpr int CO_BINS = 32768;
__device__ __forceinline__ unsigned spread5(unsigned v) {  // bit i -> bit 3i (i < 5)
  return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6) | ((v & 16u) << 8);
}
__global__ __launch_bounds__(1024) void fps_cell_order_kernel(const float *__restrict__ xyz, int32_t *__restrict__ perm,
                                                               int n) {
  extern __shared__ int s_bins[];  // [CO_BINS]
  __shared__ float s_lo[3][16], s_hi[3][16];
  __shared__ int s_part[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *pts = xyz + (size_t)blockIdx.x * n * 3;
  int32_t *out = perm + (size_t)blockIdx.x * n;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = tid; k < n; k += 1024)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = pts[k * 3 + a];
      lo[a] = fminf(lo[a], v);
      hi[a] = fmaxf(hi[a], v);
    }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float h = wave_max_f32(hi[a]), l = -wave_max_f32(-lo[a]);
    if (lane == 0) { s_hi[a][wave] = h; s_lo[a][wave] = l; }
  }
  for (int i = tid; i < CO_BINS; i += 1024) s_bins[i] = 0;
  __syncthreads();
  float scale[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float h = s_hi[a][0], l = s_lo[a][0];
    for (int w = 1; w < 16; ++w) { h = fmaxf(h, s_hi[a][w]); l = fminf(l, s_lo[a][w]); }
    lo[a] = l;
    scale[a] = h > l ? 31.0f / (h - l) : 0.f;
  }
  auto cell_of = [&](int k) {
    unsigned q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float f = (pts[k * 3 + a] - lo[a]) * scale[a];
      q[a] = f >= 31.f ? 31u : (f > 0.f ? (unsigned)f : 0u);  // NaN -> 0
    }
    return (int)(spread5(q[0]) | (spread5(q[1]) << 1) | (spread5(q[2]) << 2));
  };
  for (int k = tid; k < n; k += 1024) atomicAdd(&s_bins[cell_of(k)], 1);
  __syncthreads();
  // exclusive scan of the bins: 32 consecutive bins per thread, then the 1024 thread totals
  constexpr int PER = CO_BINS / 1024;
  int local[PER], sum = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    local[i] = sum;
    sum += s_bins[tid * PER + i];
  }
  int incl = sum;  // inclusive scan across the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) s_part[wave] = incl;
  __syncthreads();
  int base = incl - sum;
  for (int w = 0; w < wave; ++w) base += s_part[w];
#pragma unroll
  for (int i = 0; i < PER; ++i) s_bins[tid * PER + i] = base + local[i];
  __syncthreads();
  for (int k = tid; k < n; k += 1024) out[atomicAdd(&s_bins[cell_of(k)], 1)] = k;
}

// Round 5: a visiting order whose ROWS are compact.  gb_fps_pruned updates a row of 64 consecutive points whenever the
// new sample comes closer to the row's bounding box than the row's largest min-distance, so what matters is how tight
// the boxes of the rows are - and 64 consecutive points of a space-filling curve over a uniform grid are not a box: on
// the bench clouds 11.6 of 313 rows are touched per sample in 32^3-cell Morton order, 8.0 in Hilbert order, 4.9 with the
// leaves of a kd-tree whose splits fall on multiples of 64 points (CPU simulation of the update rule).  Two levels of
// equal-count splits along the locally widest axis get all but a tenth of that (5.4 rows), and each level is one
// counting sort like the one above:
//   level 1  counting sort of the cloud by its widest-axis coordinate (4096 bins); the sorted sequence is cut into
//            K ~ sqrt(rows) SLABS at multiples of 64 points - no selection step: a slab is a range of positions;
//   level 2  every slab is counting-sorted by ITS widest-axis coordinate (1024 bins per slab), so a row - 64
//            consecutive positions - is a short piece of a thin slab.
// Points that share a bin keep whatever order the atomics give them (bin width = extent / 4096 resp. / 1024); the
// samples gb_fps_pruned returns do not depend on the permutation.
constexpr int SO_BINS1 = 4096, SO_BINS2 = 1024, SO_MAXK = 20, SO_MAXN = 24576, SO_MAXK_WS = 32, SO_MAXN_WS = 65536;
__device__ __forceinline__ unsigned so_fkey(float f) {  // order-preserving float -> uint
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float so_funkey(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}
__device__ __forceinline__ int so_quant(float c, float lo, float scale, int bins) {
  const float f = (c - lo) * scale;
  return f >= (float)(bins - 1) ? bins - 1 : (f > 0.f ? (int)f : 0);  // NaN -> 0
}
// exclusive scan of s_bins[0 .. 1024*PER) in place (PER consecutive bins per thread, 1024 threads)
template <int PER>
__device__ __forceinline__ void so_scan(int *s_bins, int *s_part, int tid, int lane, int wave) {
  int local[PER], sum = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    local[i] = sum;
    sum += s_bins[tid * PER + i];
  }
  int incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) s_part[wave] = incl;
  __syncthreads();
  int base = incl - sum;
  for (int w = 0; w < wave; ++w) base += s_part[w];
#pragma unroll
  for (int i = 0; i < PER; ++i) s_bins[tid * PER + i] = base + local[i];
  __syncthreads();
}
// WS: the level-1 order lives in a caller's workspace (`ws`, n int32 per cloud) instead of the LDS - clouds of more than
// 24 576 points, whose 16-bit level-1 order (128 KB at 65 536 points) would not fit beside the level-2 bins
template <bool WS>
__global__ __launch_bounds__(1024) void fps_slab_order_kernel(const float *__restrict__ xyz, int32_t *__restrict__ perm,
                                                               int32_t *__restrict__ ws, int n, int K) {
  constexpr int MAXK = WS ? SO_MAXK_WS : SO_MAXK;
  extern __shared__ int s_bins[];  // [MAXK * SO_BINS2] (>= SO_BINS1), then (!WS) the level-1 order as uint16 [n]
  __shared__ float s_lo[3][16], s_hi[3][16];
  __shared__ int s_part[16];
  __shared__ unsigned s_slo[MAXK][3], s_shi[MAXK][3];
  __shared__ int s_axis[MAXK];
  __shared__ float s_lo2[MAXK], s_scale2[MAXK];
  unsigned short *s_perm1 = reinterpret_cast<unsigned short *>(s_bins + MAXK * SO_BINS2);
  int32_t *g_perm1 = WS ? ws + (size_t)blockIdx.x * n : nullptr;
  auto perm1_get = [&](int pos) { return WS ? (int)g_perm1[pos] : (int)s_perm1[pos]; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *pts = xyz + (size_t)blockIdx.x * n * 3;
  int32_t *out = perm + (size_t)blockIdx.x * n;
  const int nrow = (n + 63) / 64;
  // ---- level 1: the cloud's box, its widest axis, counting sort along it
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = tid; k < n; k += 1024)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = pts[k * 3 + a];
      lo[a] = fminf(lo[a], v);
      hi[a] = fmaxf(hi[a], v);
    }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float h = wave_max_f32(hi[a]), l = -wave_max_f32(-lo[a]);
    if (lane == 0) { s_hi[a][wave] = h; s_lo[a][wave] = l; }
  }
  for (int i = tid; i < SO_BINS1; i += 1024) s_bins[i] = 0;
  for (int i = tid; i < MAXK * 3; i += 1024) {
    (&s_slo[0][0])[i] = 0xFFFFFFFFu;
    (&s_shi[0][0])[i] = 0u;
  }
  __syncthreads();
  int a1 = 0;
  float lo1 = 0.f, ext1 = -1.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float h = s_hi[a][0], l = s_lo[a][0];
    for (int w = 1; w < 16; ++w) { h = fmaxf(h, s_hi[a][w]); l = fminf(l, s_lo[a][w]); }
    if (h - l > ext1) { ext1 = h - l; a1 = a; lo1 = l; }  // (NaN extents never win; all NaN: axis 0, scale 0)
  }
  const float scale1 = ext1 > 0.f ? (float)(SO_BINS1 - 1) / ext1 : 0.f;
  for (int k = tid; k < n; k += 1024) atomicAdd(&s_bins[so_quant(pts[k * 3 + a1], lo1, scale1, SO_BINS1)], 1);
  __syncthreads();
  so_scan<SO_BINS1 / 1024>(s_bins, s_part, tid, lane, wave);
  for (int k = tid; k < n; k += 1024) {
    const int pos = atomicAdd(&s_bins[so_quant(pts[k * 3 + a1], lo1, scale1, SO_BINS1)], 1);
    if constexpr (WS) g_perm1[pos] = k;
    else s_perm1[pos] = (unsigned short)k;
  }
  if constexpr (WS) __threadfence_block();   // (other waves of this workgroup read the order below: through the L2)
  __syncthreads();
  // ---- the slabs: slab i = rows [i*nrow/K, (i+1)*nrow/K) of the level-1 order; their boxes (a row lies in ONE slab)
  auto slab_of_row = [&](int r) {
    int i = (int)(((long long)r * K) / nrow);
    while (i + 1 < K && (int)(((long long)(i + 1) * nrow) / K) <= r) ++i;
    while (i > 0 && (int)(((long long)i * nrow) / K) > r) --i;
    return i;
  };
  for (int r = wave; r < nrow; r += 16) {
    const int pos = r * 64 + lane;
    const bool live = pos < n;
    float x = 0.f, y = 0.f, z = 0.f;
    if (live) {
      const int k = WS ? __builtin_nontemporal_load(g_perm1 + pos) : (int)s_perm1[pos];
      x = pts[k * 3 + 0]; y = pts[k * 3 + 1]; z = pts[k * 3 + 2];
    }
    float bhx = live ? x : -INFINITY, bhy = live ? y : -INFINITY, bhz = live ? z : -INFINITY;
    float blx = live ? -x : -INFINITY, bly = live ? -y : -INFINITY, blz = live ? -z : -INFINITY;
    wave_max_f32_x6(bhx, bhy, bhz, blx, bly, blz);
    if (lane == 0) {
      const int sl = slab_of_row(r);
      atomicMax(&s_shi[sl][0], so_fkey(bhx)); atomicMax(&s_shi[sl][1], so_fkey(bhy)); atomicMax(&s_shi[sl][2], so_fkey(bhz));
      atomicMin(&s_slo[sl][0], so_fkey(-blx)); atomicMin(&s_slo[sl][1], so_fkey(-bly)); atomicMin(&s_slo[sl][2], so_fkey(-blz));
    }
  }
  for (int i = tid; i < K * SO_BINS2; i += 1024) s_bins[i] = 0;  // (the LDS order lies behind the bins: untouched)
  __syncthreads();
  if (tid < K) {
    int a2 = 0;
    float l2 = 0.f, e2 = -1.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float h = so_funkey(s_shi[tid][a]), l = so_funkey(s_slo[tid][a]);
      if (h - l > e2) { e2 = h - l; a2 = a; l2 = l; }
    }
    s_axis[tid] = a2;
    s_lo2[tid] = l2;
    s_scale2[tid] = e2 > 0.f ? (float)(SO_BINS2 - 1) / e2 : 0.f;
  }
  __syncthreads();
  // ---- level 2: counting sort of every slab along its own widest axis (the bin of a position is formed twice -
  // histogram and scatter - from L2-resident data rather than kept in up to 64 registers per thread)
  auto bin_of = [&](int pos, int &k) {
    const int sl = slab_of_row(pos >> 6);
    k = WS ? __builtin_nontemporal_load(g_perm1 + pos) : (int)s_perm1[pos];
    return sl * SO_BINS2 + so_quant(pts[k * 3 + s_axis[sl]], s_lo2[sl], s_scale2[sl], SO_BINS2);
  };
  constexpr int KEEP = WS ? 1 : SO_MAXN / 1024;   // (!WS: a position's bin is kept in a register between the two passes)
  int mybin[KEEP];
  if constexpr (!WS) {
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
      const int pos = tid + i * 1024;
      mybin[i] = -1;
      if (pos < n) {
        int k;
        mybin[i] = bin_of(pos, k);
        atomicAdd(&s_bins[mybin[i]], 1);
      }
    }
  } else {
    for (int pos = tid; pos < n; pos += 1024) {
      int k;
      atomicAdd(&s_bins[bin_of(pos, k)], 1);
    }
  }
  __syncthreads();
  so_scan<MAXK * SO_BINS2 / 1024>(s_bins, s_part, tid, lane, wave);  // (bins beyond K*SO_BINS2: stale, never read)
  if constexpr (!WS) {
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
      const int pos = tid + i * 1024;
      if (pos < n) out[atomicAdd(&s_bins[mybin[i]], 1)] = (int)s_perm1[pos];
    }
  } else {
    for (int pos = tid; pos < n; pos += 1024) {
      int k;
      const int b = bin_of(pos, k);
      out[atomicAdd(&s_bins[b], 1)] = k;
    }
  }
  (void)perm1_get;
}

// Fallback for clouds larger than one CU's register file: min-distances stay in `temp` (global).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void fps_stream_kernel(const float *__restrict__ xyz,
                                                            float *__restrict__ temp,
                                                            int32_t *__restrict__ idx, int n, int m,
                                                            int skip, int bs_log2) {
  __shared__ float s_d[32];
  __shared__ unsigned s_key[32];
  const int tid = threadIdx.x;
  const float *pts = xyz + (size_t)blockIdx.x * n * 3;
  int32_t *out = idx + (size_t)blockIdx.x * m;
  float *t = temp + (size_t)blockIdx.x * n;
  int old = 0;
  if (tid == 0) out[0] = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
    float best = -1.0f;
    int bestk = 0;
    for (int k = tid; k < n; k += BLOCK) {
      const f3 v = reinterpret_cast<const f3 *>(pts)[k];
      if (skip) {
        const float mag = ((v.x * v.x) + (v.y * v.y)) + (v.z * v.z);
        if (mag < 1e-3f) continue;
      }
      const float dx = v.x - x1, dy = v.y - y1, dz = v.z - z1;
      const float d = ((dx * dx) + (dy * dy)) + (dz * dz);
      const float tk = t[k];
      const float d2 = __builtin_fminf(d, tk);
      t[k] = d2;
      if (d2 > best) { best = d2; bestk = k; }
    }
    old = block_argmax<BLOCK>(best, fps_key(bestk, bs_log2), bs_log2, s_d, s_key, j & 1);
    if (tid == 0) out[j] = old;
  }
}

// Segmented FPS: one workgroup per segment of a packed point list (the per-object sampling of the reference's
// ObjectBalanceSampling, modules.py:178-221: `furthest_point_sample(points[seg == j], share_j)` for every object of
// every cloud - dozens of one-workgroup launches there, one launch here).  Segment s = points
// [seg_off[s], seg_off[s+1]), its samples go to idx[out_off[s] .. out_off[s+1]) as indices WITHIN the segment; the
// tie rule is the one gb_fps applies to a cloud of that size.  Min-distances live in `temp` (L2-resident).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void fps_segments_kernel(const float *__restrict__ xyz,
                                                              const int32_t *__restrict__ seg_off,
                                                              const int32_t *__restrict__ out_off,
                                                              float *__restrict__ temp, int32_t *__restrict__ idx,
                                                              int skip, int tie_cap) {
  __shared__ float s_d[32];
  __shared__ unsigned s_key[32];
  const int tid = threadIdx.x, sgm = blockIdx.x;
  const int p0 = seg_off[sgm], n = seg_off[sgm + 1] - p0;
  const int o0 = out_off[sgm], m = out_off[sgm + 1] - o0;
  if (n <= 0 || m <= 0) return;  // block-uniform
  int bs_log2 = -1;
  if (tie_cap >= 0) {
    const int fl = 31 - __clz(n);
    bs_log2 = fl < tie_cap ? fl : tie_cap;
  }
  const float *pts = xyz + (size_t)p0 * 3;
  float *t = temp + p0;
  int32_t *out = idx + o0;
  for (int k = tid; k < n; k += BLOCK) t[k] = 1e10f;
  int old = 0;
  if (tid == 0) out[0] = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
    float best = -1.0f;
    unsigned bestkey = 0xFFFFFFFFu;
    for (int k = tid; k < n; k += BLOCK) {
      const f3 v = reinterpret_cast<const f3 *>(pts)[k];
      if (skip) {
        const float mag = ((v.x * v.x) + (v.y * v.y)) + (v.z * v.z);
        if (mag < 1e-3f) continue;
      }
      const float dx = v.x - x1, dy = v.y - y1, dz = v.z - z1;
      const float d = ((dx * dx) + (dy * dy)) + (dz * dz);
      const float d2 = __builtin_fminf(d, t[k]);
      t[k] = d2;
      const unsigned key = fps_key(k, bs_log2);  // BLOCK need not be a multiple of the reference block size here
      if (d2 > best || (d2 == best && key < bestkey)) { best = d2; bestkey = key; }
    }
    old = block_argmax<BLOCK>(best, bestkey, bs_log2, s_d, s_key, j & 1);
    if (tid == 0) out[j] = old;
  }
}

static int floor_log2(int v) {
  int l = 0;
  while ((2 << l) <= v) ++l;
  return l;
}

template <int BLOCK, int P>
static void launch_reg(const float *xyz, float *temp, int32_t *idx, int b, int n, int m, int skip,
                       int bs_log2, hipStream_t s, const int32_t *guard, const float *guard_temp) {
  hipLaunchKernelGGL((fps_reg_kernel<BLOCK, P>), dim3(b), dim3(BLOCK), 0, s, xyz, temp, idx, n, m,
                     skip, bs_log2, guard, guard_temp);
}

template <int BLOCK>
static bool dispatch_p(int p_need, const float *xyz, float *temp, int32_t *idx, int b, int n, int m,
                       int skip, int bs_log2, hipStream_t s, const int32_t *guard = nullptr,
                       const float *guard_temp = nullptr) {
#define GB_CASE(PV)                                                            \
  if (p_need <= PV) {                                                          \
    launch_reg<BLOCK, PV>(xyz, temp, idx, b, n, m, skip, bs_log2, s, guard, guard_temp); \
    return true;                                                               \
  }
  GB_CASE(1) GB_CASE(2) GB_CASE(4) GB_CASE(8)
  if constexpr (BLOCK == 1024) { GB_CASE(12) GB_CASE(16) GB_CASE(20) GB_CASE(24) }
  else { GB_CASE(16) }
#undef GB_CASE
  return false;
}

}  // namespace gb

static int fps_impl(const float *xyz, float *temp, int32_t *idx, int b, int n, int m, unsigned flags, void *stream,
                    const int32_t *guard, const float *guard_temp) {
  using namespace gb;
  if (b < 0 || n < 1 || m < 0 || !xyz || !idx) return GB_EINVAL;
  if ((long long)n * 3 > 0x7fffffffLL) return GB_ERANGE;
  if (b == 0 || m == 0) return GB_OK;
  const unsigned tie = flags & GB_FPS_TIE_MASK;
  if (tie != GB_FPS_TIE_LOWEST && tie != GB_FPS_TIE_TREE512 && tie != GB_FPS_TIE_TREE1024)
    return GB_EINVAL;
  const int skip = (flags & GB_FPS_SKIP_NEAR_ORIGIN) ? 1 : 0;
  int bs_log2 = -1;
  if (tie != GB_FPS_TIE_LOWEST) {
    const int cap = tie == GB_FPS_TIE_TREE512 ? 9 : 10;
    bs_log2 = floor_log2(n) < cap ? floor_log2(n) : cap;
    if ((n >> bs_log2) >= (1 << GB_FPS_KEY_SHIFT)) return GB_ERANGE;
  }
  hipStream_t s = as_stream(stream);
  // block size: small clouds are latency-bound on the reduction -> fewer waves; it must be a
  // multiple of the reference block size whose tie-break is reproduced.
  int block = n <= 4096 ? 256 : 1024;
  if (bs_log2 >= 0 && (1 << bs_log2) > block) block = 1 << bs_log2;
  bool done = false;
  const int p_need = ceil_div(n, block);
  if (block == 256) done = dispatch_p<256>(p_need, xyz, temp, idx, b, n, m, skip, bs_log2, s, guard, guard_temp);
  else if (block == 512) done = dispatch_p<512>(p_need, xyz, temp, idx, b, n, m, skip, bs_log2, s, guard, guard_temp);
  else done = dispatch_p<1024>(p_need, xyz, temp, idx, b, n, m, skip, bs_log2, s, guard, guard_temp);
  if (!done) {
    if (guard) return GB_ERANGE;  // guarded form: register-resident clouds only
    if (!temp) return GB_EINVAL;  // the streaming path needs the caller's (b,n) scratch
    hipLaunchKernelGGL((fps_stream_kernel<1024>), dim3(b), dim3(1024), 0, s, xyz, temp, idx, n, m,
                       skip, bs_log2);
  }
  return check_launch("gb_fps");
}

extern "C" int gb_fps(const float *xyz, float *temp, int32_t *idx, int b, int n, int m, unsigned flags, void *stream) {
  return fps_impl(xyz, temp, idx, b, n, m, flags, stream, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------
// Prefix check.  FPS of an FPS: when the input points are themselves in farthest-point order (the centres of set
// abstraction level l are the first 2048 samples of level l-1's sampling, in order), the m samples are 0..m-1 -
// the j-th pick over the whole previous cloud was point j, which is in the subset, so it is also the pick within
// it.  That holds unless an exact tie is broken differently (the tie key depends on the index within the set).
// Instead of 1023 sequential iterations the hypothesis "the result is 0..m-1" is VERIFIED in parallel, exactly:
//   T[j]  = min(temp0[j], min_{i<j} d(j, i))        the value point j must win with          (thread per j)
//   every point k: its running min-distance t_k after samples 0..j-1 must not beat T[j]      (thread per k, no
//   (t_k > T[j], or t_k == T[j] with a smaller tie key) for any j                             cross-thread step)
// ok[cloud] = 1 only if no point objects; the FPS kernel launched next returns at once for such clouds (idx and the
// final min-distances come from here) and runs normally otherwise.  Same outputs as gb_fps in every case.
namespace gb {
// Both kernels stage the m samples (x, y, z, T) in LDS first: their loops then read broadcast LDS words instead of
// issuing a dependent global load per iteration (which would cost as much as the sequential FPS they replace).
__global__ __launch_bounds__(256) void fps_prefix_T_kernel(const float *__restrict__ xyz, const float *__restrict__ temp0,
                                                           int n, int m, int skip, float *__restrict__ T,
                                                           int32_t *__restrict__ idx, int32_t *__restrict__ ok) {
  extern __shared__ float4 smp[];  // [m] = (x, y, z, -)
  const float *pts = xyz + (size_t)blockIdx.y * n * 3;
  for (int i = threadIdx.x; i < m; i += 256) smp[i] = make_float4(pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2], 0.f);
  __syncthreads();
  // eight lanes per sample j, lane e takes i = e, e+8, ... < j; min over the eight at the end
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int j = gid / 8, e = gid % 8;
  if (gid == 0) ok[blockIdx.y] = 1;
  const int jj = j < m ? j : m - 1;
  const float x = smp[jj].x, y = smp[jj].y, z = smp[jj].z;
  float t = 3.0e38f;
  for (int i0 = e; i0 < jj; i0 += 64) {
    float4 q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = smp[i0 + 8 * u < jj ? i0 + 8 * u : jj];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + 8 * u < jj) {
        const float dx = x - q[u].x, dy = y - q[u].y, dz = z - q[u].z;
        t = __builtin_fminf(((dx * dx) + (dy * dy)) + (dz * dz), t);
      }
  }
#pragma unroll
  for (int off = 4; off >= 1; off >>= 1) t = __builtin_fminf(__shfl_xor(t, off, 8), t);
  if (j >= m || e != 0) return;
  idx[(size_t)blockIdx.y * m + j] = j;
  t = __builtin_fminf(temp0 ? temp0[(size_t)blockIdx.y * n + j] : 1e10f, t);
  if (skip && (((x * x) + (y * y)) + (z * z)) < 1e-3f) t = -1.0f;  // point j can never be picked (j >= 1)
  T[(size_t)blockIdx.y * m + j] = t;
}

// Eight lanes per point k, each walking one eighth of the samples twice: first its segment's minimum distance to k
// (so the running minimum at the START of every segment is a prefix over 8 lanes), then the segment again with the
// true running value, checking it against T[j].  Twice the arithmetic, one eighth of the dependent iterations.
constexpr int FPV_S = 8;
__global__ __launch_bounds__(256) void fps_prefix_verify_kernel(const float *__restrict__ xyz,
                                                                const float *__restrict__ temp0, int n, int m,
                                                                int skip, int bs_log2, const float *__restrict__ T,
                                                                float *__restrict__ temp_out, int32_t *__restrict__ ok) {
  extern __shared__ float4 smp[];  // [m] = (x, y, z, T)
  const float *pts = xyz + (size_t)blockIdx.y * n * 3;
  const float *Tc = T + (size_t)blockIdx.y * m;
  for (int i = threadIdx.x; i < m; i += 256) smp[i] = make_float4(pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2], Tc[i]);
  __syncthreads();
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int k = gid / FPV_S, sgm = gid % FPV_S;
  const bool live = k < n;
  const int kk = live ? k : n - 1;
  const float x = pts[kk * 3], y = pts[kk * 3 + 1], z = pts[kk * 3 + 2];
  const float t0 = temp0 ? temp0[(size_t)blockIdx.y * n + kk] : 1e10f;
  const bool sk = skip && (((x * x) + (y * y)) + (z * z)) < 1e-3f;
  const unsigned mykey = fps_key(kk, bs_log2);
  const int len = (m - 1 + FPV_S - 1) / FPV_S;        // iterations j = 1..m-1 split into FPV_S segments
  const int j_lo = 1 + sgm * len, j_hi = (j_lo + len < m) ? j_lo + len : m;
  // pass 1: minimum over my segment's samples (sample j-1 enters at iteration j)
  float mseg = 3.0e38f;
  for (int j0 = j_lo; j0 < j_hi; j0 += 8) {
    float4 q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = smp[(j0 + u < j_hi ? j0 + u : j_hi - 1) - 1];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (j0 + u < j_hi) {
        const float dx = x - q[u].x, dy = y - q[u].y, dz = z - q[u].z;
        mseg = __builtin_fminf(((dx * dx) + (dy * dy)) + (dz * dz), mseg);
      }
  }
  // exclusive prefix-min over the FPV_S lanes of this point (lanes are consecutive: width-8 shuffles)
  float t = t0;
#pragma unroll
  for (int e = 1; e < FPV_S; ++e) {
    const float other = __shfl_up(mseg, e, FPV_S);
    if (sgm >= e) t = __builtin_fminf(other, t);
  }
  // pass 2: the true running value through my segment, against T[j]
  bool bad = false;
  for (int j0 = j_lo; j0 < j_hi; j0 += 8) {
    float4 q[8];
    float tj[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + u < j_hi ? j0 + u : j_hi - 1;
      q[u] = smp[j - 1];
      tj[u] = smp[j].w;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + u;
      if (j < j_hi) {
        const float dx = x - q[u].x, dy = y - q[u].y, dz = z - q[u].z;
        t = __builtin_fminf(((dx * dx) + (dy * dy)) + (dz * dz), t);
        if (kk == j) bad |= sk || !(tj[u] >= 0.f);  // j itself must be a candidate
        else if (!sk) {                              // nobody may beat it
          bad |= t > tj[u];
          if (t == tj[u]) bad |= mykey < fps_key(j, bs_log2);
        }
      }
    }
  }
  bad = bad && live;
  // the last non-empty segment ends with the final running value; empty trailing segments just carry the prefix
  if (live && sgm == FPV_S - 1) temp_out[(size_t)blockIdx.y * n + k] = sk ? t0 : t;
  if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) atomicExch(ok + blockIdx.y, 0);
}
}  // namespace gb

extern "C" int gb_fps_guarded(const float *xyz, float *temp, int32_t *idx, int b, int n, int m, unsigned flags,
                              float *scratch_T, float *scratch_temp, int32_t *ok, void *stream) {
  using namespace gb;
  if (b < 0 || n < 1 || m < 1 || m > n || !xyz || !idx || !scratch_T || !scratch_temp || !ok) return GB_EINVAL;
  if (b > 65535 || n > 24576 || m > 4096) return GB_ERANGE;  // the m samples are staged in 64 KB of LDS
  if (b == 0) return GB_OK;
  const unsigned tie = flags & GB_FPS_TIE_MASK;
  if (tie != GB_FPS_TIE_LOWEST && tie != GB_FPS_TIE_TREE512 && tie != GB_FPS_TIE_TREE1024) return GB_EINVAL;
  const int skip = (flags & GB_FPS_SKIP_NEAR_ORIGIN) ? 1 : 0;
  int bs_log2 = -1;
  if (tie != GB_FPS_TIE_LOWEST) {
    const int cap = tie == GB_FPS_TIE_TREE512 ? 9 : 10;
    bs_log2 = floor_log2(n) < cap ? floor_log2(n) : cap;
  }
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(fps_prefix_T_kernel, dim3((m * 8 + 255) / 256, b), dim3(256), (size_t)m * sizeof(float4), s, xyz, temp, n,
                     m, skip, scratch_T, idx, ok);
  hipLaunchKernelGGL(fps_prefix_verify_kernel, dim3((n * FPV_S + 255) / 256, b), dim3(256), (size_t)m * sizeof(float4), s, xyz,
                     temp, n, m, skip, bs_log2, scratch_T, scratch_temp, ok);
  const int rc = check_launch("gb_fps_guarded");
  if (rc != GB_OK) return rc;
  return fps_impl(xyz, temp, idx, b, n, m, flags, stream, ok, scratch_temp);
}

extern "C" int gb_fps_morton_keys(const float *xyz, int32_t *keys, int b, int n, void *stream) {
  using namespace gb;
  if (b < 0 || n < 1 || !xyz || !keys) return GB_EINVAL;
  if ((long long)n * 3 > 0x7fffffffLL) return GB_ERANGE;
  if (b == 0) return GB_OK;
  hipLaunchKernelGGL(fps_morton_kernel, dim3(b), dim3(1024), 0, as_stream(stream), xyz, keys, n);
  return check_launch("gb_fps_morton_keys");
}

extern "C" int gb_fps_cell_order(const float *xyz, int32_t *perm, int b, int n, void *stream) {
  using namespace gb;
  if (b < 0 || n < 1 || !xyz || !perm) return GB_EINVAL;
  if ((long long)n * 3 > 0x7fffffffLL) return GB_ERANGE;
  if (b == 0) return GB_OK;
  static std::atomic<unsigned long long> attr{0};
  allow_dynamic_lds(fps_cell_order_kernel, CO_BINS * (int)sizeof(int), attr);
  hipLaunchKernelGGL(fps_cell_order_kernel, dim3(b), dim3(1024), CO_BINS * sizeof(int), as_stream(stream), xyz, perm, n);
  return check_launch("gb_fps_cell_order");
}

static int fps_row_order_impl(const float *xyz, int32_t *perm, int32_t *ws, int b, int n, void *stream) {
  using namespace gb;
  if (b < 0 || n < 1 || !xyz || !perm) return GB_EINVAL;
  if ((long long)n * 3 > 0x7fffffffLL) return GB_ERANGE;
  const bool big = n > SO_MAXN;
  if (big && (!ws || n > SO_MAXN_WS)) return gb_fps_cell_order(xyz, perm, b, n, stream);  // no room for the level-1 order
  if (b == 0) return GB_OK;
  const int nrow = (n + 63) / 64;
  int K = 1;
  while ((2 * K + 1) * (2 * K + 1) <= 4 * nrow) ++K;  // K = round(sqrt(nrow))
  const int kmax = big ? SO_MAXK_WS : SO_MAXK;
  if (K > kmax) K = kmax;
  if (big) {
    const int lds = SO_MAXK_WS * SO_BINS2 * (int)sizeof(int);
    static std::atomic<unsigned long long> attr{0};
    allow_dynamic_lds(fps_slab_order_kernel<true>, lds, attr);
    hipLaunchKernelGGL(fps_slab_order_kernel<true>, dim3(b), dim3(1024), lds, as_stream(stream), xyz, perm, ws, n, K);
  } else {
    const int lds = SO_MAXK * SO_BINS2 * (int)sizeof(int) + ((n + 1) / 2) * 4;
    static std::atomic<unsigned long long> attr{0};
    allow_dynamic_lds(fps_slab_order_kernel<false>, SO_MAXK * SO_BINS2 * (int)sizeof(int) + SO_MAXN * 2, attr);
    hipLaunchKernelGGL(fps_slab_order_kernel<false>, dim3(b), dim3(1024), lds, as_stream(stream), xyz, perm, nullptr, n, K);
  }
  return check_launch("gb_fps_row_order");
}

extern "C" int gb_fps_row_order(const float *xyz, int32_t *perm, int b, int n, void *stream) {
  return fps_row_order_impl(xyz, perm, nullptr, b, n, stream);
}

extern "C" int gb_fps_row_order_ws(const float *xyz, int32_t *perm, int32_t *ws, int b, int n, void *stream) {
  return fps_row_order_impl(xyz, perm, ws, b, n, stream);
}

#ifdef GB_FPS_STAMPS
static unsigned long long *g_fps_stamps = nullptr;  // diagnostic build only
extern "C" void gb_debug_fps_stamps(unsigned long long *buf) { g_fps_stamps = buf; }
#define GB_STAMP_ARG , g_fps_stamps
#else
#define GB_STAMP_ARG
#endif
extern "C" int gb_fps_pruned(const float *xyz, const int32_t *perm, float *temp, int32_t *idx, int b, int n, int m,
                             unsigned flags, float *scratch, void *stream) {
  using namespace gb;
  if (b < 0 || n < 1 || m < 0 || !xyz || !perm || !idx) return GB_EINVAL;
  if (n > 1024 * 64 || (n > 1024 * 63 && (flags & GB_FPS_LAYOUT_MASK) == GB_FPS_LAYOUT_R4)) return GB_ERANGE;  // (round 4's kernel: 24 rows in registers + 39 in LDS)
  if (n > 1024 * 20 && (!scratch || reinterpret_cast<uintptr_t>(scratch) % 16 != 0)) return GB_EINVAL;
  if (b == 0 || m == 0) return GB_OK;
  const unsigned tie = flags & GB_FPS_TIE_MASK;
  if (tie != GB_FPS_TIE_LOWEST && tie != GB_FPS_TIE_TREE512 && tie != GB_FPS_TIE_TREE1024) return GB_EINVAL;
  const int skip = (flags & GB_FPS_SKIP_NEAR_ORIGIN) ? 1 : 0;
  int bs_log2 = -1;
  if (tie != GB_FPS_TIE_LOWEST) {
    const int cap = tie == GB_FPS_TIE_TREE512 ? 9 : 10;
    bs_log2 = floor_log2(n) < cap ? floor_log2(n) : cap;
    if ((n >> bs_log2) >= (1 << GB_FPS_KEY_SHIFT)) return GB_ERANGE;
  }
  hipStream_t s = as_stream(stream);
  const unsigned layout = flags & GB_FPS_LAYOUT_MASK;
  if (layout > GB_FPS_LAYOUT_R4) return GB_EINVAL;
  if (n <= 1024 * 20 && layout != GB_FPS_LAYOUT_R4) {
    // round 5: 4 waves (default) or 8 waves per cloud, the winner's coordinates carried through LDS (fps_rows_kernel)
#define GB_ROWS(WV, A0, A1, A2)                                                                               \
  if (ceil_div(n, WV * 64) <= A0 + A1 + A2) {                                                                  \
    static std::atomic<unsigned long long> attr{0};                                                            \
    constexpr int lds = WV * 64 * (A0 + A1 + A2) * 4;                                                          \
    allow_dynamic_lds(fps_rows_kernel<WV, A0, A1, A2>, lds, attr);                                             \
    hipLaunchKernelGGL((fps_rows_kernel<WV, A0, A1, A2>), dim3(b), dim3(WV * 64), lds, s, xyz, perm, temp, idx, n, m, \
                       skip, bs_log2, (float4 *)nullptr GB_STAMP_ARG);                                         \
    return check_launch("gb_fps_pruned");                                                                      \
  }
    // measured on 4 x 20000 -> 2048 (tools/fps_bench.py, row order): 12 waves 1.33 ms, 8 waves 1.46, 16 waves 1.56,
    // 4 waves 2.42 (its rows spill into accumulation registers, which run-time indexing cannot reach), round 4's 1.84
    if (layout == GB_FPS_LAYOUT_W16) {
      GB_ROWS(16, 16, 0, 0) GB_ROWS(16, 16, 4, 0)
    } else if (layout == GB_FPS_LAYOUT_W8) {
      GB_ROWS(8, 16, 0, 0) GB_ROWS(8, 32, 0, 0) GB_ROWS(8, 32, 16, 0)
    } else if (layout == GB_FPS_LAYOUT_W4) {
      GB_ROWS(4, 32, 0, 0) GB_ROWS(4, 32, 32, 0) GB_ROWS(4, 32, 32, 16)
    } else {
      GB_ROWS(12, 16, 0, 0) GB_ROWS(12, 32, 0, 0)
    }
#undef GB_ROWS
  }
  const int p_need = ceil_div(n, 1024);
#define GB_PR(PV)                                                                                              \
  if (p_need <= PV) {                                                                                          \
    static std::atomic<unsigned long long> attr{0};                                                            \
    allow_dynamic_lds(fps_pruned_kernel<1024, PV>, 1024 * PV * 4, attr);                                       \
    hipLaunchKernelGGL((fps_pruned_kernel<1024, PV>), dim3(b), dim3(1024), 1024 * PV * sizeof(unsigned), s, xyz, perm, \
                       temp, idx, n, m, skip, bs_log2);                                                        \
    return check_launch("gb_fps_pruned");                                                                      \
  }
  GB_PR(4) GB_PR(8) GB_PR(12) GB_PR(16) GB_PR(20)
#undef GB_PR
  if (n > 1024 * 20 && layout != GB_FPS_LAYOUT_R4) {
    // round 5: the same kernel with only the min-distances in registers (16 waves: the records of up to 64 rows per wave
    // fill one set of lanes); coordinates and keys come from the sorted copy in `scratch`
#define GB_ROWS_BIG(A0, A1, LRV)                                                                                    \
  if (ceil_div(n, 1024) <= A0 + A1 + LRV) {                                                                      \
    static std::atomic<unsigned long long> attr{0};                                                            \
    allow_dynamic_lds(fps_rows_kernel<16, A0, A1, 0, true, LRV>, LRV * 1024 * 4, attr);                        \
    hipLaunchKernelGGL((fps_rows_kernel<16, A0, A1, 0, true, LRV>), dim3(b), dim3(1024), LRV * 1024 * sizeof(float), s, \
                       xyz, perm, temp, idx, n, m, skip, bs_log2, reinterpret_cast<float4 *>(scratch) GB_STAMP_ARG); \
    return check_launch("gb_fps_pruned");                                                                      \
  }
    GB_ROWS_BIG(32, 0, 0) GB_ROWS_BIG(32, 16, 0) GB_ROWS_BIG(32, 16, 16)
#undef GB_ROWS_BIG
  }
#define GB_PB(PV)                                                                                               \
  if (p_need <= PV) {                                                                                          \
    hipLaunchKernelGGL((fps_pruned_big_kernel<1024, PV, 24>), dim3(b), dim3(1024), 0, s, xyz, perm,                 \
                       reinterpret_cast<float4 *>(scratch), temp, idx, n, m, skip, bs_log2);                    \
    return check_launch("gb_fps_pruned");                                                                      \
  }
  GB_PB(32) GB_PB(48) GB_PB(63)
#undef GB_PB
  return GB_ERANGE;
}

// Segmented FPS (see fps_segments_kernel).  xyz (T,3) packed points; seg_off / out_off (S+1) int32 device arrays
// (non-decreasing); temp (T) float workspace; idx (out_off[S]) int32.  max_n = the largest segment (host-known
// bound, for the tie-key range check only).
extern "C" int gb_fps_segments(const float *xyz, const int32_t *seg_off, const int32_t *out_off, float *temp,
                               int32_t *idx, int S, int max_n, unsigned flags, void *stream) {
  using namespace gb;
  if (S < 0 || max_n < 0 || !xyz || !seg_off || !out_off || !temp || !idx) return GB_EINVAL;
  if (S == 0) return GB_OK;
  const unsigned tie = flags & GB_FPS_TIE_MASK;
  if (tie != GB_FPS_TIE_LOWEST && tie != GB_FPS_TIE_TREE512 && tie != GB_FPS_TIE_TREE1024) return GB_EINVAL;
  const int tie_cap = tie == GB_FPS_TIE_LOWEST ? -1 : (tie == GB_FPS_TIE_TREE512 ? 9 : 10);
  if (tie_cap >= 0 && max_n > 0) {
    const int bl = floor_log2(max_n) < tie_cap ? floor_log2(max_n) : tie_cap;
    if ((max_n >> bl) >= (1 << GB_FPS_KEY_SHIFT)) return GB_ERANGE;
  }
  hipLaunchKernelGGL((fps_segments_kernel<256>), dim3(S), dim3(256), 0, as_stream(stream), xyz, seg_off, out_off, temp,
                     idx, (flags & GB_FPS_SKIP_NEAR_ORIGIN) ? 1 : 0, tie_cap);
  return check_launch("gb_fps_segments");
}

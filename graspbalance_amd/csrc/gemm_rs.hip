// Row-streaming fp32 MFMA GEMM for the tall-skinny products of the channel-last SharedMLP path
// (gfx950, v_mfma_f32_32x32x2_f32).  D (P,C) = f(A (P,R)) B (R,C) with P ~ 10^5..10^6 rows and a
// small weight matrix B (R*C*4 <= ~128 KB):
//
//   * the whole of B is staged ONCE per workgroup in LDS ([r][C32] image, zero padded) and stays
//     there; workgroups are persistent (one or two per CU) and their 8 waves walk 32-row tiles of A;
//   * A never touches LDS.  The MFMA A operand of lane l is A[row l&31][k-slot l>>5], and any
//     permutation of the reduction index is legal as long as B uses the same one, so each lane takes
//     16 CONTIGUOUS reduction indices of its row per chunk (half 0: k0..k0+15, half 1: k0+16..k0+31)
//     straight from global memory as four 16-byte loads, one chunk ahead of the MFMAs that consume it
//     (register double buffer; a chunk is 16*NT MFMAs = 1024*NT cycles of cover for the next loads);
//   * every element of A is read from HBM exactly once and every element of D written once; the
//     previous layer's BatchNorm affine + ReLU is applied to A in registers, the BatchNorm column
//     statistics of D (forward) or the BatchNorm-backward sums (dgrad) accumulate in fp64 registers
//     across all the tiles of a wave and leave as ONE atomic per column per workgroup.
//
// Used by gb_gemm_fwd / gb_gemm_dgrad (csrc/gemm_cl.hip) when the shape fits; the LDS-tiled kernel
// there remains the general path.  Replaces the cuBLAS/cuDNN 1x1 convolutions the reference reaches
// through torch (pytorch_utils.py:61-113).
#include <stdlib.h>

#include <type_traits>

#include "gb_common.h"
#include "gemm_rs.h"

namespace gb {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int RS_TPB = 512, RS_WAVES = RS_TPB / 64;
constexpr int RS_CH = 32;  // reduction indices per chunk: 16 contiguous per lane half

struct RsArgs {
  const float *a;       // (P,R), row pitch lda
  const float *w;       // weights, see w_kc
  float *d;             // (P,C), row pitch ldd
  const float *aff;     // optional [a(R), b(R)]: A is used as relu(a_k x + b_k)
  double *stats;        // fp64 [slots][2C]
  const float *epi_y;   // RS_BNBWD: (P,C) pitch ldd, pre-BatchNorm output of the layer D is the gradient of
  const float *epi_ab;  // RS_BNBWD: [a, b, mean, rstd](C)
  const float *epi_x;   // RS_BNBWD_X: (P,3) the 3-channel input of the layer D is the gradient of
  const uint16_t *epi_w16;  // RS_STATS (optional): per-row multiplicity, padded with zeros to a multiple of 32 rows
  const int32_t *epi_key;   // RS_STATS_POOL: per-row (seed << 13) | (multiplicity << 4) | member bits, zero-padded likewise
  const float *epi_gamma;   // RS_STATS_POOL: BatchNorm weight (C): its sign decides whether a crop's max of relu(a*y+b)
                            // sits at the largest or the smallest y
  float2 *pairs;            // RS_STATS_POOL: [(tile + seed)][D][C] (sign*y, row bits) partial extrema, see pool_epilogue
  int pool_d;               // RS_STATS_POOL: crops per seed (1..4)
  const float *gen_x;       // GEN3 / RS_BNBWD_X: (P,3) rows the A operand / the epilogue's y are generated from
  const float *gen_w;       // ... with the 3-input first layer's weight (R,3) resp. (C,3): y1 = ((x*w0) + (y*w1)) + (z*w2)
  const long long *rows_dev;  // optional (device): the actual row count, <= P (GbGemmOpts.rows_dev); P is then the capacity
  long long P;
  int R, C, lda, ldd;
  int w_kc;             // 1: B[r][c] = w[c*R + r] (forward, W (C,R));  0: B[r][c] = w[r*C + c] (dgrad, W (R,C))
  int slots, nch;       // statistics slot rows ; chunks = ceil(R / 32)
  int stagger;          // delay waves 4-7 by half a tile
  int tail_split;       // cut the tiles of a short last round into column groups
};

// BF (GB_PREC_BF16): B lives in LDS as bf16 in [k / 8][C32][8] order - the 8 reduction indices one lane feeds to a
// v_mfma_f32_32x32x16_bf16 are 16 contiguous bytes - and a lane's 16 fp32 values of A become two bf16x8 operands.
// GEN3: the A operand is not read - it is relu(a_k * y1 + b_k) of a 3-input first layer, y1 = x0[row] . W1[k], formed in
// registers from the row's 12 bytes (gen_x) and a per-k table in LDS: that layer's output is never written or read.
__device__ __forceinline__ float lin3(float x, float y, float z, float w0, float w1, float w2) {
  return ((x * w0) + (y * w1)) + (z * w2);   // the ONE evaluation order every consumer of the folded layer uses
}

// ---- the B operand's LDS reads of the fp32 MFMA loop, software-pipelined by hand -----------------------------------
// Left to the compiler the loop came out as  ds_read2_b32 -> s_waitcnt lgkmcnt(0) -> 2 MFMAs  through ONE register pair:
// every pair of MFMAs waited for a full LDS round trip (PMC: 32-34 % of a wave's cycles in s_waitcnt, MFMA pipe 50-60 %
// busy).  Here the NT values of reduction step J+1 are requested BEFORE the NT MFMAs of step J are issued, as inline
// asm with immediate offsets (the compiler sinks a plain load to its first use), and the wait is a counted
// `s_waitcnt lgkmcnt(NT)` whose operands are tied in/out ("+v"): the values the MFMAs read come out of the wait, so no
// use can move above it, and the registers stay allocated while the read is in flight.  LDS returns in order, so any
// LDS / scalar-memory operation of the compiler's own in between only makes the counted wait stricter.
// tools/wg_check_isa.py (family `gemm_rs`; run by `make check-isa` as part of the library build and by a CPU test) walks the generated code for any instruction that touches
// a register between its asm read and the asm wait that retires it.
template <int NT, int J, int Q = 0>
__device__ __forceinline__ void rs_b_request(float (&b)[NT], unsigned addr) {
  if constexpr (Q < NT) {
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(b[Q]) : "v"(addr), "n"((J * NT * 32 + Q * 32) * 4));
    rs_b_request<NT, J, Q + 1>(b, addr);
  }
}

#define RS_LGKM(N, ...) asm volatile("s_waitcnt lgkmcnt(" #N ")" : __VA_ARGS__)
template <int NT, bool MORE>   // MORE: the next step's NT reads were issued behind the ones waited for
__device__ __forceinline__ void rs_b_wait(float (&b)[NT]) {
  static_assert(NT == 2 || NT == 4 || NT == 5 || NT == 8, "add the operand list");
  if constexpr (NT == 2) {
    if constexpr (MORE) RS_LGKM(2, "+v"(b[0]), "+v"(b[1]));
    else RS_LGKM(0, "+v"(b[0]), "+v"(b[1]));
  } else if constexpr (NT == 4) {
    if constexpr (MORE) RS_LGKM(4, "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
    else RS_LGKM(0, "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
  } else if constexpr (NT == 5) {
    if constexpr (MORE) RS_LGKM(5, "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]));
    else RS_LGKM(0, "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]));
  } else {
    if constexpr (MORE)
      RS_LGKM(8, "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
    else
      RS_LGKM(0, "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
  }
}

// steps J..15 of a chunk: `cur` holds step J's values (in flight), `nxt` receives step J+1's
template <int NT, int J>
__device__ __forceinline__ void rs_b_steps(f32x16 (&acc)[NT], const float (&av)[16], float (&cur)[NT], float (&nxt)[NT],
                                           unsigned addr) {
  if constexpr (J + 1 < 16) rs_b_request<NT, J + 1>(nxt, addr);
  rs_b_wait<NT, (J + 1 < 16)>(cur);
#pragma unroll
  for (int q = 0; q < NT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[J], cur[q], acc[q], 0, 0, 0);
  if constexpr (J + 1 < 16) rs_b_steps<NT, J + 1>(acc, av, nxt, cur, addr);
}

// ---- the A operand's global loads, requested one chunk ahead by hand -------------------------------------------------
// The plain form - four loads, each under its own (row < P && k < R) branch - could not be counted by the compiler's
// s_waitcnt pass: the chunk's first use of `cur` came out as s_waitcnt vmcnt(0) placed BEHIND the requests for `nxt`, so
// every chunk waited for the loads it had just issued (the "one chunk ahead" of the design never overlapped anything).
// Here the four loads are unconditional inline asm from clamped, always valid addresses, and the wait is ONE
// `s_waitcnt vmcnt(0)` behind the chunk's MFMAs - in front of the tile's epilogue, so a tile's stores are not waited
// for - with the registers tied through it like the B operand's.
typedef float rs_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void rs_a_request(rs_f32x4 (&d)[4], const float *p0, const float *p1, const float *p2,
                                             const float *p3) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d[0]) : "v"(p0));
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d[1]) : "v"(p1));
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d[2]) : "v"(p2));
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d[3]) : "v"(p3));
}
__device__ __forceinline__ void rs_a_wait(rs_f32x4 (&d)[4]) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
}
// The same wait for the instantiations in which the allocator satisfied the tied operands above with copies of the
// registers in flight (the BatchNorm-backward dgrads in bf16 and as split): no tie - the 16 values leave the asm
// statement in FRESH registers (early-clobber outputs), moved there behind the wait inside the statement itself.
__device__ __forceinline__ void rs_a_wait_mov(const rs_f32x4 (&d)[4], float (&o)[16]) {
  asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, %8\n\tv_mov_b32 %1, %9\n\tv_mov_b32 %2, %10\n\tv_mov_b32 %3, %11\n\t"
               "v_mov_b32 %4, %12\n\tv_mov_b32 %5, %13\n\tv_mov_b32 %6, %14\n\tv_mov_b32 %7, %15"
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
               : "v"(d[0][0]), "v"(d[0][1]), "v"(d[0][2]), "v"(d[0][3]), "v"(d[1][0]), "v"(d[1][1]), "v"(d[1][2]), "v"(d[1][3]));
  asm volatile("v_mov_b32 %0, %8\n\tv_mov_b32 %1, %9\n\tv_mov_b32 %2, %10\n\tv_mov_b32 %3, %11\n\t"
               "v_mov_b32 %4, %12\n\tv_mov_b32 %5, %13\n\tv_mov_b32 %6, %14\n\tv_mov_b32 %7, %15"
               : "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15])
               : "v"(d[2][0]), "v"(d[2][1]), "v"(d[2][2]), "v"(d[2][3]), "v"(d[3][0]), "v"(d[3][1]), "v"(d[3][2]), "v"(d[3][3]));
}

// SP (GB_PREC_F32_SPLIT3, round 5): fp32 products through the bf16 matrix cores.  a = a_hi + a_mid + a_lo EXACTLY (three
// 8-bit slices of the 24-bit mantissa, by truncation), likewise the weights - three bf16 images [k / 8][C32][8] in LDS -
// and the six products of weight >= 2^-16 (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) on v_mfma_f32_32x32x16_bf16,
// smallest first, fp32 accumulate: the error against fp64 is that of the fp32 MFMA (tools/split3_probe.hip: 1.7e-7
// against 2.0e-7 relative L2) at twice the inner loop's rate.
// CGS: column groups.  Three images of a 128 x 256 matrix are 192 KB, so a workgroup owns C32 = NT*32 of the CGS*C32
// output columns; the CGS groups of a row range sit on workgroups 8 apart (one XCD: the second read of A is an L2 hit).
__device__ __forceinline__ float rs_trunc16(float x) { return __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }
__device__ __forceinline__ unsigned rs_pack_hi(float lo_elem, float hi_elem) {   // two exactly-bf16 floats -> one register
  return __builtin_amdgcn_perm(__float_as_uint(hi_elem), __float_as_uint(lo_elem), 0x07060302u);
}

#ifndef GB_RS_NT_STORE
#define GB_RS_NT_STORE 1
#endif
#if GB_RS_NT_STORE
#define RS_STORE_Y(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define RS_STORE_Y(ptr, val) (*(ptr) = (val))
#endif

template <int NT, int EPI, bool BF = false, bool GEN3 = false, bool SP = false, int CGS = 1>
__global__ __launch_bounds__(RS_TPB, ((NT <= 2 && EPI != RS_BNBWD && EPI != RS_BNBWD_X && !SP) ? 4 : 2)) void gemm_rs_kernel(RsArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  static_assert(!(SP && BF), "the split is an fp32 mode");
  static_assert(CGS == 1 || SP, "column groups exist for the split's three images");
  constexpr int C32 = NT * 32;
  constexpr int LDD = C32 * CGS;          // packed row pitch of the whole product
  const int rpad = g.nch * RS_CH;
  float *Bs = lds;                        // fp32: [rpad][C32] ; bf16: [rpad / 8][C32][8] (half the bytes) ; split: three of those
  float *s_aff = lds + (size_t)rpad * C32 * (SP ? 3 : BF ? 1 : 2) / 2;  // [2][rpad]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int m = lane & 31, h = lane >> 5;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
  // column group: this workgroup's C32 columns of the ctot; every global array is rebased to its first column, what is
  // left to the code below are the strides (weights, tables, statistics, pairs) - ctot instead of g.C
  const int ctot = g.C;
  unsigned bid = blockIdx.x, nwg = gridDim.x;
  if constexpr (CGS > 1) {
    const int cg = (int)((blockIdx.x >> 3) % CGS);
    bid = ((blockIdx.x >> 3) / CGS) * 8 + (blockIdx.x & 7);
    nwg = gridDim.x / CGS;
    const int col0 = cg * C32;
    g.C = C32;                            // (the host launches CGS > 1 only for ctot == CGS * C32)
    g.w += g.w_kc ? (size_t)col0 * g.R : (size_t)col0;
    if (g.d) g.d += col0;
    if (g.epi_y) g.epi_y += col0;
    if (g.stats) g.stats += col0;
    if (g.epi_ab) g.epi_ab += col0;
    if (g.epi_gamma) g.epi_gamma += col0;
    if (g.pairs) g.pairs = reinterpret_cast<float2 *>(reinterpret_cast<float *>(g.pairs) + col0);
  }
  auto put_b = [&](int r, int c, float v) {
    if constexpr (SP) {
      __bf16 *img = reinterpret_cast<__bf16 *>(lds);
      const size_t o = ((size_t)(r >> 3) * C32 + c) * 8 + (r & 7), one = (size_t)rpad * C32;
      const float hi = rs_trunc16(v), r1 = v - hi, mid = rs_trunc16(r1), lo = r1 - mid;   // all exact
      img[o] = (__bf16)hi;
      img[one + o] = (__bf16)mid;
      img[2 * one + o] = (__bf16)lo;      // (<= 8 significant bits)
    } else if constexpr (BF) reinterpret_cast<__bf16 *>(lds)[((size_t)(r >> 3) * C32 + c) * 8 + (r & 7)] = (__bf16)v;
    else Bs[(size_t)r * C32 + c] = v;
  };

  // ---- stage B (zero padded); consecutive threads write consecutive columns: conflict-free.  Four 16-byte loads per
  // thread are issued back to back from clamped (always valid) addresses before anything uses them - with a load
  // per loop trip the 128 KB image took 16-64 dependent round trips (50 us of a 73 us launch at P = 16 384).
  if (g.w_kc) {
    const int total = (rpad / 4) * C32;
    for (int i0 = t; i0 < total; i0 += 4 * RS_TPB) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * RS_TPB;
        const int c = i % C32, r = (i / C32) * 4;
        const bool ok = i < total && c < g.C && r < g.R;  // R % 4 == 0
        v[u] = *reinterpret_cast<const float4 *>(g.w + (ok ? (size_t)c * g.R + r : 0));
        if (!ok) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * RS_TPB;
        if (i < total) {
          const int c = i % C32, r = (i / C32) * 4;
          put_b(r + 0, c, v[u].x);
          put_b(r + 1, c, v[u].y);
          put_b(r + 2, c, v[u].z);
          put_b(r + 3, c, v[u].w);
        }
      }
    }
  } else if (ctot % 4 == 0 && reinterpret_cast<uintptr_t>(g.w) % 16 == 0) {
    constexpr int Q = C32 / 4;  // float4 per LDS row
    const int total = rpad * Q;
    for (int i0 = t; i0 < total; i0 += 4 * RS_TPB) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * RS_TPB;
        const int r = i / Q, c = (i % Q) * 4;
        const bool ok = i < total && r < g.R && c < g.C;
        v[u] = *reinterpret_cast<const float4 *>(g.w + (ok ? (size_t)r * ctot + c : 0));
        if (!ok) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * RS_TPB;
        if (i < total) {
          if constexpr (BF || SP) {
            put_b(i / Q, (i % Q) * 4 + 0, v[u].x);
            put_b(i / Q, (i % Q) * 4 + 1, v[u].y);
            put_b(i / Q, (i % Q) * 4 + 2, v[u].z);
            put_b(i / Q, (i % Q) * 4 + 3, v[u].w);
          } else {
            *reinterpret_cast<float4 *>(Bs + (size_t)(i / Q) * C32 + (i % Q) * 4) = v[u];
          }
        }
      }
    }
  } else {
    for (int i = t; i < rpad * C32; i += RS_TPB) {
      const int c = i % C32, r = i / C32;
      put_b(r, c, (r < g.R && c < g.C) ? g.w[(size_t)r * ctot + c] : 0.f);
    }
  }
  if (g.aff)
    for (int i = t; i < rpad; i += RS_TPB) {
      s_aff[i] = i < g.R ? g.aff[i] : 0.f;
      s_aff[rpad + i] = i < g.R ? g.aff[g.R + i] : 0.f;
    }
  float *s_gen = s_aff + 2 * rpad;  // GEN3: [rpad][4] = (w0, w1, w2, 0) of reduction index k
  if constexpr (GEN3)
    for (int i = t; i < rpad * 4; i += RS_TPB) s_gen[i] = ((i & 3) < 3 && (i >> 2) < g.R) ? g.gen_w[(i >> 2) * 3 + (i & 3)] : 0.f;
  // RS_STATS_POOL_V: the fp64 column sums live in LDS, not in registers - [wave pair][2][C32] doubles behind the tables,
  // updated per tile with ds_add_f64 by the two waves that share them.  With 128 accumulators, 16 row keys and 32
  // registers of fp64 sums the pooled epilogue spilled 130 registers (0.25 GB of scratch written per launch).
  // Round 5: the 256-wide RS_STATS instantiations as well (NT = 8: 128 accumulators + 32 registers of fp64 sums + the
  // chunk double buffer; the bf16 one spilled 78 registers - the stress configuration's dominant launches)
  constexpr bool LSTAT = EPI == RS_STATS_POOL_V || (EPI == RS_STATS && NT == 8) || (EPI == RS_BNBWD && BF) ||
                         (SP && (EPI == RS_STATS || EPI == RS_BNBWD));   // (the split's operands take the registers)
  static_assert(!(LSTAT && GEN3 && !SP), "s_gen and s_st would share the LDS behind the tables");
  double *s_st = reinterpret_cast<double *>(s_aff + (g.aff ? 2 * rpad : 0) + (GEN3 ? 4 * rpad : 0));
  if constexpr (LSTAT)
    for (int i = t; i < (RS_WAVES / 2) * 2 * C32; i += RS_TPB) s_st[i] = 0.0;
  __syncthreads();

  // rows of this launch: the caller's device-side count when there is one (it is not known on the host: no read-back)
  long long gP = g.P;
  if (g.rows_dev) {
    const long long pd = *g.rows_dev;
    gP = pd < g.P ? (pd > 0 ? pd : 0) : g.P;
  }
  const long long ntiles = (gP + 31) / 32;
  const long long nw = (long long)nwg * RS_WAVES;
  // wave w of every workgroup before wave w+1 of any: with fewer tiles than waves the work spreads over all CUs
  // (and over the four SIMDs of each) instead of filling the 8 waves of the first workgroups
  long long tile = (long long)wave * nwg + bid;

  constexpr bool BNB = EPI == RS_BNBWD || EPI == RS_BNBWD_X;
  constexpr int NS = EPI == RS_BNBWD_X ? 5 : 2;  // column sums per column: [g, g*xhat (, g*x0, g*x1, g*x2)] / [y, y^2]
  double dsum[NT], dsq[NT], dtx[EPI == RS_BNBWD_X ? NT : 1][3];
#pragma unroll
  for (int j = 0; j < NT; ++j) { dsum[j] = 0.0; dsq[j] = 0.0; }
#pragma unroll
  for (int j = 0; j < (EPI == RS_BNBWD_X ? NT : 1); ++j) { dtx[j][0] = 0.0; dtx[j][1] = 0.0; dtx[j][2] = 0.0; }
  // BatchNorm coefficients of this lane's columns: resident in registers unless the accumulators need them
  constexpr bool COEF_REGS = NT <= 5;
  float ea[NT], eb[NT], em[NT], er[NT];
  if constexpr (BNB && COEF_REGS) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = j * 32 + m;
      const bool ok = col < g.C;
      ea[j] = ok ? g.epi_ab[col] : 0.f;
      eb[j] = ok ? g.epi_ab[ctot + col] : 0.f;
      em[j] = ok ? g.epi_ab[2 * ctot + col] : 0.f;
      er[j] = ok ? g.epi_ab[3 * ctot + col] : 0.f;
    }
  }

  // RS_BNBWD_X with gen_w: the layer's pre-BatchNorm output is not stored - y = x_in[row] . W1[col] is re-formed from the
  // input rows the epilogue holds anyway (same lin3 as the forward's GEN3 operand)
  float gx[EPI == RS_BNBWD_X ? NT : 1][3];
  if constexpr (EPI == RS_BNBWD_X) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = j * 32 + m;
#pragma unroll
      for (int e = 0; e < 3; ++e) gx[j][e] = (g.gen_w && col < g.C) ? g.gen_w[col * 3 + e] : 0.f;
    }
  }
  // LSTAT: a column tile's two partial sums (this lane: 16 of the column's 32 rows) go to the wave pair's fp64 slots
  auto lds_stat_add = [&](int q, float cs, float cq) {
    const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(cs), __float_as_uint(cs), false, false);
    const auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(cq), __float_as_uint(cq), false, false);
    cs += __uint_as_float(h ? s1[0] : s1[1]);   // the column's other 16 rows sit in lane ^ 32
    cq += __uint_as_float(h ? s2[0] : s2[1]);
    if (h == 0) {
      double *sp = s_st + (size_t)(wave & 3) * 2 * C32 + q * 32 + m;
      __hip_atomic_fetch_add(sp, (double)cs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(sp + C32, (double)cq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  auto load_chunk = [&](float4 (&dst)[4], long long tl, int kc) {
    const long long row = tl * 32 + m;
    if constexpr (GEN3) {  // the row's xyz (an L1 hit after the tile's first chunk); the values are formed at use
      const bool ok = row < gP;
      dst[0] = make_float4(ok ? g.gen_x[row * 3] : 0.f, ok ? g.gen_x[row * 3 + 1] : 0.f, ok ? g.gen_x[row * 3 + 2] : 0.f, 0.f);
      return;
    }
    const int k = kc * RS_CH + h * 16;
    const float *p = g.a + row * g.lda + k;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      dst[i] = (row < gP && k + 4 * i < g.R) ? *reinterpret_cast<const float4 *>(p + 4 * i)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  // (asm form, not GEN3) rows past the end read the last row - every epilogue masks them; reduction indices past R read
  // the row's first floats and are zeroed at use (a_tail: R % 32 != 0 only)
  const bool a_tail = g.R % RS_CH != 0;
  auto request_chunk = [&](rs_f32x4 (&dst)[4], long long tl, int kc) {
    long long row = tl * 32 + m;
    row = row < gP ? row : gP - 1;
    const int k = kc * RS_CH + h * 16;
    const float *p = g.a + row * g.lda;
    rs_a_request(dst, p + (k < g.R ? k : 0), p + (k + 4 < g.R ? k + 4 : 0), p + (k + 8 < g.R ? k + 8 : 0),
                 p + (k + 12 < g.R ? k + 12 : 0));
  };
  // The two waves that share a SIMD (w and w+4) run the same program from the same start and would reach their
  // MFMA phases and their epilogues together; delaying the upper four by half a tile's worth of MFMA time lets one
  // partner's epilogue (stores, statistics) run under the other's MFMAs (+2-4 % on the 1 M-row layers).
  if (g.stagger && wave >= RS_WAVES / 2) {
    // half a tile's MFMA time in s_sleep units of 64 cycles: fp32 nch*16*NT MFMAs of 64 cycles; bf16 nch*2*NT of 32; the
    // split nch*12*NT of 32 (round 6: the split instantiations slept the fp32 figure - 1.3 tile times, a third of a
    // 131 072-row launch's four tiles per wave - and every remainder was rounded up to 100 units)
    int units = SP ? g.nch * NT * 3 : BF ? (g.nch * NT + 1) / 2 : g.nch * NT * 8;
    for (; units >= 100; units -= 100) __builtin_amdgcn_s_sleep(100);
    for (; units >= 10; units -= 10) __builtin_amdgcn_s_sleep(10);
    for (; units > 0; --units) __builtin_amdgcn_s_sleep(1);
  }
  // Tail balance: ntiles is rarely a multiple of the nw waves, and a last round with a few tiles costs a whole tile
  // time (6.06 rounds run as 7).  When the remainder is small, its tiles are cut into G column groups handled by G
  // different waves (each re-reads the 32 rows of A - L2 hits - and owns NT/G column tiles: stores, statistics and
  // fp64 sums are per column, so nothing else changes): the last round then costs 1/G of a tile time.
  long long main_end = ntiles;
  int G = 1;
  {
    const long long rem = ntiles % nw;
    if (g.tail_split && rem > 0) {
      const long long gmax = nw / rem;
      while (G * 2 <= gmax && G * 2 <= NT && NT % (G * 2) == 0) G *= 2;
      if (G >= 2) main_end = ntiles - rem;
      else G = 1;
    }
  }

  auto walk = [&](auto part, long long tile, const long long tend, const int qlo, const int qhi) {
    constexpr bool PART = decltype(part)::value;  // true: only column tiles [qlo, qhi) of the tile
    auto in = [&](int q) { return !PART || (q >= qlo && q < qhi); };
  if (tile < tend) {
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // (not the bf16 instantiations: in two of them the allocator still copied in-flight tuples in front of the wait)
    // (the split's dgrad: the same copies, found by the same check - those two take the untied wait, A_MOV)
    constexpr bool A_MOV = (SP || BF) && EPI == RS_BNBWD && !GEN3;
    constexpr bool A_ASM = !GEN3 && (!BF || A_MOV);
    float4 cur[4], nxt[4];
    // The chunk in flight lives in `afl` only between its request and its wait INSIDE one loop trip; what is carried
    // round the loop are the 16 plain values taken out of it behind the wait.  (Carrying the tuples themselves made the
    // register allocator satisfy the wait's tied operands with copies of the in-flight registers - tools/wg_check_isa.py.)
    rs_f32x4 afl[4];
    float anext[16];
    if constexpr (A_ASM) {
      request_chunk(afl, tile, 0);
      if constexpr (A_MOV) rs_a_wait_mov(afl, anext);
      else {
        rs_a_wait(afl);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) anext[4 * i + e] = afl[i][e];
      }
    } else {
      load_chunk(cur, tile, 0);
    }
    int kc = 0;
    while (true) {
      int nkc = kc + 1;
      long long ntile = tile;
      if (nkc == g.nch) { nkc = 0; ntile = tile + nw; }
      const bool more = ntile < tend;
      // RS_STATS_POOL_V: the next tile's first chunk is requested AFTER this tile's epilogue (its 16 registers are what the
      // epilogue's row keys need; holding both spilled into scratch, and the epilogue then ran 3x the tile's MFMA time)
#ifndef GB_DEFER
#define GB_DEFER 1
#endif
      // (the asm form does not defer: without the per-load branches and with the fp64 sums in LDS the 16 registers fit)
      const bool defer = GB_DEFER && !A_ASM && EPI == RS_STATS_POOL_V && kc == g.nch - 1;
      if constexpr (!A_ASM) {
        if (more && !defer) load_chunk(nxt, ntile, nkc);
      }

      // (fp32, whole tiles) the B values of the chunk's first step are requested here, ahead of the operand's prologue
      float bq0[BF || SP || PART ? 1 : NT], bq1[BF || SP || PART ? 1 : NT];
      const unsigned baddr = lds0 + (unsigned)(((kc * RS_CH + h * 16) * C32 + m) * 4);
      // (not where the prologue is long vector code of the compiler's own - the generated operand, the 3-channel rows'
      // address arithmetic: the checker found it reading undefined high halves out of registers in flight there)
      constexpr bool B_EARLY = !BF && !SP && !PART && !GEN3 && EPI != RS_BNBWD_X;
      if constexpr (B_EARLY) rs_b_request<NT, 0>(bq0, baddr);
      float av[16];
      if constexpr (GEN3) {
        const float4 *gw = reinterpret_cast<const float4 *>(s_gen) + kc * RS_CH + h * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float4 w = gw[i];
          av[i] = lin3(cur[0].x, cur[0].y, cur[0].z, w.x, w.y, w.z);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (A_ASM) av[4 * i + e] = anext[4 * i + e];
            else av[4 * i + e] = e == 0 ? cur[i].x : e == 1 ? cur[i].y : e == 2 ? cur[i].z : cur[i].w;
          }
        if (A_ASM && a_tail) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) av[4 * i + e] = (kc * RS_CH + h * 16 + 4 * i < g.R) ? av[4 * i + e] : 0.f;
        }
      }
      if (g.aff) {  // previous layer's BatchNorm + ReLU; channels >= R have a = b = 0 and stay zero
        const float *ca = s_aff + kc * RS_CH + h * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float z = ca[i] * av[i] + ca[rpad + i];
          av[i] = z > 0.f ? z : 0.f;
        }
      }
      // RS_BNBWD: the y values the epilogue needs are requested BEFORE the tile's last MFMA block, so their
      // latency hides behind it (registers permitting; otherwise per column tile inside the epilogue)
      constexpr bool YPRE = BNB && NT <= 4;
      float yv[YPRE ? NT : 1][16];
      float xr[EPI == RS_BNBWD_X ? 16 : 1][3];  // RS_BNBWD_X: the 3-channel input rows of this tile, same early request
      if constexpr (EPI == RS_BNBWD_X) {
        if (kc == g.nch - 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const long long row = tile * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
#pragma unroll
            for (int j = 0; j < 3; ++j) xr[r][j] = row < gP ? g.epi_x[row * 3 + j] : 0.f;
          }
        }
      }
      uint2 wq[EPI == RS_STATS ? 4 : 1];  // RS_STATS with row weights: 16 uint16 multiplicities of this lane's rows
      if constexpr (EPI == RS_STATS) {
        if (g.epi_w16 && kc == g.nch - 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            wq[i] = *reinterpret_cast<const uint2 *>(g.epi_w16 + tile * 32 + 4 * h + 8 * i);
        }
      }
      if constexpr (YPRE) {
        if (kc == g.nch - 1 && !(EPI == RS_BNBWD_X && g.gen_w)) {
#pragma unroll
          for (int q = 0; q < NT; ++q)
            if (in(q)) {
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const long long row = tile * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
                const int col = q * 32 + m;
                yv[q][r] = (col < g.C && row < gP) ? g.epi_y[row * g.ldd + col] : 0.f;
              }
            }
        }
      }
      // the next chunk's request: behind the prologue and the epilogue's early requests - the compiler guards the
      // registers it reuses there with s_waitcnt vmcnt(N) of its own book-keeping (it cannot see these loads), and one of
      // those behind the request would wait for it at once.  Unconditional: past the end it re-reads the last row.
      if constexpr (A_ASM) request_chunk(afl, ntile, nkc);
      if constexpr (SP) {
        // the lane's 16 values as three exact 8-bit slices each (vector ALU: and, subtract, and, subtract, three packs
        // per pair of values - paid in full beside the MFMAs, tools/mfma_valu_overlap.hip), then per k-group of 8 the
        // six products, term-major over the NT accumulators: a chain's next link is NT MFMAs later (with the six
        // products of one accumulator back to back the probe ran 2.5x slower)
        unsigned ph[8], pm[8], pl[8];
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const float x0 = av[i], x1 = av[i + 1];
          const float h0 = rs_trunc16(x0), h1 = rs_trunc16(x1), r0 = x0 - h0, r1 = x1 - h1;
          const float m0 = rs_trunc16(r0), m1 = rs_trunc16(r1), l0 = r0 - m0, l1 = r1 - m1;
          ph[i / 2] = rs_pack_hi(h0, h1);
          pm[i / 2] = rs_pack_hi(m0, m1);
          pl[i / 2] = rs_pack_hi(l0, l1);
        }
        const bf16x8 *img = reinterpret_cast<const bf16x8 *>(lds);
        const size_t one = (size_t)(rpad / 8) * C32;   // bf16x8 elements per image
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          bf16x8 ah, am, al;
          __builtin_memcpy(&ah, &ph[4 * u], 16);
          __builtin_memcpy(&am, &pm[4 * u], 16);
          __builtin_memcpy(&al, &pl[4 * u], 16);
          const bf16x8 *bp8 = img + (size_t)(kc * 4 + 2 * h + u) * C32 + m;
          bf16x8 bh[NT], bm[NT], bl[NT];
#pragma unroll
          for (int q = 0; q < NT; ++q)
            if (in(q)) { bh[q] = bp8[q * 32]; bm[q] = bp8[one + q * 32]; bl[q] = bp8[2 * one + q * 32]; }
#pragma unroll
          for (int q = 0; q < NT; ++q)
            if (in(q)) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[q], acc[q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < NT; ++q)
            if (in(q)) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[q], acc[q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < NT; ++q)
            if (in(q)) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm[q], acc[q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < NT; ++q)
            if (in(q)) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh[q], acc[q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < NT; ++q)
            if (in(q)) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm[q], acc[q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < NT; ++q)
            if (in(q)) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[q], acc[q], 0, 0, 0);
        }
      } else if constexpr (BF) {
        // this lane's 16 reduction indices kc*32 + h*16 + (0..15) = the two k-groups kc*4 + 2h + {0, 1}
        bf16x8 a8[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) a8[u][e] = (__bf16)av[8 * u + e];
        const bf16x8 *bp8 = reinterpret_cast<const bf16x8 *>(lds) + (size_t)(kc * 4 + 2 * h) * C32 + m;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          bf16x8 b8[NT];
#pragma unroll
          for (int q = 0; q < NT; ++q)
            if (in(q)) b8[q] = bp8[u * C32 + q * 32];
#pragma unroll
          for (int q = 0; q < NT; ++q)
            if (in(q)) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[u], b8[q], acc[q], 0, 0, 0);
        }
      } else if constexpr (!PART) {
        if constexpr (!B_EARLY) rs_b_request<NT, 0>(bq0, baddr);
        rs_b_steps<NT, 0>(acc, av, bq0, bq1, baddr);
      } else {
      const float *bp = Bs + (size_t)(kc * RS_CH + h * 16) * C32 + m;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float bv[NT];
#pragma unroll
        for (int q = 0; q < NT; ++q)
          if (in(q)) bv[q] = bp[j * C32 + q * 32];
#pragma unroll
        for (int q = 0; q < NT; ++q)
          if (in(q)) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[q], acc[q], 0, 0, 0);
      }
      }

      // the next chunk's values: waited for behind this chunk's MFMAs and IN FRONT of the tile's epilogue (its stores
      // are then not waited for by anything until the next chunk's wait, a chunk of MFMAs later)
      if constexpr (A_ASM) {
        // (the MFMAs are not volatile: without the fence the scheduler may sink them below this wait - in one build of
        // the split 46 of a chunk's 48 went there and the request had nothing to run under)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (A_MOV) rs_a_wait_mov(afl, anext);
        else {
          rs_a_wait(afl);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) anext[4 * i + e] = afl[i][e];
        }
      }
      if (kc == g.nch - 1) {
        // acc[q][r] = D[tile*32 + (r&3) + 8*(r>>2) + 4*h][q*32 + m]
        const long long row0 = tile * 32 + 4 * h;
        // Full tiles (all 32 rows and all NT*32 columns valid - every tile but the last of the usual shapes) skip the
        // per-element bounds checks and 64-bit index arithmetic: one uniform row pointer per accumulator register plus a
        // 32-bit lane offset.  (The generic path spent ~16 vector instructions per stored element on them: ~10 k cycles
        // per tile beside 32 k cycles of MFMA.)
        const bool full = tile * 32 + 32 <= gP && g.C == C32;
        if constexpr (EPI == RS_STATS_POOL_V) {
          // As RS_STATS_POOL below, values only: per (seed in the tile, crop, column) max over the member rows of
          // sign(gamma)*y.  The member predicates are formed once per (seed, crop) and shared by the NT column tiles; each
          // element then costs one select and half a v_max3 (the form that also tracks the row - compare, two selects, mask logic through SGPR pairs - ran 3x the
          // tile's MFMA time).
          const long long trow = (long long)__builtin_amdgcn_readfirstlane((int)tile) * 32;
          const long long lastrow = trow + 31 < gP ? trow + 31 : gP - 1;
          const int s_lo = g.epi_key[trow] >> 13, s_hi = g.epi_key[lastrow] >> 13;  // wave-uniform (scalar loads)
          int rk[16];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int4 k4 = *reinterpret_cast<const int4 *>(g.epi_key + trow + 4 * h + 8 * i);
            rk[4 * i] = k4.x; rk[4 * i + 1] = k4.y; rk[4 * i + 2] = k4.z; rk[4 * i + 3] = k4.w;
          }
          const unsigned lane_off = (unsigned)(4 * h) * (unsigned)g.ldd + (unsigned)m;
          const int nrow = (int)(gP - trow) - 4 * h;    // rows (r&3) + 8(r>>2) below this are valid
          const bool dense_tile = trow + 32 <= gP && g.ldd == LDD;   // (wave-uniform)
          float *pv = reinterpret_cast<float *>(g.pairs);
          // Pass 1, column tile by column tile: weighted BatchNorm sums, the Y store, then the accumulators are turned
          // into sign(gamma) * y IN PLACE (no second copy of the tile's 128 registers).
#pragma unroll
          for (int q = 0; q < NT; ++q) {
            if (!in(q)) continue;
            float cs = 0.f, cq = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float v = acc[q][r];
              const float wv = (float)((rk[r] >> 4) & 0x1FF) * v;
              cs += wv;
              cq += wv * v;
            }
            {
              const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(cs), __float_as_uint(cs), false, false);
              const auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(cq), __float_as_uint(cq), false, false);
              cs += __uint_as_float(h ? s1[0] : s1[1]);   // the column's other 16 rows sit in lane ^ 32
              cq += __uint_as_float(h ? s2[0] : s2[1]);
              if (h == 0) {
                double *sp = s_st + (size_t)(wave & 3) * 2 * C32 + q * 32 + m;
                __hip_atomic_fetch_add(sp, (double)cs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(sp + C32, (double)cq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            }
            if (g.d) {  // keep Y for the backward (rows >= P are not stored)
              if (dense_tile) {
                // a whole tile at the packed pitch (every tile but the last of a launch): ONE uniform tile pointer, one
                // 32-bit lane offset, and row / column-tile offsets that are compile-time constants (they fit the store's
                // immediate field up to 3 rows + 7 column tiles; four uniform bases cover the rest).  With a run-time
                // pitch and a per-row guard every one of the 128 stores sat in its own branch with its own spilled
                // address: 152 spilled registers, 0.25 GB of scratch written per launch.
                float *tb = g.d + trow * LDD;
                const unsigned lo = (unsigned)(4 * h) * (unsigned)LDD + (unsigned)m;
#pragma unroll
                for (int r = 0; r < 16; ++r) RS_STORE_Y(&tb[lo + (unsigned)(((r & 3) + 8 * (r >> 2)) * LDD + q * 32)], acc[q][r]);
              } else {
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const long long ro = (trow + (r & 3) + 8 * (r >> 2)) * (long long)g.ldd + q * 32;  // wave-uniform
                if ((r & 3) + 8 * (r >> 2) < nrow) g.d[ro + lane_off] = acc[q][r];
              }
              }
            }
            const float sg = g.epi_gamma[q * 32 + m] < 0.f ? -1.f : 1.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] *= sg;
          }
          // Pass 2, (seed, crop) outer: the 16 member predicates of a (seed, crop) are the same for every column tile -
          // formed once (lane masks in scalar register pairs) and reused by the NT tiles, so an element costs one select
          // and half a max; with the predicates re-formed per column tile the epilogue's vector ALU time was about half
          // of the tile's MFMA time.
          for (int sd = s_lo; sd <= s_hi; ++sd) {
            for (int d = 0; d < g.pool_d; ++d) {
              bool mem[16];
#pragma unroll
              for (int r = 0; r < 16; ++r) mem[r] = (rk[r] >> 13) == sd && ((rk[r] >> d) & 1);
              float *slot = pv + ((size_t)(tile + sd) * g.pool_d + d) * LDD + m;
#pragma unroll
              for (int q = 0; q < NT; ++q) {
                if (!in(q)) continue;
                // (v_max3_f32 by hand: fmaxf() of a select's result is preceded by a canonicalising v_max x, x - 216
                // v_max + 32 v_max3 per (seed, crop) where 64 v_max3 do; vector ALU time adds to MFMA time on this chip,
                // tools/mfma_valu_overlap.hip)
                float best = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                  const float x0 = mem[r] ? acc[q][r] : -INFINITY, x1 = mem[r + 1] ? acc[q][r + 1] : -INFINITY;
                  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(best) : "v"(best), "v"(x0), "v"(x1));
                }
                const auto sb = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
                best = fmaxf(best, __uint_as_float(h ? sb[0] : sb[1]));   // the other 16 rows sit in lane ^ 32
                if (h == 0) slot[q * 32] = best;
              }
            }
          }
#pragma unroll
          for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        } else
        if (full) {
          const unsigned lane_off = (unsigned)(4 * h) * (unsigned)g.ldd + (unsigned)m;
          // the tile index is the same in every lane, but derived from threadIdx: tell the compiler (scalar registers,
          // scalar address arithmetic, stores of the form  scalar base + 32-bit lane offset)
          const long long trow = (long long)__builtin_amdgcn_readfirstlane((int)tile) * 32;
          bool weighted = false;
          if constexpr (EPI == RS_STATS) weighted = g.epi_w16 != nullptr;
#pragma unroll
          for (int q = 0; q < NT; ++q) {
            if (!in(q)) continue;
            float cs = 0.f, cq = 0.f, ct[3] = {0.f, 0.f, 0.f};
            if constexpr (BNB && !COEF_REGS) {
              const int col = q * 32 + m;
              ea[q] = g.epi_ab[col];
              eb[q] = g.epi_ab[ctot + col];
              em[q] = g.epi_ab[2 * ctot + col];
              er[q] = g.epi_ab[3 * ctot + col];
            }
            float yq[16];
            if constexpr (BNB && !YPRE) {
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                yq[r] = g.epi_y[(trow + (r & 3) + 8 * (r >> 2)) * (long long)g.ldd + q * 32 + lane_off];
              }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float v = acc[q][r];
              if constexpr (EPI != RS_BNBWD_X) {
                float *dp = g.d + (trow + (r & 3) + 8 * (r >> 2)) * (long long)g.ldd + q * 32;  // wave-uniform
                RS_STORE_Y(&dp[lane_off], v);
              }
              if constexpr (EPI == RS_STATS) {
                if (weighted) {
                  const unsigned pk = (r & 2) ? wq[r >> 2].y : wq[r >> 2].x;
                  const float wv = (float)((r & 1) ? (pk >> 16) : (pk & 0xFFFFu)) * v;
                  cs += wv;
                  cq += wv * v;
                } else {
                  cs += v;
                  cq += v * v;
                }
              }
              if constexpr (BNB) {
                float y = YPRE ? yv[YPRE ? q : 0][r] : yq[r];
                if constexpr (EPI == RS_BNBWD_X) {
                  if (g.gen_w) y = lin3(xr[r][0], xr[r][1], xr[r][2], gx[q][0], gx[q][1], gx[q][2]);
                }
                const float gg = (ea[q] * y + eb[q]) > 0.f ? v : 0.f;
                cs += gg;
                cq += gg * ((y - em[q]) * er[q]);
                if constexpr (EPI == RS_BNBWD_X) {
#pragma unroll
                  for (int j = 0; j < 3; ++j) ct[j] += gg * xr[r][j];
                }
              }
              acc[q][r] = 0.f;
            }
            if constexpr (LSTAT) lds_stat_add(q, cs, cq);
            else if constexpr (EPI != RS_STORE) { dsum[q] += (double)cs; dsq[q] += (double)cq; }
            if constexpr (EPI == RS_BNBWD_X) {
#pragma unroll
              for (int j = 0; j < 3; ++j) dtx[q][j] += (double)ct[j];
            }
          }
        } else {
#pragma unroll
        for (int q = 0; q < NT; ++q) {
          if (!in(q)) continue;
          const int col = q * 32 + m;
          const bool colok = col < g.C;
          float cs = 0.f, cq = 0.f, ct[3] = {0.f, 0.f, 0.f};
          if constexpr (BNB && !COEF_REGS) {
            ea[q] = colok ? g.epi_ab[col] : 0.f;
            eb[q] = colok ? g.epi_ab[ctot + col] : 0.f;
            em[q] = colok ? g.epi_ab[2 * ctot + col] : 0.f;
            er[q] = colok ? g.epi_ab[3 * ctot + col] : 0.f;
          }
          float yq[16];
          if constexpr (BNB && !YPRE) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const long long row = row0 + (r & 3) + 8 * (r >> 2);
              yq[r] = (colok && row < gP) ? g.epi_y[row * g.ldd + col] : 0.f;
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const long long row = row0 + (r & 3) + 8 * (r >> 2);
            float v = acc[q][r];
            if (colok && row < gP) {
              if constexpr (EPI != RS_BNBWD_X) g.d[row * g.ldd + col] = v;  // _X: D itself is not needed
              if constexpr (EPI == RS_STATS) {
                if (g.epi_w16) {  // the row stands for `mult` identical rows of the original batch
                  const unsigned pk = (r & 2) ? wq[r >> 2].y : wq[r >> 2].x;
                  const float wv = (float)((r & 1) ? (pk >> 16) : (pk & 0xFFFFu)) * v;
                  cs += wv;
                  cq += wv * v;
                } else {
                  cs += v;
                  cq += v * v;
                }
              }
              if constexpr (BNB) {
                float y = YPRE ? yv[YPRE ? q : 0][r] : yq[r];
                if constexpr (EPI == RS_BNBWD_X) {
                  if (g.gen_w) y = lin3(xr[r][0], xr[r][1], xr[r][2], gx[q][0], gx[q][1], gx[q][2]);
                }
                const float gg = (ea[q] * y + eb[q]) > 0.f ? v : 0.f;
                cs += gg;
                cq += gg * ((y - em[q]) * er[q]);
                if constexpr (EPI == RS_BNBWD_X) {
#pragma unroll
                  for (int j = 0; j < 3; ++j) ct[j] += gg * xr[r][j];
                }
              }
            }
            acc[q][r] = 0.f;
          }
          if constexpr (LSTAT) lds_stat_add(q, cs, cq);
          else if constexpr (EPI != RS_STORE) { dsum[q] += (double)cs; dsq[q] += (double)cq; }
          if constexpr (EPI == RS_BNBWD_X) {
#pragma unroll
            for (int j = 0; j < 3; ++j) dtx[q][j] += (double)ct[j];
          }
        }
        }
      }
      if (!more) break;
      if constexpr (!A_ASM) {
        if (defer) load_chunk(nxt, ntile, nkc);
#pragma unroll
        for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
      }
      kc = nkc;
      tile = ntile;
    }
  }
  };
  walk(std::false_type{}, tile, main_end, 0, NT);
  if (G >= 2) {
    const long long u = (long long)wave * nwg + bid;
    const int cg = (int)(u % G), per = NT / G;
    walk(std::true_type{}, main_end + u / G, (u / G) < (ntiles - main_end) ? main_end + u / G + 1 : 0, cg * per, (cg + 1) * per);
  }

  if constexpr (EPI != RS_STORE) {
    // per-column totals: lanes l / l+32 share a column, then the 8 waves through LDS (B is dead now)
    __syncthreads();
    if constexpr (LSTAT) {
      double *st = g.stats + (size_t)(blockIdx.x % g.slots) * 2 * ctot;
      for (int i = t; i < 2 * C32; i += RS_TPB) {
        const int which = i / C32, col = i % C32;
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < RS_WAVES / 2; ++w) sum += s_st[(size_t)(w * 2 + which) * C32 + col];
        if (col < g.C) atomicAdd(st + which * ctot + col, sum);
      }
      return;
    }
    double *sd = reinterpret_cast<double *>(lds);  // [wave][NS][C32]
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      double v[NS];
      v[0] = dsum[q];
      v[1] = dsq[q];
      if constexpr (EPI == RS_BNBWD_X) { v[2] = dtx[q][0]; v[3] = dtx[q][1]; v[4] = dtx[q][2]; }
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const double t2 = v[i] + __shfl_xor(v[i], 32);
        if (h == 0) sd[(wave * NS + i) * C32 + q * 32 + m] = t2;
      }
    }
    __syncthreads();
    double *st = g.stats + (size_t)(blockIdx.x % g.slots) * NS * ctot;
    for (int i = t; i < NS * C32; i += RS_TPB) {
      const int which = i / C32, col = i % C32;
      if (col < g.C) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) sum += sd[(w * NS + which) * C32 + col];
        atomicAdd(st + which * ctot + col, sum);
      }
    }
  }
}

// CUs the persistent grid is sized for.  GbGemmOpts.reserved_cus takes r of them out: a caller that keeps a long
// latency-bound kernel resident on a side stream (the next step's furthest-point sampling: one workgroup per cloud,
// 80 KB of LDS each, for ~2 ms) tells the statically partitioned GEMM not to count on those CUs - otherwise the
// workgroups that find no free CU start only when others finish and the launch takes two rounds.
static int num_cus(int reserved) {
  static std::atomic<int> cached{0};  // a device property, not state: every thread computes the same value
  int n = cached.load(std::memory_order_relaxed);
  if (!n) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    cached.store(n, std::memory_order_relaxed);
  }
  return (reserved > 0 && n - reserved >= 16) ? n - reserved : n;
}

template <int NT, int EPI, bool BF, bool GEN3 = false, bool SP = false, int CGS = 1>
static void rs_launch_p(const RsArgs &g, size_t lds_bytes, int blocks_per_cu, hipStream_t s, int reserved) {
  static std::atomic<unsigned long long> attr_set{0};
  auto kern = gemm_rs_kernel<NT, EPI, BF, GEN3, SP, CGS>;
  allow_dynamic_lds(kern, 160 * 1024, attr_set);
  const long long ntiles = (g.P + 31) / 32;   // (with rows_dev: of the row CAPACITY - the grid does not depend on the count)
  long long blocks = ntiles;  // at least one tile per workgroup; all 512 threads stage B either way
  const long long cap = (long long)num_cus(reserved) * blocks_per_cu / CGS;
  if (blocks > cap) blocks = cap;
  if (CGS > 1) blocks = blocks >= 8 ? blocks / 8 * 8 : 8;   // the kernel finds a workgroup's column group in bits 3.. of its index
  hipLaunchKernelGGL(kern, dim3((unsigned)(blocks * CGS)), dim3(RS_TPB), lds_bytes, s, g);
}

// GB_PREC_F32_SPLIT3: which instantiation runs the product as a three-way bf16 split, if any.  nt * 32 columns per
// workgroup, cgs column groups; the three bf16 images of the group's weights must fit the LDS beside the tables.
static bool rs_split_shape(long long P, int R, int C, int epi, bool has_aff, bool gen3, int *nt_out, int *cgs_out,
                           size_t *lds_out) {
  if (P < 16384 || R % 4 != 0 || R < 32 || C % 64 != 0) return false;
  if (epi != RS_STATS && epi != RS_BNBWD && epi != RS_STATS_POOL_V && epi != RS_STORE && epi != RS_BNBWD_X) return false;
  if (epi == RS_BNBWD_X && C != 64) return false;   // (the 3-channel epilogue: 64-wide first layers, one column group)
  if (gen3 && (epi != RS_STATS || !has_aff)) return false;
  const int nch = (R + RS_CH - 1) / RS_CH;
  for (int nt = ((epi == RS_BNBWD || epi == RS_BNBWD_X) ? 2 : 4); nt >= 2; nt -= 2) {   // (dgrad: the y prefetch takes the registers of two column tiles)
    if (C % (nt * 32) != 0) continue;
    const int cgs = C / (nt * 32);
    if (cgs != 1 && cgs != 2) continue;
    if (gen3 && cgs != 1) continue;
    size_t lds = (size_t)nch * RS_CH * nt * 32 * 6 + (has_aff ? (size_t)2 * nch * RS_CH * sizeof(float) : 0);
    if (gen3) lds += (size_t)nch * RS_CH * 4 * sizeof(float);   // the per-k first-layer weights
    if (epi == RS_BNBWD_X) {   // the closing reduction's [waves][5][C32] doubles live in the images' LDS
      const size_t red = (size_t)RS_WAVES * 5 * nt * 32 * sizeof(double);
      if (lds < red) lds = red;
    } else
    if (epi != RS_STORE) lds += (size_t)(RS_WAVES / 2) * 2 * nt * 32 * sizeof(double);   // the fp64 column sums (LSTAT)
    if (lds > 150 * 1024) continue;
    *nt_out = nt; *cgs_out = cgs; *lds_out = lds;
    return true;
  }
  return false;
}

template <int NT, int CGS>
static void rs_launch_split(const RsArgs &g, size_t lds, int epi, bool gen3, hipStream_t s, int reserved) {
  if constexpr (CGS == 1) {
    if (gen3) { rs_launch_p<NT, RS_STATS, false, true, true, 1>(g, lds, 1, s, reserved); return; }
    if constexpr (NT == 2)
      if (epi == RS_BNBWD_X) { rs_launch_p<2, RS_BNBWD_X, false, false, true, 1>(g, lds, 1, s, reserved); return; }
  }
  if (epi == RS_STATS_POOL_V) rs_launch_p<NT, RS_STATS_POOL_V, false, false, true, CGS>(g, lds, 1, s, reserved);
  else if (epi == RS_STATS) rs_launch_p<NT, RS_STATS, false, false, true, CGS>(g, lds, 1, s, reserved);
  else if (epi == RS_BNBWD) {
    if constexpr (NT == 2) rs_launch_p<NT, RS_BNBWD, false, false, true, CGS>(g, lds, 1, s, reserved);   // (rs_split_shape)
  } else rs_launch_p<NT, RS_STORE, false, false, true, CGS>(g, lds, 1, s, reserved);
}

template <int NT, int EPI, bool GEN3 = false>
static void rs_launch(const RsArgs &g, size_t lds_bytes, int blocks_per_cu, hipStream_t s, bool bf16, int reserved) {
  if (bf16) {
    // bf16 image of B = half the bytes (+ the affine table); never below what the closing column reduction
    // ([waves][sums per column][C32] doubles in the same LDS) needs
    const size_t b_fp32 = (size_t)g.nch * RS_CH * NT * 32 * sizeof(float);
    size_t need = lds_bytes - b_fp32 / 2;
    const size_t red = (size_t)RS_WAVES * (EPI == RS_BNBWD_X ? 5 : 2) * NT * 32 * sizeof(double);
    if (EPI != RS_STORE && need < red) need = red;
    rs_launch_p<NT, EPI, true, GEN3>(g, need, blocks_per_cu, s, reserved);
  } else {
    rs_launch_p<NT, EPI, false, GEN3>(g, lds_bytes, blocks_per_cu, s, reserved);
  }
}

// shape part of the dispatch rule (pointer alignment aside); nt_out = column tiles of the instantiation
static bool rs_shape_ok(long long P, int R, int C, int epi, bool has_aff, int *nt_out, size_t *lds_out) {
  if (P < 16384 || R % 4 != 0 || R < 16 || C < 33) return false;
  const int nch = (R + RS_CH - 1) / RS_CH;
  const int tiles_c = (C + 31) / 32;
  const int nt = tiles_c <= 2 ? 2 : tiles_c <= 4 ? 4 : tiles_c == 5 ? 5 : tiles_c <= 8 ? 8 : 0;
  if (!nt) return false;
  if (epi == RS_BNBWD && nt == 8) return false;  // accumulators + y prefetch do not fit the register file
  if (epi == RS_BNBWD_X && nt > 2) return false;  // 64-wide first layers only (registers)
  if (epi == RS_STATS_POOL_V && C != nt * 32) return false;  // the pooled epilogues have no column bounds checks
  // MFMA work wasted on padding must stay small
  if ((long long)nch * RS_CH * nt * 32 * 4 > (long long)R * C * 5) return false;
  const size_t lds_bytes = ((size_t)nch * RS_CH * nt * 32 + (has_aff ? 2 * nch * RS_CH : 0)) * sizeof(float);
  if (lds_bytes > 156 * 1024) return false;
  if (nt_out) *nt_out = nt;
  if (lds_out) *lds_out = lds_bytes;
  return true;
}

bool rs_gemm_try(const float *a, const float *w, float *d, const float *aff, double *stats, int slots,
                 const float *epi_y, const float *epi_ab, long long P, int R, int C, int w_kc, int epi,
                 hipStream_t s, bool bf16, int reserved_cus, const float *epi_x, const uint16_t *epi_w16,
                 const RsPool *pool, const long long *rows_dev, bool split3) {
  int nt = 0;
  size_t lds_bytes = 0;
  const bool sp_gen3 = pool && pool->gen_x && epi == RS_STATS;   // (the generated operand; RS_BNBWD_X carries gen_x for its epilogue)
  if (split3 && !bf16 && (sp_gen3 ? (aff && pool->gen_w) : reinterpret_cast<uintptr_t>(a) % 16 == 0) && (epi == RS_BNBWD_X) == (epi_x != nullptr) &&
      (!w_kc || reinterpret_cast<uintptr_t>(w) % 16 == 0) && !(epi == RS_STATS_POOL_V && (!pool || pool->D < 1 || pool->D > 4))) {
    int cgs = 0;
    if (rs_split_shape(P, R, C, epi, aff != nullptr, sp_gen3, &nt, &cgs, &lds_bytes)) {
      const int nch = (R + RS_CH - 1) / RS_CH;
      RsArgs g = {a, w, d, aff, stats, epi_y, epi_ab, epi_x, epi_w16, pool ? pool->key : nullptr,
                  pool ? pool->gamma : nullptr, pool ? pool->pairs : nullptr, pool ? pool->D : 0,
                  pool ? pool->gen_x : nullptr, pool ? pool->gen_w : nullptr,
                  rows_dev, P, R, C, R, C, w_kc, slots < 1 ? 1 : slots, nch, 1, 1};
      if (nt == 4 && cgs == 1) rs_launch_split<4, 1>(g, lds_bytes, epi, sp_gen3, s, reserved_cus);
      else if (nt == 4) rs_launch_split<4, 2>(g, lds_bytes, epi, sp_gen3, s, reserved_cus);
      else if (cgs == 1) rs_launch_split<2, 1>(g, lds_bytes, epi, sp_gen3, s, reserved_cus);
      else rs_launch_split<2, 2>(g, lds_bytes, epi, sp_gen3, s, reserved_cus);
      return true;
    }
  }
  if (!rs_shape_ok(P, R, C, epi, aff != nullptr, &nt, &lds_bytes)) return false;
  if (epi == RS_STATS_POOL_V && (!pool || C != nt * 32 || pool->D < 1 || pool->D > 4)) return false;
  if (reinterpret_cast<uintptr_t>(a) % 16 != 0 || (w_kc && reinterpret_cast<uintptr_t>(w) % 16 != 0)) return false;
  const int nch = (R + RS_CH - 1) / RS_CH;
  const int stagger = 1, tail_split = 1;  // both measured to help
  RsArgs g = {a, w, d, aff, stats, epi_y, epi_ab, epi_x, epi_w16, pool ? pool->key : nullptr,
              pool ? pool->gamma : nullptr, pool ? pool->pairs : nullptr, pool ? pool->D : 0,
              pool ? pool->gen_x : nullptr, pool ? pool->gen_w : nullptr,
              rows_dev, P, R, C, R, C, w_kc, slots < 1 ? 1 : slots, nch, stagger, tail_split};
  if (epi == RS_STATS_POOL_V || (epi == RS_STATS && nt == 8) || (epi == RS_BNBWD && bf16)) {  // + the fp64 column sums: [wave pair][2][C32]
    lds_bytes += (size_t)(RS_WAVES / 2) * 2 * nt * 32 * sizeof(double);
    if (lds_bytes > 156 * 1024) return false;
  }
  const bool gen3 = pool && pool->gen_x && epi == RS_STATS;
  if (gen3) {  // the A operand is generated from (P,3) rows and a per-k table: 16 more bytes of LDS per reduction index
    if (!aff || !pool->gen_w || (nt != 2 && nt != 4)) return false;
    lds_bytes += (size_t)nch * RS_CH * 4 * sizeof(float);
    if (lds_bytes > 156 * 1024) return false;
  }
  const int bpc = (nt <= 2 && epi != RS_BNBWD && epi != RS_BNBWD_X && lds_bytes <= 78 * 1024) ? 2 : 1;
  if (gen3) {
    if (nt == 2) rs_launch<2, RS_STATS, true>(g, lds_bytes, bpc, s, bf16, reserved_cus);
    else rs_launch<4, RS_STATS, true>(g, lds_bytes, bpc, s, bf16, reserved_cus);
    return true;
  }
  if (epi == RS_BNBWD_X) {
    // the closing per-column reduction reuses the LDS of B as [waves][5][64] doubles = 20 KB: more than the B image of
    // a reduction of <= 64 (16 KB).  (Round 1 launched with the B size only: for a 64 -> 64 second layer - SA1's
    // 3 -> 64 -> 64 -> 128 stack - the tail of that array lay outside the allocation and the five sums were wrong.)
    const size_t red_bytes = (size_t)RS_WAVES * 5 * 2 * 32 * sizeof(double);
    rs_launch<2, RS_BNBWD_X>(g, lds_bytes > red_bytes ? lds_bytes : red_bytes, bpc, s, bf16, reserved_cus);
    return true;
  }
#define GB_RS(NT_)                                                         \
  do {                                                                     \
    if (epi == RS_STATS_POOL_V) rs_launch<NT_, RS_STATS_POOL_V>(g, lds_bytes, bpc, s, bf16, reserved_cus); \
    else if (epi == RS_STATS) rs_launch<NT_, RS_STATS>(g, lds_bytes, bpc, s, bf16, reserved_cus);   \
    else if (epi == RS_BNBWD) rs_launch<NT_, RS_BNBWD>(g, lds_bytes, bpc, s, bf16, reserved_cus); \
    else rs_launch<NT_, RS_STORE>(g, lds_bytes, bpc, s, bf16, reserved_cus);                   \
  } while (0)
  if (nt == 2) GB_RS(2);
  else if (nt == 4) GB_RS(4);
  else if (nt == 5) GB_RS(5);
  else {   // (rs_shape_ok refuses RS_BNBWD at 8 column tiles: no instantiation of it - it would spill 128 registers)
    if (epi == RS_STATS_POOL_V) rs_launch<8, RS_STATS_POOL_V>(g, lds_bytes, bpc, s, bf16, reserved_cus);
    else if (epi == RS_STATS) rs_launch<8, RS_STATS>(g, lds_bytes, bpc, s, bf16, reserved_cus);
    else rs_launch<8, RS_STORE>(g, lds_bytes, bpc, s, bf16, reserved_cus);
  }
#undef GB_RS
  return true;
}

}  // namespace gb

// which kernel gb_gemm_fwd (dgrad = 0) / gb_gemm_dgrad (dgrad = 1) launches for 16-byte aligned operands:
// 1 = gemm_rs_kernel, 0 = gemm_cl_kernel.  Introspection for bench.py's per-kernel roofline accounting.
extern "C" int gb_gemm_uses_rs(long long P, int K, int N, int dgrad, int fused_stats, int has_aff) {
  const int R = dgrad ? N : K, C = dgrad ? K : N;
  // fused_stats = 2 (dgrad only): the first-layer form gb_gemm_dgrad_first; 3 (forward only): gb_gemm_fwd_pool
  const int epi = fused_stats ? (dgrad ? (fused_stats == 2 ? gb::RS_BNBWD_X : gb::RS_BNBWD)
                                       : (fused_stats == 3 ? gb::RS_STATS_POOL_V : gb::RS_STATS)) : gb::RS_STORE;
  return gb::rs_shape_ok(P, R, C, epi, has_aff != 0, nullptr, nullptr) ? 1 : 0;
}

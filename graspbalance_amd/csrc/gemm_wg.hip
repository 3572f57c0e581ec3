// Tall weight-gradient products without LDS staging (gfx950; fp32 MFMA, or bf16 MFMA with fp32 accumulation).
//
//   dW (N,K) += dY (P,N)^T f(X (P,K)),   P >> N, K        (the reference reaches it as the weight gradient of its 1x1
//   convolutions through torch autograd: pytorch_utils.py:61-113 under pointnet2_modules.py:176-188 and modules.py:104-124)
//
// Round 5.  The reduction runs over the ROWS, and both operands are row-contiguous.  v_mfma_f32_32x32x2_f32 takes from
// lane (j = lane & 31, h = lane >> 5) the A element [m = j][k = h] and the B element [k = h][n = j]: with k = the row
// (p + h) and m / n = a column, the 64 lanes of one MFMA operand are two runs of 32 consecutive floats of two consecutive
// rows - exactly what a coalesced global load delivers.  So nothing is staged: a lane loads NTW consecutive dY columns
// (one 16-byte load) and KTW consecutive X columns (one 8-byte load) of its row, which are the operands of NTW x KTW
// MFMAs (column c of a lane's vector belongs to tile c: tile qn holds the output rows n = nb + NTW*m + qn - a permutation
// of the output that only the final write has to know).  Per pair of rows a wave issues 2 loads and 8 MFMAs (512 cycles
// of the matrix pipe); 6 row pairs are requested ahead per wave (36 registers), two waves per SIMD.  No LDS, no barriers
// and no ds_read in the loop; every element of dY and X leaves HBM once (rows shared by workgroups meet in an XCD's L2).
// f: the previous layer's BatchNorm + ReLU applied in registers (2 VALU operations per X element); or X is GENERATED from
// the row's xyz (the folded 3-input first layer, gemm_rs.hip's lin3) - 5 operations per element against 4 x 64 cycles
// of MFMA that consume it.
//
// A workgroup = 8 waves over a run of rows of ONE sub-block of the output (the output is cut into up to four): WN x WK
// waves tile the sub-block (each 128 x 64 or 64 x 64), the other factor PS = 8 / (WN * WK) splits the run of rows.  At the
// end the PS partial outputs are added through LDS (a pairwise tree), the totals are laid out as the sub-block is and
// leave as coalesced fp32 atomics, each workgroup starting at a different offset - the chip retires those at 1.3 TB/s:
// 256 x N*K of them were 26 us of a 250 us launch (measured with the epilogue compiled out), the sub-blocks cut that by
// four.  One workgroup per CU.
//
// BF (GbGemmOpts.precision = GB_PREC_BF16): both operands rounded to bf16 on their way into v_mfma_f32_32x32x16_bf16, fp32
// accumulation.  That instruction takes 8 reduction indices per lane - lane (j, h) supplies rows p + 8h .. p + 8h + 7 of
// its columns - so a step is 16 rows: 8 + 8 loads per lane and slot, 8 MFMAs of 32 cycles.  The product is then bound by
// reading dY and X once (the matrix pipe is ~10 % busy): 4 waves per workgroup with up to 512 registers each keep 2 slots
// (24 KB per wave) in flight.
#include "gb_common.h"
#include "gemm_wg.h"

#include <type_traits>

namespace gb {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_TPB = 512;
constexpr int WG_WAVES = 8;
#ifndef GB_WG_DEPTH
#define GB_WG_DEPTH 6     // slots (row pairs) per wave
#endif
constexpr int WG_DEPTH = GB_WG_DEPTH;
static_assert(WG_DEPTH % 2 == 0 && WG_DEPTH >= 4 && WG_DEPTH <= 8, "s_waitcnt immediates of wg_wait");

enum { WG_PLAIN = 0, WG_AFF = 1, WG_GEN3 = 2 };

struct WgArgs {
  const float *dy, *x, *aff, *gen_x, *gen_w;
  float *dw;
  long long P;
  int N, K;
  const long long *rows_dev;
  int qn, qk;   // the output is cut into qn x qk sub-blocks, one per workgroup of a run of rows
};

__device__ __forceinline__ float wg_lin3(float x, float y, float z, float w0, float w1, float w2) {
  return ((x * w0) + (y * w1)) + (z * w2);  // == gemm_rs.hip's lin3 (the one evaluation order of the folded layer)
}

// The loads are inline assembly with manual s_waitcnt: left to the compiler, every load of the software pipeline below is
// sunk to just in front of its first use ("load, s_waitcnt vmcnt(0), multiply"), i.e. no load is in flight while the
// matrix pipe works.  saddr form: uniform 64-bit base + a 32-bit byte offset per lane.
template <int V> struct WgVec;
template <> struct WgVec<4> { typedef float T __attribute__((ext_vector_type(4))); };
template <> struct WgVec<3> { typedef float T __attribute__((ext_vector_type(3))); };
template <> struct WgVec<2> { typedef float T __attribute__((ext_vector_type(2))); };

// "+v": the destination is the slot's OWN register, in and out - with a plain output the compiler is free to let the load
// write a fresh register and copy it into the loop-carried one right away, i.e. while the load is still in flight.
template <int V>
__device__ __forceinline__ void wg_load(typename WgVec<V>::T &dst, const float *base, unsigned byte_off) {
  if constexpr (V == 4) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(byte_off), "s"(base));
  else if constexpr (V == 3) asm volatile("global_load_dwordx3 %0, %1, %2" : "+v"(dst) : "v"(byte_off), "s"(base));
  else asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(dst) : "v"(byte_off), "s"(base));
}
// the slot's registers pass THROUGH the wait: nothing that reads them can be scheduled in front of it
template <int N, typename A, typename B>
__device__ __forceinline__ void wg_wait(A &a, B &b) {
  static_assert(N >= 0 && N <= 14 && N % 2 == 0, "add the s_waitcnt immediate");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b));
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" : "+v"(a), "+v"(b));
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" : "+v"(a), "+v"(b));
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" : "+v"(a), "+v"(b));
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" : "+v"(a), "+v"(b));
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" : "+v"(a), "+v"(b));
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" : "+v"(a), "+v"(b));
  else asm volatile("s_waitcnt vmcnt(14)" : "+v"(a), "+v"(b));
}

// ... and of a bf16 slot (8 + 8 loads): every register of the slot passes through the wait
template <int N, typename A, typename B>
__device__ __forceinline__ void wg_wait16(A (&a)[8], B (&b)[8]) {
  static_assert(N == 0 || N == 16 || N == 32 || N == 48, "s_waitcnt immediates");
#define GB_WG_SLOT_REGS "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
                        "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" : GB_WG_SLOT_REGS);
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" : GB_WG_SLOT_REGS);
  else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" : GB_WG_SLOT_REGS);
  else asm volatile("s_waitcnt vmcnt(48)" : GB_WG_SLOT_REGS);
#undef GB_WG_SLOT_REGS
}

// SP (GbGemmOpts.precision = GB_PREC_F32_SPLIT3): an fp32 mode on the bf16 skeleton (BF).  Both operands are cut EXACTLY into
// three 8-bit slices of their 24-bit mantissas in registers and the six products of weight >= 2^-16 go to
// v_mfma_f32_32x32x16_bf16, smallest first, term-major over the step's accumulators (csrc/gemm_rs.hip has the account).
__device__ __forceinline__ float wg_trunc16(float x) { return __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }
__device__ __forceinline__ unsigned wg_pack_hi(float lo_elem, float hi_elem) {   // two exactly-bf16 floats -> one register
  return __builtin_amdgcn_perm(__float_as_uint(hi_elem), __float_as_uint(lo_elem), 0x07060302u);
}
struct WgSplit8 { bf16x8 hi, mid, lo; };
__device__ __forceinline__ WgSplit8 wg_split8(const float (&v)[8]) {
  unsigned ph[4], pm[4], pl[4];
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const float x0 = v[i], x1 = v[i + 1];
    const float h0 = wg_trunc16(x0), h1 = wg_trunc16(x1), r0 = x0 - h0, r1 = x1 - h1;
    const float m0 = wg_trunc16(r0), m1 = wg_trunc16(r1), l0 = r0 - m0, l1 = r1 - m1;
    ph[i / 2] = wg_pack_hi(h0, h1);
    pm[i / 2] = wg_pack_hi(m0, m1);
    pl[i / 2] = wg_pack_hi(l0, l1);
  }
  WgSplit8 o;
  __builtin_memcpy(&o.hi, ph, 16);
  __builtin_memcpy(&o.mid, pm, 16);
  __builtin_memcpy(&o.lo, pl, 16);
  return o;
}

template <int NTW, int KTW, int MODE, bool BF, bool SP = false>
__global__ __launch_bounds__(BF ? 256 : WG_TPB) void wgrad_direct_kernel(WgArgs g) {
  static_assert(!SP || BF, "the split runs on the 16-row skeleton");
  extern __shared__ __attribute__((aligned(16))) float s_out[];  // [N][K]
  constexpr int WAVES = BF ? 4 : WG_WAVES, TPB = 64 * WAVES;
  constexpr int RSTEP = BF ? 16 : 2;   // rows of one step of a wave
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int j = lane & 31, h = lane >> 5;
  int gP = (int)g.P;
  if (g.rows_dev) {
    const long long pd = *g.rows_dev;
    gP = pd < g.P ? (pd > 0 ? (int)pd : 0) : (int)g.P;
  }
  const int N = g.N, K = g.K;
  // A workgroup owns ONE of the qn x qk sub-blocks (SN x SK) of the output over a run of rows; the Q workgroups of a run
  // are 8 apart in the grid - the same XCD, started together: the rows they share meet in that XCD's L2.  Fewer, longer
  // runs: Q times fewer atomics at the end.
  const int Q = g.qn * g.qk, SN = N / g.qn, SK = K / g.qk;
  const int WK = SK / (32 * KTW), WNK = (SN / (32 * NTW)) * WK, PS = WAVES / WNK;
  const int sub = wave % WNK, ps = wave / WNK;
  const int bq = (int)(blockIdx.x >> 3) % Q, run = ((int)(blockIdx.x >> 3) / Q) * 8 + (int)(blockIdx.x & 7);
  const int n0 = (bq / g.qk) * SN, k0 = (bq % g.qk) * SK;
  const int nb = n0 + (sub / WK) * 32 * NTW, kb = k0 + (sub % WK) * 32 * KTW;
  // rows of this workgroup (whole steps each), then of this wave's share of them
  const int runs = (int)gridDim.x / Q;    // (the host launches a multiple of 8 Q workgroups when Q > 1)
  const int per_wg = (((gP + runs - 1) / runs) + RSTEP - 1) / RSTEP * RSTEP;
  const int c0 = run * per_wg;
  if (c0 >= gP) return;  // (the whole workgroup: nothing to add)
  const int c1 = c0 + per_wg < gP ? c0 + per_wg : gP;
  const int per_ps = (((c1 - c0 + PS - 1) / PS) + RSTEP - 1) / RSTEP * RSTEP;
  const int r0 = c0 + ps * per_ps;
  const int r1 = r0 + per_ps < c1 ? r0 + per_ps : c1;
  const int last = gP - 1;

  f32x16 acc[NTW][KTW];
#pragma unroll
  for (int a = 0; a < NTW; ++a)
#pragma unroll
    for (int b = 0; b < KTW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  if (r0 < r1) {
    // this lane's columns of X: BatchNorm table / first-layer weights in registers
    float fa[KTW], fb[KTW], gw[MODE == WG_GEN3 ? KTW : 1][3];
    if constexpr (MODE != WG_PLAIN) {
#pragma unroll
      for (int q = 0; q < KTW; ++q) {
        fa[q] = g.aff[kb + KTW * j + q];
        fb[q] = g.aff[K + kb + KTW * j + q];
      }
    }
    if constexpr (MODE == WG_GEN3) {
#pragma unroll
      for (int q = 0; q < KTW; ++q)
#pragma unroll
        for (int e = 0; e < 3; ++e) gw[q][e] = g.gen_w[(kb + KTW * j + q) * 3 + e];
    }
    // (the compiler must have these in registers before the pipeline's own loads start: it does not count those)
    if constexpr (MODE != WG_PLAIN) {
#pragma unroll
      for (int q = 0; q < KTW; ++q) asm volatile("" : "+v"(fa[q]), "+v"(fb[q]));
    }
    if constexpr (MODE == WG_GEN3) {
#pragma unroll
      for (int q = 0; q < KTW; ++q) asm volatile("" : "+v"(gw[q][0]), "+v"(gw[q][1]), "+v"(gw[q][2]));
    }
    constexpr int BV = MODE == WG_GEN3 ? 3 : KTW;
    if constexpr (BF) {
      // slots of 16 rows: 2 in flight while one is multiplied (a fourth slot spilled registers); the split keeps its three
      // slices of the A operand in 48 registers and a step of it is 6x the MFMAs: one slot in flight under each step
      constexpr int D = SP ? 2 : 3;
      typedef typename WgVec<NTW>::T AV;
      typedef typename WgVec<BV>::T BVT;
      AV abuf[D][8];
      BVT bbuf[D][8];
#pragma unroll
      for (int u = 0; u < D; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          abuf[u][i] = 0.f;
          bbuf[u][i] = 0.f;
        }
      const unsigned a_pitch = (unsigned)N * 4u, b_pitch = (MODE == WG_GEN3 ? 3u : (unsigned)K) * 4u;
      const unsigned a_lane = (unsigned)(nb + NTW * j) * 4u, b_lane = MODE == WG_GEN3 ? 0u : (unsigned)(kb + KTW * j) * 4u;
      int next_row = r0 + 8 * h;   // this lane's first row of the next slot to request (its rows: + 0 .. 7)
      auto request = [&](AV (&a)[8], BVT (&b)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const unsigned rc = (unsigned)(next_row + i < last ? next_row + i : last);
          wg_load<NTW>(a[i], g.dy, rc * a_pitch + a_lane);
          if constexpr (MODE == WG_GEN3) wg_load<3>(b[i], g.gen_x, rc * b_pitch);
          else wg_load<KTW>(b[i], g.x, rc * b_pitch + b_lane);
        }
        next_row += 16;
      };
      // one step: wait for the slot (N younger loads may stay in flight), operands to bf16, 8 MFMAs
      auto multiply = [&](auto younger, AV (&a)[8], BVT (&b)[8], int base, bool masked) {
        constexpr int YOUNGER = decltype(younger)::value;
        wg_wait16<YOUNGER>(a, b);
        if constexpr (SP) {
          WgSplit8 as[NTW];
#pragma unroll
          for (int qa = 0; qa < NTW; ++qa) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (masked && base + 8 * h + i >= r1) ? 0.f : a[i][qa];
            as[qa] = wg_split8(v);
          }
#pragma unroll
          for (int qb = 0; qb < KTW; ++qb) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              if constexpr (MODE == WG_GEN3) v[i] = wg_lin3(b[i][0], b[i][1], b[i][2], gw[qb][0], gw[qb][1], gw[qb][2]);
              else v[i] = b[i][qb];
              if constexpr (MODE != WG_PLAIN) {
                const float z = fa[qb] * v[i] + fb[qb];
                v[i] = z > 0.f ? z : 0.f;
              }
            }
            const WgSplit8 bs = wg_split8(v);
#pragma unroll
            for (int qa = 0; qa < NTW; ++qa) acc[qa][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[qa].lo, bs.hi, acc[qa][qb], 0, 0, 0);
#pragma unroll
            for (int qa = 0; qa < NTW; ++qa) acc[qa][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[qa].hi, bs.lo, acc[qa][qb], 0, 0, 0);
#pragma unroll
            for (int qa = 0; qa < NTW; ++qa) acc[qa][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[qa].mid, bs.mid, acc[qa][qb], 0, 0, 0);
#pragma unroll
            for (int qa = 0; qa < NTW; ++qa) acc[qa][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[qa].mid, bs.hi, acc[qa][qb], 0, 0, 0);
#pragma unroll
            for (int qa = 0; qa < NTW; ++qa) acc[qa][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[qa].hi, bs.mid, acc[qa][qb], 0, 0, 0);
#pragma unroll
            for (int qa = 0; qa < NTW; ++qa) acc[qa][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[qa].hi, bs.hi, acc[qa][qb], 0, 0, 0);
          }
          return;
        }
#pragma unroll
        for (int qb = 0; qb < KTW; ++qb) {
          bf16x8 b8;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float v;
            if constexpr (MODE == WG_GEN3) v = wg_lin3(b[i][0], b[i][1], b[i][2], gw[qb][0], gw[qb][1], gw[qb][2]);
            else v = b[i][qb];
            if constexpr (MODE != WG_PLAIN) {
              const float z = fa[qb] * v + fb[qb];
              v = z > 0.f ? z : 0.f;
            }
            b8[i] = (__bf16)v;
          }
#pragma unroll
          for (int qa = 0; qa < NTW; ++qa) {
            bf16x8 a8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              float v = a[i][qa];
              if (masked && base + 8 * h + i >= r1) v = 0.f;
              a8[i] = (__bf16)v;
            }
            acc[qa][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[qa][qb], 0, 0, 0);
          }
        }
      };
#pragma unroll
      for (int u = 0; u < D; ++u) request(abuf[u], bbuf[u]);
      int base = r0;
      // every trip but the last requests again what it has multiplied; the last one counts its waits down (see below)
      for (; base + 16 * D < r1; base += 16 * D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
          multiply(std::integral_constant<int, 16 * (D - 1)>{}, abuf[u], bbuf[u], base + 16 * u, false);
          request(abuf[u], bbuf[u]);
        }
      }
      if constexpr (D == 3) {
        multiply(std::integral_constant<int, 32>{}, abuf[0], bbuf[0], base, true);
        multiply(std::integral_constant<int, 16>{}, abuf[1], bbuf[1], base + 16, true);
        multiply(std::integral_constant<int, 0>{}, abuf[2], bbuf[2], base + 32, true);
      } else {
        multiply(std::integral_constant<int, 16>{}, abuf[0], bbuf[0], base, true);
        multiply(std::integral_constant<int, 0>{}, abuf[1], bbuf[1], base + 16, true);
      }
      static_assert(D == 2 || D == 3, "the last trip is written out");
    } else {
    constexpr int D = WG_DEPTH;
    typedef typename WgVec<NTW>::T AV;
    typedef typename WgVec<BV>::T BVT;
    AV abuf[D];
    BVT bbuf[D];
#pragma unroll
    for (int u = 0; u < D; ++u) {   // (the loads take their destination as an in/out operand)
      abuf[u] = 0.f;
      bbuf[u] = 0.f;
    }
    const unsigned a_pitch2 = (unsigned)N * 8u, b_pitch2 = (MODE == WG_GEN3 ? 3u : (unsigned)K) * 8u;  // two rows, in bytes
    // running byte offsets of the NEXT row to request (slots are refilled in step order: always 2 rows further), clamped to
    // the tensor's last row: rows past the end of this wave's share are requested from valid addresses and multiplied by 0
    unsigned a_next = (unsigned)(r0 + h) * (a_pitch2 / 2) + (unsigned)(nb + NTW * j) * 4u;
    unsigned b_next = (unsigned)(r0 + h) * (b_pitch2 / 2) + (MODE == WG_GEN3 ? 0u : (unsigned)(kb + KTW * j) * 4u);
    const unsigned a_max = (unsigned)last * (a_pitch2 / 2) + (unsigned)(nb + NTW * j) * 4u;
    const unsigned b_max = (unsigned)last * (b_pitch2 / 2) + (MODE == WG_GEN3 ? 0u : (unsigned)(kb + KTW * j) * 4u);
    auto request = [&](AV &a, BVT &b) {
      wg_load<NTW>(a, g.dy, a_next < a_max ? a_next : a_max);
      if constexpr (MODE == WG_GEN3) wg_load<3>(b, g.gen_x, b_next < b_max ? b_next : b_max);
      else wg_load<KTW>(b, g.x, b_next < b_max ? b_next : b_max);
      a_next += a_pitch2;
      b_next += b_pitch2;
    };
    // the X half of a slot -> the B operands of its step (BatchNorm + ReLU, or the generated layer), in four stages of
    // independent vector instructions: the pipeline below places one stage behind each quarter of the running step's MFMAs
    auto stage = [&](int st, BVT &b, float (&tmp)[KTW][2], float (&bv)[KTW]) {
#pragma unroll
      for (int q = 0; q < KTW; ++q) {
        if constexpr (MODE == WG_GEN3) {   // wg_lin3(b, gw[q]), then relu(fa * . + fb)
          if (st == 0) { tmp[q][0] = b[0] * gw[q][0]; tmp[q][1] = b[1] * gw[q][1]; }
          else if (st == 1) { tmp[q][0] = tmp[q][0] + tmp[q][1]; tmp[q][1] = b[2] * gw[q][2]; }
          else if (st == 2) { tmp[q][0] = tmp[q][0] + tmp[q][1]; tmp[q][0] = fa[q] * tmp[q][0]; }
          else { const float z = tmp[q][0] + fb[q]; bv[q] = z > 0.f ? z : 0.f; }
        } else if constexpr (MODE == WG_AFF) {
          if (st == 0) tmp[q][0] = fa[q] * b[q];
          else if (st == 1) tmp[q][0] = tmp[q][0] + fb[q];
          else if (st == 2) bv[q] = tmp[q][0] > 0.f ? tmp[q][0] : 0.f;
        } else {
          if (st == 0) bv[q] = b[q];
        }
      }
    };
    auto prepare = [&](BVT &b, float (&bv)[KTW]) {
      float tmp[KTW][2];
#pragma unroll
      for (int st = 0; st < 4; ++st) stage(st, b, tmp, bv);
    };
#pragma unroll
    for (int u = 0; u < D; ++u) request(abuf[u], bbuf[u]);
    // Software pipeline.  Step u multiplies slot u: the dY half feeds the MFMAs straight from the slot's registers, the X
    // half through `prepare`.  While the 8 MFMAs of step u run, slot u + 1 is waited for and prepared; slot u is requested
    // again right behind its MFMAs (they read their operands when they issue).  With the preparation in FRONT of its own
    // MFMAs the two waves of a SIMD - same program, same start - stay in lockstep and the matrix pipe idles while both
    // prepare (measured: the BatchNorm form 15 % slower than the plain one for 6 more vector instructions per 512 cycles).
    // NOTHING may copy a slot register between its request and its wait - the compiler does not know the load is in
    // flight.  The slots are therefore only ever touched by the tied in/out operands of the asm statements and read by
    // MFMA / prepare after their wait; tools/wg_check_isa.py walks the generated code for exactly that.
    static_assert(D % 2 == 0 && D >= 4, "the two B operand sets swap roles every step");
    float bv[2][KTW];
    wg_wait<2 * (D - 1)>(abuf[0], bbuf[0]);
    prepare(bbuf[0], bv[0]);
    // D steps: rows base + 2u + h.  LAST: the wave's final steps (rows >= r1 multiply by zero) - it requests nothing: the
    // registers of a request nobody consumes are free for the compiler to reuse while the load is still on its way (that
    // was a memory fault: an address register overwritten by a late arrival), so the waits count down instead.
    auto trip = [&](auto last_trip, int base) {
      constexpr bool LAST = decltype(last_trip)::value;
#pragma unroll
      for (int u = 0; u < D; ++u) {
        const int c = u & 1, v = (u + 1) % D;
        if (!LAST || u + 1 < D) {   // (compile time after unrolling)
          if constexpr (LAST) {
            // slots u + 2 .. D - 1 are younger (2 loads each)
            if (u == 0) wg_wait<2 * (D - 2)>(abuf[v], bbuf[v]);
            else if (u == 1) wg_wait<(D >= 3 ? 2 * (D - 3) : 0)>(abuf[v], bbuf[v]);
            else if (u == 2) wg_wait<(D >= 4 ? 2 * (D - 4) : 0)>(abuf[v], bbuf[v]);
            else if (u == 3) wg_wait<(D >= 5 ? 2 * (D - 5) : 0)>(abuf[v], bbuf[v]);
            else if (u == 4) wg_wait<(D >= 6 ? 2 * (D - 6) : 0)>(abuf[v], bbuf[v]);
            else wg_wait<0>(abuf[v], bbuf[v]);
          } else {
            wg_wait<2 * (D - 2)>(abuf[v], bbuf[v]);   // slots u + 2 .. u + D - 1 (2 loads each) may still be in flight
          }
        }
        const bool prep = !LAST || u + 1 < D;
        float av[NTW];
        if constexpr (LAST) {
          const bool ok = base + 2 * u + h < r1;
#pragma unroll
          for (int q = 0; q < NTW; ++q) av[q] = ok ? abuf[u][q] : 0.f;
        } else {
#pragma unroll
          for (int q = 0; q < NTW; ++q) av[q] = abuf[u][q];
        }
        // a quarter of the step's MFMAs, then one stage of the next step's operands: the fences keep the scheduler from
        // sinking all of the preparation to just in front of the MFMAs that consume it (where the matrix pipe waits for
        // the chain multiply - add - max; measured +20 % on the BatchNorm form)
        float tmp[KTW][2];
        constexpr int QM = NTW * KTW / 4;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
#pragma unroll
          for (int i = st * QM; i < (st + 1) * QM; ++i)
            acc[i / KTW][i % KTW] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i / KTW], bv[c][i % KTW], acc[i / KTW][i % KTW], 0, 0, 0);
          if (prep) stage(st, bbuf[v], tmp, bv[c ^ 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!LAST) request(abuf[u], bbuf[u]);
      }
    };
    // every trip but the last one requests; the last one (always run: it also drains the slots) has D or fewer steps
    int base = r0;
    for (; base + 2 * D < r1; base += 2 * D) trip(std::false_type{}, base);
    trip(std::true_type{}, base);
    }
  }

  // The PS partial outputs are added pairwise through LDS (a tree: PS/2, PS/4 .. 1 writers per round, everybody else adds
  // its partner's block into its registers; blocks in the accumulators' own layout, 16-byte accesses without bank
  // conflicts).  The PS sequential rounds of read-modify-write this replaces took 15 us of a 95 us launch at PS = 8.
  {
    constexpr int BLK = NTW * KTW * 16 * 64;   // floats of one wave's accumulators
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    for (int s2 = PS / 2; s2 >= 1; s2 >>= 1) {
      if (ps >= s2 && ps < 2 * s2) {
        f32x4 *blk = reinterpret_cast<f32x4 *>(s_out + (size_t)((ps - s2) * WNK + sub) * BLK) + lane;
#pragma unroll
        for (int a = 0; a < NTW; ++a)
#pragma unroll
          for (int b = 0; b < KTW; ++b)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
              blk[((a * KTW + b) * 4 + r4) * 64] =
                  f32x4{acc[a][b][4 * r4], acc[a][b][4 * r4 + 1], acc[a][b][4 * r4 + 2], acc[a][b][4 * r4 + 3]};
      }
      __syncthreads();
      if (ps < s2) {
        const f32x4 *blk = reinterpret_cast<const f32x4 *>(s_out + (size_t)(ps * WNK + sub) * BLK) + lane;
#pragma unroll
        for (int a = 0; a < NTW; ++a)
#pragma unroll
          for (int b = 0; b < KTW; ++b)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              const f32x4 v = blk[((a * KTW + b) * 4 + r4) * 64];
              acc[a][b][4 * r4] += v[0];
              acc[a][b][4 * r4 + 1] += v[1];
              acc[a][b][4 * r4 + 2] += v[2];
              acc[a][b][4 * r4 + 3] += v[3];
            }
      }
      __syncthreads();
    }
  }
  // the totals (waves with ps = 0) in the sub-block's layout [SN][SK]:
  // acc[a][b][r] = dW[nb + NTW*((r&3) + 8*(r>>2) + 4*h) + a][kb + KTW*j + b]
  if (ps == 0) {
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float *row = s_out + (nb - n0 + NTW * ((r & 3) + 8 * (r >> 2) + 4 * h) + a) * SK + (kb - k0) + KTW * j;
#pragma unroll
        for (int b = 0; b < KTW; ++b) row[b] = acc[a][b][r];
      }
  }
  __syncthreads();
  const int total = SN * SK, sk_log2 = 31 - __builtin_clz((unsigned)SK);   // (SK is a power of two: the host checks)
  const int rot = (int)(((unsigned)run * 1024u) % (unsigned)total);  // workgroups finish together: not all on one line
  for (int i = t; i < total; i += TPB) {
    int idx = i + rot;
    if (idx >= total) idx -= total;
    atomicAdd(g.dw + (size_t)(n0 + (idx >> sk_log2)) * K + k0 + (idx & (SK - 1)), s_out[idx]);
  }
}

static int wg_num_cus(int reserved) {
  static std::atomic<int> cached{0};  // a device property, not state
  int n = cached.load(std::memory_order_relaxed);
  if (!n) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    cached.store(n, std::memory_order_relaxed);
  }
  return (reserved > 0 && n - reserved >= 16) ? n - reserved : n;
}

#ifndef GB_WG_MAXQ
#define GB_WG_MAXQ 4
#endif

// How a product is cut: wave tiles of 128 x 64 (form 1: NTW 4, KTW 2) or 64 x 64 (form 2), the output in qn x qk
// sub-blocks (one per workgroup of a run of rows), a sub-block's wave tiles a divisor of the workgroup's waves.
struct WgCut {
  int form, qn, qk;
};

static bool wg_cut(long long P, int K, int N, bool gen, int waves, long long blocks, WgCut *cut) {
  if (P < 1 || N < 64 || K < 64 || N % 64 != 0 || K % 64 != 0) return false;
  if (P * (long long)(N > K ? N : K) >= (1LL << 30) - (1 << 16)) return false;   // 32-bit byte offsets (+ the look-ahead)
  if (gen && K != 64) return false;
  if ((K & (K - 1)) != 0) return false;                                        // the closing loop splits an index by shifts
  for (int form = 1; form <= 2; ++form) {
    const int ntw = form == 1 ? 4 : 2, ktw = 2;
    if (N % (32 * ntw) != 0) continue;
    int qn = 1, qk = 1;
    auto sets = [&]() { return (N / qn / (32 * ntw)) * (K / qk / (32 * ktw)); };
    auto cut_once = [&]() {   // halve N first (dY is the wider operand of the shapes this serves)
      if ((N / qn) % (2 * 32 * ntw) == 0 && qn <= qk) { qn *= 2; return true; }
      if ((K / qk) % (2 * 32 * ktw) == 0) { qk *= 2; return true; }
      if ((N / qn) % (2 * 32 * ntw) == 0) { qn *= 2; return true; }
      return false;
    };
    // as many sub-blocks as leave a run of rows worth its prologue, the grid a multiple of 8 Q ...
    while (qn * qk * 2 <= GB_WG_MAXQ && blocks >= 8 * qn * qk * 2 && P / (blocks / (qn * qk * 2)) >= 512)
      if (!cut_once()) break;
    // ... and at least as many as make a sub-block fit a workgroup (its wave tiles, its 128 KB of LDS)
    while (sets() > waves || (long long)(N / qn) * (K / qk) > 32768)
      if (qn * qk * 2 > 64 || blocks < 8 * qn * qk * 2 || !cut_once()) break;
    const int t = sets();
    if (t > waves || waves % t != 0 || (long long)(N / qn) * (K / qk) > 32768) continue;
    *cut = {form, qn, qk};
    return true;
  }
  return false;
}

// The grid and wave count wg_wgrad_try launches a product with: the 16-row skeleton (bf16 and GB_PREC_F32_SPLIT3) runs 4
// waves per workgroup, the fp32 MFMA form WG_WAVES; one workgroup per CU, never more than the rows give a loop trip each.
static long long wg_grid(long long P, bool skeleton16, int reserved_cus, int *waves) {
  *waves = skeleton16 ? 4 : WG_WAVES;
  long long blocks = wg_num_cus(reserved_cus);
  const long long rows_per_trip = (skeleton16 ? 16 * 3 : 2 * WG_DEPTH) * *waves;   // >= one loop trip per wave
  const long long most = (P + rows_per_trip - 1) / rows_per_trip;
  if (blocks > most) blocks = most;
  return blocks < 1 ? 1 : blocks;
}

bool wg_wgrad_suits(long long P, int K, int N, bool gen, bool skeleton16, int reserved_cus) {
  WgCut c;
  int waves;
  const long long blocks = wg_grid(P, skeleton16, reserved_cus, &waves);
  return wg_cut(P, K, N, gen, waves, blocks, &c);
}

template <int NTW, int KTW, int MODE, bool BF, bool SP = false>
static void wg_launch(WgArgs g, long long blocks, hipStream_t s) {
  static std::atomic<unsigned long long> attr_set{0};
  auto kern = wgrad_direct_kernel<NTW, KTW, MODE, BF, SP>;
  allow_dynamic_lds(kern, 160 * 1024, attr_set);
  constexpr int WAVES = BF ? 4 : WG_WAVES;
  const int Q = g.qn * g.qk;
  if (Q > 1) blocks = blocks / (8 * Q) * (8 * Q);
  const int wnk = (g.N / g.qn / (32 * NTW)) * (g.K / g.qk / (32 * KTW)), half_ps = WAVES / wnk / 2;
  // the sub-block in its own layout, or the PS/2 blocks of the first round of the closing tree (never more than 128 KB)
  const size_t lds = (size_t)(g.N / g.qn) * (g.K / g.qk) * sizeof(float) * (half_ps > 1 ? half_ps : 1);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * WAVES), lds, s, g);
}

bool wg_wgrad_try(const float *dy, const float *x, const float *aff, const float *gen_x, const float *gen_w, float *dw,
                  long long P, int K, int N, const long long *rows_dev, int reserved_cus, hipStream_t s, bool bf16,
                  bool split3) {
  const bool gen = gen_x != nullptr;
  if (split3 && !bf16) bf16 = true;   // the 16-row skeleton (4 waves, 3 slots); the arithmetic is chosen at the launch below
  else split3 = false;
  if (!dy || !dw || (gen ? (!gen_w || !aff || x) : !x)) return false;
  auto al16 = [](const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  if (!al16(dy) || (x && !al16(x)) || !al16(dw)) return false;
  int waves;
  const long long blocks = wg_grid(P, bf16, reserved_cus, &waves);
  WgCut cut;
  if (!wg_cut(P, K, N, gen, waves, blocks, &cut)) return false;
  WgArgs g = {dy, x, aff, gen_x, gen_w, dw, P, N, K, rows_dev, cut.qn, cut.qk};
  const int mode = gen ? WG_GEN3 : (aff ? WG_AFF : WG_PLAIN);
#define GB_WG3(NTW_, KTW_, BF_)                                                \
  do {                                                                         \
    if (mode == WG_GEN3) wg_launch<NTW_, KTW_, WG_GEN3, BF_>(g, blocks, s);    \
    else if (mode == WG_AFF) wg_launch<NTW_, KTW_, WG_AFF, BF_>(g, blocks, s); \
    else wg_launch<NTW_, KTW_, WG_PLAIN, BF_>(g, blocks, s);                   \
  } while (0)
#define GB_WG3S(NTW_, KTW_)                                                          \
  do {                                                                              \
    if (mode == WG_GEN3) wg_launch<NTW_, KTW_, WG_GEN3, true, true>(g, blocks, s);    \
    else if (mode == WG_AFF) wg_launch<NTW_, KTW_, WG_AFF, true, true>(g, blocks, s); \
    else wg_launch<NTW_, KTW_, WG_PLAIN, true, true>(g, blocks, s);                   \
  } while (0)
#define GB_WG(NTW_, KTW_)                    \
  do {                                       \
    if (split3) GB_WG3S(NTW_, KTW_);         \
    else if (bf16) GB_WG3(NTW_, KTW_, true); \
    else GB_WG3(NTW_, KTW_, false);          \
  } while (0)
  if (cut.form == 1) GB_WG(4, 2); else GB_WG(2, 2);
#undef GB_WG
#undef GB_WG3S
#undef GB_WG3
  return true;
}

}  // namespace gb

// fp32 MFMA GEMM for the FEW-ROW products of the channel-last SharedMLP path (gfx950, v_mfma_f32_32x32x2_f32): the
// C -> 4C -> C pointwise pairs of the 15 InvResMLP blocks on 1 024 .. 8 192 rows, the feature-propagation stacks and the
// Conv1d grasp heads on 4 096 / 16 384 rows (reference drp.py:97-106, pointnet2_modules.py:402-435, modules.py:49-175) -
// forward, dgrad and wgrad.  These launches have 64 .. 1 024 output tiles and 4 .. 32 reduction steps each: what bounds
// them is not the matrix cores but how long a workgroup waits for its operands (csrc/gemm_cl.hip fetches one 16-deep step
// ahead through registers: 512 MFMA cycles of cover against 500 - 900 cycles of L2 / HBM latency, waves parked at
// s_waitcnt 30 - 59 % of the time - profiles/r03_fewrow_sq_counters.txt).
//
// Structure: an LDS ring of S = 4 stages filled by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write), three
// 32-deep reduction steps in flight, ONE raw s_barrier per step.
//   step t:  s_waitcnt vmcnt(pieces of the stages after t)   - the issuing wave's share of stage t has landed
//            s_barrier                                       - every wave's share has; and everybody is done reading stage t-1
//            issue the DMA of stage t+3 into the slot of stage t-1
//            ds_read the fragments of stage t, MFMA
// The prologue issues three stages (and the BatchNorm table of the operand prologue) before the first wait, so the cold
// first loads of a short-lived workgroup overlap each other instead of the first steps.
//
// LDS images (an LDS-DMA instruction writes 64 x 16 bytes lane-linearly: the image is shaped by choosing each lane's
// SOURCE address):
//   reduction-contiguous operand (X (P,K) of forward, dY (P,N) of dgrad, W (N,K) of forward): [row][8 chunks of 4 floats],
//     chunk c of row r at position c ^ ((r >> 1) & 7): a lane's ds_read_b128 of "its" chunk is bank-conflict free.  Lane
//     (m, h) of a 32-row block reads chunks 2c' + h, c' = 0..3, and feeds element e of chunk c' to MFMA slice 4c' + e: the
//     reduction index that lane half h contributes to slice (c', e) is 4 (2c' + h) + e - any bijection is legal as long as
//     both operands use it.
//   row-contiguous operand (W (N,K) of dgrad: element (k, n) at w[n K + k]; dY and X of wgrad): [reduction index][cols] as
//     in memory; lane (j, h) reads one float of row "the reduction index of its slice": 32 consecutive floats per half wave.
// Operand prologue relu(a x + b) (the previous layer's BatchNorm + ReLU) is applied to the fragments after the ds_read;
// the (a, b) table travels through LDS-DMA as well (an ordinary load in flight would make the compiler drain the ring).
// Epilogues as csrc/gemm_cl.hip: store, store + BatchNorm column sums (fp64 atomics over slot rows), store + the
// BatchNorm-backward sums of the previous layer, fp32 atomics (split-K wgrad), partial products of a split reduction.
//
// Eligibility (ring_gemm_try returns false otherwise and the caller uses csrc/gemm_cl.hip): fp32 precision, reduction
// length a multiple of 32, leading dimensions and tile-row counts multiples of 4, 16-byte aligned operands, no row weights.
#include <type_traits>

#include "gb_common.h"
#include "gemm_ring.h"

namespace gb {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int RG_TPB = 256, RG_BK = 32, RG_STAGES = 4;

struct RingOp {
  const float *src;
  long long rows;    // valid tile-row indices [0, rows)
  long long ld;      // leading dimension in floats (of the row index for RK_KC, of the reduction index for RK_RC)
  const float *aff;  // optional: RK_KC [a(red), b(red)] by reduction index ; RK_RC [a(rows), b(rows)] by tile-row index
};

struct RingArgs {
  RingOp a, b;
  float *d;
  long long ldd;
  long long red;      // reduction length (multiple of 32)
  long long kchunk;   // reduction indices per blockIdx.y (multiple of 32)
  long long dchunk;   // != 0: chunk blockIdx.y stores its partial product at d + blockIdx.y * dchunk
  double *stats;      // RG_STATS / RG_BNBWD: fp64 [stat_slots][2 * b.rows]
  int stat_slots;
  const float *epi_y;   // RG_BNBWD: (a.rows, b.rows) pitch ldd, pre-BatchNorm output of the layer D is the gradient of
  const float *epi_ab;  // RG_BNBWD: [a, b, mean, rstd](b.rows)
  int tiles_n;
};

#define RG_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

template <int N>
__device__ __forceinline__ void rg_wait_vm() {
  if constexpr (N == 0) RG_WAIT_VM(0);
  else if constexpr (N == 2) RG_WAIT_VM(2);
  else if constexpr (N == 4) RG_WAIT_VM(4);
  else if constexpr (N == 6) RG_WAIT_VM(6);
  else if constexpr (N == 8) RG_WAIT_VM(8);
  else if constexpr (N == 12) RG_WAIT_VM(12);
  else if constexpr (N == 16) RG_WAIT_VM(16);
  else if constexpr (N == 24) RG_WAIT_VM(24);
  else static_assert(N < 0, "add the count");
}

#define RG_WAIT_LGKM(N) asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory")

template <int N>
__device__ __forceinline__ void rg_wait_lgkm() {
  if constexpr (N == 0) RG_WAIT_LGKM(0);
  else if constexpr (N == 2) RG_WAIT_LGKM(2);
  else if constexpr (N == 3) RG_WAIT_LGKM(3);
  else if constexpr (N == 4) RG_WAIT_LGKM(4);
  else if constexpr (N == 5) RG_WAIT_LGKM(5);
  else if constexpr (N == 6) RG_WAIT_LGKM(6);
  else if constexpr (N == 8) RG_WAIT_LGKM(8);
  else if constexpr (N == 10) RG_WAIT_LGKM(10);
  else if constexpr (N == 12) RG_WAIT_LGKM(12);
  else if constexpr (N > 12) RG_WAIT_LGKM(12);   // (the counter has 4 bits: waiting for fewer outstanding is only stricter)
  else static_assert(N < 0, "add the count");
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 rg_ds128(unsigned addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

__device__ __forceinline__ float rg_ds32(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

// ... with the offset as an instruction immediate: the row-contiguous (RK_RC) operands' fragments are 16 single reads per
// 32 x 32 block and step at compile-time distances from ONE per-lane address - left as computed addresses each read cost
// a v_add_u32 (32 vector instructions per step of a 64 x 64 wgrad tile beside its 16 MFMAs, and vector-ALU time ADDS to
// matrix time on this chip: the tile saturated at 0.67 of the matrix cores however long the reduction)
template <int OFF>
__device__ __forceinline__ float rg_ds32o(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

__device__ __forceinline__ void rg_glds16(const float *src, float *lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                   (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

// KA / KB: RK_KC or RK_RC.  BM x BN block tile, 4 waves as 2 x 2, each (BM/2) x (BN/2) = MT x NT MFMA tiles of 32 x 32.
// AFFA: operand A (RK_KC) carries a.aff by reduction index; AFFB: operand B (RK_RC) carries b.aff by tile-row index.
// BF (GbGemmOpts.precision = GB_PREC_BF16, round 5): the fragments - fp32 in LDS as ever - are rounded to bf16 in registers
// (after the operand prologue) and two reduction groups of four feed one v_mfma_f32_32x32x16_bf16: 2 matrix instructions
// per 32-deep step and tile pair instead of 16, fp32 accumulation, same epilogues.
// The workgroup's program: tile (bx, by) of product `g`, `lds` = ALL dynamic LDS of the kernel (a second LDS object would
// make the compiler drain the DMA ring before every read).
template <int KA, int KB, int BM, int BN, int EPI, bool AFFA, bool AFFB, bool BF = false>
__device__ __forceinline__ void ring_tile(const RingArgs &g, const unsigned bx, const unsigned by, float *lds) {
  constexpr int A_FL = BM * RG_BK, B_FL = BN * RG_BK, ST_FL = A_FL + B_FL;
  constexpr int PA = BM / 8, PB = BN / 8, NPW = (PA + PB) / 4;    // 1 KB pieces per stage: A, B; per wave
  constexpr int MT = BM / 64, NT = BN / 64;
  static_assert(!AFFA || KA == RK_KC, "a.aff is indexed by the reduction index");
  static_assert(!AFFB || KB == RK_RC, "b.aff is indexed by the tile-row index");
  float *tab = lds + RG_STAGES * ST_FL;   // AFFA: [2][tstride] (a, b of the chunk's reduction indices) ; AFFB: [2][256]
  const int tstride = (int)((g.kchunk + 255) / 256 * 256);

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform: operand selects below stay scalar
  const int m = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const long long m0 = (long long)(bx / g.tiles_n) * BM;
  const long long n0 = (long long)(bx % g.tiles_n) * BN;
  const long long kbeg = (long long)by * g.kchunk;
  long long kend = kbeg + g.kchunk;
  if (kend > g.red) kend = g.red;
  const int T = (int)((kend - kbeg) / RG_BK);   // reduction steps of this workgroup (>= 1: host-checked)
  float *dout = g.d + (long long)by * g.dchunk;

  // ---- this lane's source address for each of the wave's NPW pieces of a stage (advanced by one step per issue)
  const float *src[NPW];
  long long adv[NPW];
  int dst_off[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int pid = wave + 4 * i;
    const bool isA = pid < PA;
    const int p = isA ? pid : pid - PA;
    const RingOp &op = isA ? g.a : g.b;
    const int kind = isA ? KA : KB;
    const int q = p * 64 + lane;
    dst_off[i] = (isA ? 0 : A_FL) + p * 256;
    if (kind == RK_KC) {
      const int r = q >> 3, c = (q & 7) ^ ((r >> 1) & 7);
      long long row = (isA ? m0 : n0) + r;
      if (row > op.rows - 1) row = op.rows - 1;            // rows beyond the operand: a valid duplicate, masked at the end
      src[i] = op.src + row * op.ld + kbeg + 4 * c;
      adv[i] = RG_BK;
    } else {
      constexpr int CPRA = BM / 4, CPRB = BN / 4;          // 16-byte chunks per reduction row of the image
      const int cpr = isA ? CPRA : CPRB;
      const int rr = q / cpr, cc = q % cpr;
      long long col = (isA ? m0 : n0) + 4 * cc;
      if (col > op.rows - 4) col = op.rows - 4;            // (rows % 4 == 0)
      src[i] = op.src + (kbeg + rr) * op.ld + col;
      adv[i] = (long long)RG_BK * op.ld;
    }
  }
  auto issue_part = [&](int slot, int part, int parts) {   // the wave's pieces [part * NPW / parts, (part + 1) * NPW / parts)
    float *base = lds + slot * ST_FL;
#pragma unroll
    for (int i = 0; i < NPW; ++i)
      if (i * parts / NPW == part) {
        rg_glds16(src[i], base + dst_off[i]);
        src[i] += adv[i];
      }
  };
  auto issue_stage = [&](int slot) { issue_part(slot, 0, 1); };

  // ---- prologue: the operand prologue's table, then three stages - all before the first wait
  if constexpr (AFFA) {
    // [a(chunk), b(chunk)] in 1 KB pieces (256 floats; lanes past the end repeat the last chunk); the waves split them
    const int per_half = tstride / 256;
    for (int p = wave; p < 2 * per_half; p += 4) {
      const int half = p >= per_half, pp = half ? p - per_half : p;
      long long k = kbeg + pp * 256 + lane * 4;
      if (k > g.red - 4) k = g.red - 4;
      rg_glds16(g.a.aff + (half ? g.red : 0) + k, tab + half * tstride + pp * 256);
    }
  }
  if constexpr (AFFB) {
    // [a(BN), b(BN)] of this tile's columns: BN / 4 chunks per half; wave 0 (a) and wave 1 (b), lanes < BN / 4
    if (wave < 2) {
      long long col = n0 + 4 * (lane % (BN / 4));
      if (col > g.b.rows - 4) col = g.b.rows - 4;
      rg_glds16(g.b.aff + (wave ? g.b.rows : 0) + col, tab + wave * 256);   // (lanes >= BN / 4 repeat: harmless)
    }
  }
  // vmcnt bookkeeping: every wave has issued the same number of table pieces only in the AFFB case of waves 0 / 1 and in
  // the AFFA case when 2 * per_half is a multiple of 4.  The first wait below therefore is vmcnt(0)-safe by construction:
  // the table is OLDER than every stage, so "all but the youngest N" covers it for any N <= what was issued after it.
  const int pre = T < RG_STAGES - 1 ? T : RG_STAGES - 1;
  for (int s = 0; s < pre; ++s) issue_stage(s);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- fragment addresses (bytes, relative to the start of a stage).  Every LDS read of the loop is an inline-asm
  // ds_read: a read the compiler can see makes it wait vmcnt(0) first (it cannot tell the slot being read from the slots
  // the DMA is still filling), which would drain the ring at every step.  Waits are therefore placed by hand.
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
  unsigned a_kc[KA == RK_KC ? MT : 1][4], b_kc[KB == RK_KC ? NT : 1][4];   // [block][cq]: the lane's chunk of group cq
  unsigned a_rc[KA == RK_RC ? MT : 1], b_rc[KB == RK_RC ? NT : 1];         // [block]: the lane's column, reduction row 0
  if constexpr (KA == RK_KC) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int r = wm * (BM / 2) + i * 32 + m;
#pragma unroll
      for (int cq = 0; cq < 4; ++cq) a_kc[i][cq] = (unsigned)((r * 8 + ((2 * cq + h) ^ ((r >> 1) & 7))) * 16);
    }
  } else {
#pragma unroll
    for (int i = 0; i < MT; ++i) a_rc[i] = (unsigned)((h * BM + wm * (BM / 2) + i * 32 + m) * 4);   // natural order: row 2s + h
  }
  if constexpr (KB == RK_KC) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int r = wn * (BN / 2) + j * 32 + m;
#pragma unroll
      for (int cq = 0; cq < 4; ++cq) b_kc[j][cq] = (unsigned)(A_FL * 4 + (r * 8 + ((2 * cq + h) ^ ((r >> 1) & 7))) * 16);
    }
  } else {
    // the reduction row this lane half contributes to slice (cq, e): 8 cq + 4 h + e beside a KC partner, 8 cq + 2 e + h else
#pragma unroll
    for (int j = 0; j < NT; ++j)
      b_rc[j] = (unsigned)(A_FL * 4 + ((KA == RK_KC ? 4 * h : h) * BN + wn * (BN / 2) + j * 32 + m) * 4);
  }
  constexpr unsigned RC_E = (KA == RK_KC ? 1 : 2);   // reduction rows between consecutive slices e of a group
  const unsigned tab0 = lds0 + (unsigned)(RG_STAGES * ST_FL * 4);

  float fa_b[NT], fb_b[NT];   // AFFB: this lane's column coefficients
  (void)fa_b; (void)fb_b;

  // one group = MFMA slices 4cq .. 4cq+3 of a step: MT (+2 table) + NT 16-byte reads, or 4 single reads per RC block
  struct Frag { f32x4 a[MT], b[NT], ta, tb; };
  unsigned a_rcs[KA == RK_RC ? MT : 1], b_rcs[KB == RK_RC ? NT : 1];   // a_rc / b_rc + this step's stage (set per step)
  (void)a_rcs; (void)b_rcs;
  auto read_group = [&](Frag &f, unsigned sbase, auto cq_c, int kk) {
    constexpr int cq = decltype(cq_c)::value;
    if constexpr (KA == RK_KC) {
#pragma unroll
      for (int i = 0; i < MT; ++i) f.a[i] = rg_ds128(sbase + a_kc[i][cq]);
      if constexpr (AFFA) {
        f.ta = rg_ds128(tab0 + (unsigned)((kk + 4 * (2 * cq + h)) * 4));
        f.tb = rg_ds128(tab0 + (unsigned)((tstride + kk + 4 * (2 * cq + h)) * 4));
      }
    } else {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        f.a[i][0] = rg_ds32o<(8 * cq + 0) * BM * 4>(a_rcs[i]);
        f.a[i][1] = rg_ds32o<(8 * cq + 2) * BM * 4>(a_rcs[i]);
        f.a[i][2] = rg_ds32o<(8 * cq + 4) * BM * 4>(a_rcs[i]);
        f.a[i][3] = rg_ds32o<(8 * cq + 6) * BM * 4>(a_rcs[i]);
      }
    }
    if constexpr (KB == RK_KC) {
#pragma unroll
      for (int j = 0; j < NT; ++j) f.b[j] = rg_ds128(sbase + b_kc[j][cq]);
    } else {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f.b[j][0] = rg_ds32o<(8 * cq + RC_E * 0) * BN * 4>(b_rcs[j]);
        f.b[j][1] = rg_ds32o<(8 * cq + RC_E * 1) * BN * 4>(b_rcs[j]);
        f.b[j][2] = rg_ds32o<(8 * cq + RC_E * 2) * BN * 4>(b_rcs[j]);
        f.b[j][3] = rg_ds32o<(8 * cq + RC_E * 3) * BN * 4>(b_rcs[j]);
      }
    }
  };
  using CQ0 = std::integral_constant<int, 0>;
  using CQ1 = std::integral_constant<int, 1>;
  using CQ2 = std::integral_constant<int, 2>;
  using CQ3 = std::integral_constant<int, 3>;
  constexpr int READS = (KA == RK_KC ? MT + (AFFA ? 2 : 0) : 4 * MT) + (KB == RK_KC ? NT : 4 * NT);   // per group
  auto prologue_group = [&](Frag &f) {
    if constexpr (AFFA) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = f.ta[e] * f.a[i][e] + f.tb[e];
          f.a[i][e] = z > 0.f ? z : 0.f;
        }
    }
    if constexpr (AFFB) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = fa_b[j] * f.b[j][e] + fb_b[j];
          f.b[j][e] = z > 0.f ? z : 0.f;
        }
    }
  };
  // BF: groups (lo, hi) = reduction sub-slots 0..3 and 4..7 of this lane half in one bf16 instruction (any assignment is
  // legal as long as both operands use it)
  auto mfma_pair_bf16 = [&](Frag &lo, Frag &hi, bool dma, int slot, int part) {
    prologue_group(lo);
    prologue_group(hi);
    bf16x8 a8[MT], b8[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { a8[i][e] = (__bf16)lo.a[i][e]; a8[i][4 + e] = (__bf16)hi.a[i][e]; }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) { b8[j][e] = (__bf16)lo.b[j][e]; b8[j][4 + e] = (__bf16)hi.b[j][e]; }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8[j], acc[i][j], 0, 0, 0);
    if (dma) {
      issue_part(slot, part, 4);
      issue_part(slot, part + 1, 4);
    }
  };
  auto mfma_group = [&](Frag &f, bool dma, int slot, int part) {
    if constexpr (AFFA) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = f.ta[e] * f.a[i][e] + f.tb[e];
          f.a[i][e] = z > 0.f ? z : 0.f;
        }
    }
    if constexpr (AFFB) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = fa_b[j] * f.b[j][e] + fb_b[j];
          f.b[j][e] = z > 0.f ? z : 0.f;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][e], f.b[j][e], acc[i][j], 0, 0, 0);
      // this group's share of the next stage's DMA goes out behind the first MFMAs: its issue slots (60 - 180 cycles per
      // piece) then lie in the shadow of the matrix pipe instead of in front of the step
      if (e == 0 && dma) issue_part(slot, part, 4);
    }
  };

  // RG_BNBWD: every y value and column coefficient the tile's epilogue needs is requested at the START OF THE LAST
  // reduction step (round 5; it used to be after the loop): the ring has no DMA in flight any more (the wait in front of
  // the last step is vmcnt(0)), so plain loads do not disturb the hand-counted waits, and the memory round trip for the
  // tile (1 - 2 us on a 20 - 40 us launch) runs under the step's 16 MT NT MFMAs instead of behind them
  float ea[EPI == RG_BNBWD ? NT : 1], eb[EPI == RG_BNBWD ? NT : 1], em[EPI == RG_BNBWD ? NT : 1], er[EPI == RG_BNBWD ? NT : 1];
  float yv[EPI == RG_BNBWD ? MT : 1][EPI == RG_BNBWD ? NT : 1][16];
  // (whole tiles: see the epilogue - one uniform tile pointer and a 32-bit lane offset instead of guarded 64-bit indices)
  const bool whole = m0 + BM <= g.a.rows && n0 + BN <= g.b.rows && g.ldd * (long long)BM < (1LL << 30);   // (wave-uniform)
  auto prefetch_epilogue = [&]() {
    if constexpr (EPI == RG_BNBWD) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const long long col = n0 + wn * (BN / 2) + j * 32 + m;
        const bool ok = col < g.b.rows;
        ea[j] = ok ? g.epi_ab[col] : 0.f;
        eb[j] = ok ? g.epi_ab[g.b.rows + col] : 0.f;
        em[j] = ok ? g.epi_ab[2 * g.b.rows + col] : 0.f;
        er[j] = ok ? g.epi_ab[3 * g.b.rows + col] : 0.f;
        if (whole) {
          const float *yp = g.epi_y + (m0 + wm * (BM / 2)) * g.ldd + n0 + wn * (BN / 2) + j * 32;   // uniform
          const unsigned pitch = (unsigned)g.ldd, lo = (unsigned)(4 * h) * pitch + (unsigned)m;
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) yv[i][j][r] = (yp + (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * pitch)[lo];
        } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const long long row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            yv[i][j][r] = (ok && row < g.a.rows) ? g.epi_y[row * g.ldd + col] : 0.f;
          }
        }
      }
    }
  };

  for (int step = 0; step < T; ++step) {
    // stages issued so far: min(T, step + 3); the ones after `step` may stay in flight
    const int ahead = (T - 1 - step) < (RG_STAGES - 2) ? (T - 1 - step) : (RG_STAGES - 2);
    if (ahead >= 2) rg_wait_vm<2 * NPW>();
    else if (ahead == 1) rg_wait_vm<NPW>();
    else rg_wait_vm<0>();
    __builtin_amdgcn_s_barrier();   // every wave's share of stage `step` has landed; all reads of stage step-1 retired
    __builtin_amdgcn_sched_barrier(0);
    const bool dma = step + RG_STAGES - 1 < T;
    const int nslot = (step + RG_STAGES - 1) % RG_STAGES;
    if constexpr (AFFB) {
      if (step == 0) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          fa_b[j] = rg_ds32(tab0 + (unsigned)((wn * (BN / 2) + j * 32 + m) * 4));
          fb_b[j] = rg_ds32(tab0 + (unsigned)((256 + wn * (BN / 2) + j * 32 + m) * 4));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (EPI == RG_BNBWD && step == T - 1) prefetch_epilogue();
    const unsigned sbase = lds0 + (unsigned)((step % RG_STAGES) * ST_FL * 4);
    const int kk = step * RG_BK;   // offset of this step inside the chunk (table index)
    if constexpr (KA == RK_RC) {
#pragma unroll
      for (int i = 0; i < MT; ++i) a_rcs[i] = sbase + a_rc[i];
    }
    if constexpr (KB == RK_RC) {
#pragma unroll
      for (int j = 0; j < NT; ++j) b_rcs[j] = sbase + b_rc[j];
    }
    // two fragment sets: the reads of group cq+1 are in flight under the MFMAs of group cq
    Frag f0, f1;
    if constexpr (BF) {
      Frag f2, f3;
      read_group(f0, sbase, CQ0{}, kk);
      read_group(f1, sbase, CQ1{}, kk);
      read_group(f2, sbase, CQ2{}, kk);
      read_group(f3, sbase, CQ3{}, kk);
      rg_wait_lgkm<2 * READS>();      // groups 0 and 1 have arrived (LDS operations retire in order)
      __builtin_amdgcn_sched_barrier(0);
      mfma_pair_bf16(f0, f1, dma, nslot, 0);
      rg_wait_lgkm<0>();              // every read of this stage has retired before the next barrier
      __builtin_amdgcn_sched_barrier(0);
      mfma_pair_bf16(f2, f3, dma, nslot, 2);
      continue;
    }
    read_group(f0, sbase, CQ0{}, kk);
    read_group(f1, sbase, CQ1{}, kk);
    rg_wait_lgkm<READS>();            // group 0 has arrived (LDS operations retire in order)
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(f0, dma, nslot, 0);
    read_group(f0, sbase, CQ2{}, kk);
    rg_wait_lgkm<READS>();            // group 1
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(f1, dma, nslot, 1);
    read_group(f1, sbase, CQ3{}, kk);
    rg_wait_lgkm<READS>();            // group 2
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(f0, dma, nslot, 2);
    rg_wait_lgkm<0>();                // group 3: every read of this stage has retired before the next barrier
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(f1, dma, nslot, 3);
  }

  // ---- epilogue: acc[i][j][r] = D[m0 + wm*BM/2 + i*32 + (r&3) + 8*(r>>2) + 4*h][n0 + wn*BN/2 + j*32 + m]
  float csum[NT], csq[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { csum[j] = 0.f; csq[j] = 0.f; }
  // Whole tiles (every tile of the step's shapes but a ragged last row / column of tiles) skip the per-element bounds
  // checks and the 64-bit index arithmetic: ONE wave-uniform tile pointer, one 32-bit lane offset, and row offsets that
  // are scalar multiples of the pitch.  The generic form below spent ~12 vector instructions per stored element on them -
  // two quarter-rate 32-bit multiplies and a 64-bit multiply-add among them - ~2000 cycles per 64 x 64 tile and four
  // times that per 128 x 128 tile, beside 8 - 32 reduction steps of ~2000.
  if (whole) {
    float *tp = dout + (m0 + wm * (BM / 2)) * g.ldd + n0 + wn * (BN / 2);   // uniform: this wave's quarter of the tile
    const unsigned pitch = (unsigned)g.ldd;
    const unsigned lo = (unsigned)(4 * h) * pitch + (unsigned)m;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[i][j][r];
          float *rp = tp + (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * pitch + j * 32;   // uniform
          if constexpr (EPI == RG_ATOMIC) atomicAdd(rp + lo, v);
          else rp[lo] = v;
          if constexpr (EPI == RG_STATS) {
            csum[j] += v;
            csq[j] += v * v;
          }
          if constexpr (EPI == RG_BNBWD) {
            const float y = yv[i][j][r];
            const float gg = (ea[j] * y + eb[j]) > 0.f ? v : 0.f;
            csum[j] += gg;
            csq[j] += gg * ((y - em[j]) * er[j]);
          }
        }
      }
  } else {
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const long long col = n0 + wn * (BN / 2) + j * 32 + m;
      const bool colok = col < g.b.rows;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = acc[i][j][r];
        if (row < g.a.rows && colok) {
          if constexpr (EPI == RG_ATOMIC) atomicAdd(dout + row * g.ldd + col, v);
          else dout[row * g.ldd + col] = v;
          if constexpr (EPI == RG_STATS) {
            csum[j] += v;
            csq[j] += v * v;
          }
          if constexpr (EPI == RG_BNBWD) {
            const float y = yv[i][j][r];
            const float gg = (ea[j] * y + eb[j]) > 0.f ? v : 0.f;
            csum[j] += gg;
            csq[j] += gg * ((y - em[j]) * er[j]);
          }
        }
      }
    }
  }
  if constexpr (EPI == RG_STATS || EPI == RG_BNBWD) {
    // column partials: lanes l and l+32 hold the same column; then the two row-waves (wm) through LDS (the ring is dead)
    __syncthreads();
    float *s_col = lds;   // [2][2][BN]
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      csum[j] += __shfl_xor(csum[j], 32);
      csq[j] += __shfl_xor(csq[j], 32);
      if (h == 0) {
        s_col[(0 * 2 + wm) * BN + wn * (BN / 2) + j * 32 + m] = csum[j];
        s_col[(1 * 2 + wm) * BN + wn * (BN / 2) + j * 32 + m] = csq[j];
      }
    }
    __syncthreads();
    if (t < BN) {
      const long long col = n0 + t;
      if (col < g.b.rows) {
        double *st = g.stats + (size_t)((bx / g.tiles_n) % g.stat_slots) * 2 * g.b.rows;
        atomicAdd(st + col, (double)s_col[(0 * 2 + 0) * BN + t] + (double)s_col[(0 * 2 + 1) * BN + t]);
        atomicAdd(st + g.b.rows + col, (double)s_col[(1 * 2 + 0) * BN + t] + (double)s_col[(1 * 2 + 1) * BN + t]);
      }
    }
  }
}

template <int KA, int KB, int BM, int BN, int EPI, bool AFFA, bool AFFB, bool BF = false>
__global__ __launch_bounds__(RG_TPB, (BM == 128 && BN == 128) ? 1 : 2) void gemm_ring_kernel(RingArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  ring_tile<KA, KB, BM, BN, EPI, AFFA, AFFB, BF>(g, blockIdx.x, blockIdx.y, lds);
}

// Round 5 (VERDICT round 4 #1 i): the dgrad and the wgrad of ONE layer in ONE launch.  Both read the same dY and depend
// on nothing else of each other; alone, each of the few-row products is a grid of ~256 workgroups of 64 x 64 tiles - half
// of the 512 such workgroups the chip holds - that is bound by what a workgroup waits for, not by the matrix pipes.  In
// one launch the two grids are resident TOGETHER (two workgroups per CU, one of each), and the launch takes about as long
// as the slower of the two.  Workgroups [0, nd) run the dgrad (tile bx = id % nd_x, chunk id / nd_x), the rest the wgrad.
template <int EPI_D, bool AFFB_W, bool BF>
__global__ __launch_bounds__(RG_TPB, 2) void gemm_ring_pair_kernel(RingArgs gd, RingArgs gw, unsigned nd_x, unsigned nd,
                                                                   unsigned nw_x) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const unsigned id = blockIdx.x;
  if (id < nd) ring_tile<RK_KC, RK_RC, 64, 64, EPI_D, false, false, BF>(gd, id % nd_x, id / nd_x, lds);
  else ring_tile<RK_RC, RK_RC, 64, 64, RG_ATOMIC, false, AFFB_W, BF>(gw, (id - nd) % nw_x, (id - nd) / nw_x, lds);
}

// Round 6 (VERDICT round 5 #1 i): the weight gradients of MANY layers in ONE launch.  A wgrad depends on nothing but the
// stored X and the dY its layer's backward has formed - it is not on the backward's dependency chain - yet the few-row
// ones (the 15 InvResMLP blocks' C -> 4C -> C pairs and aggregation convs on 1 024 .. 8 192 rows, the feature-propagation
// stacks and heads on 4 096 .. 32 768) were launched one layer at a time in the middle of it: 45 launches of 64 - 1 024
// workgroups living 10 - 30 us each, 5 - 7 us of which are fixed cost, at 0.14 - 0.42 of the matrix cores.  The caller
// now records them (fused_mlp.WgradQueue) and this kernel runs up to RG_GROUP_MAX of them as one grid: workgroup `id`
// finds its product in a table that travels BY VALUE in the kernel arguments (scalar loads; nothing to stage, and a
// captured graph replays it as is), then runs the ring kernel's 64 x 64 wgrad tile on one chunk of that product's
// reduction.  The host cuts every product's rows into chunks of about the same number of steps and deals the longest
// workgroups first, so the grid is a few thousand equal-sized pieces of work with no tail.
constexpr int RG_GROUP_MAX = 63;   // 63 x 64 bytes + the count = 4040: inside the 4 KB a kernel's arguments may take
struct RingGroupItem {
  const float *dy, *x, *aff;   // dY (P,N), X (P,K), optional [a(K), b(K)]: X is used as relu(a x + b)
  float *dw;                   // dW (N, ldw >= K) += dY^T f(X)  (fp32 atomics)
  int red;                     // P (<= 2^24: ring_group_suits)
  int N, K, ldw;
  int kchunk;                  // reduction rows per workgroup (multiple of 32)
  int tiles_n;                 // 64 x 64 output tiles per row of tiles
  unsigned first;              // first workgroup of this product
  unsigned tiles;              // its output tiles
};
static_assert(sizeof(RingGroupItem) == 64, "the argument table is sized by this");
struct RingGroup {
  RingGroupItem it[RG_GROUP_MAX];
  int count;
};

template <bool BF>
__global__ __launch_bounds__(RG_TPB, 2) void gemm_ring_group_kernel(RingGroup grp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const unsigned id = blockIdx.x;
  int i = 0;
  for (int j = 1; j < grp.count; ++j)   // (items ascend in `first`; <= 63 scalar compares)
    if (id >= grp.it[j].first) i = j;
  const RingGroupItem &e = grp.it[i];
  RingArgs g = {};
  g.a = {e.dy, e.N, e.N, nullptr};
  g.b = {e.x, e.K, e.K, e.aff};
  g.d = e.dw;
  g.ldd = e.ldw;
  g.red = e.red;
  g.kchunk = e.kchunk;
  g.stat_slots = 1;
  g.tiles_n = e.tiles_n;
  const unsigned local = id - e.first;
  const unsigned bx = local % e.tiles, by = local / e.tiles;
  if (e.aff) ring_tile<RK_RC, RK_RC, 64, 64, RG_ATOMIC, false, true, BF>(g, bx, by, lds);
  else ring_tile<RK_RC, RK_RC, 64, 64, RG_ATOMIC, false, false, BF>(g, bx, by, lds);
}

static inline bool rg_aligned16(const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

bool ring_group_suits(const float *dy, const float *x, const float *aff, const float *dw, long long P, int K, int N, int ldw) {
  return rg_aligned16(dy) && rg_aligned16(x) && (!aff || rg_aligned16(aff)) && dw && ldw >= K && P >= RG_BK &&
         P % RG_BK == 0 && K % 4 == 0 && N % 4 == 0 && K >= 4 && N >= 4 && P <= (1LL << 24);
}

// Launch the wgrads items[0 .. count) (every one ring_group_suits) in as few grids as the argument table allows.
void ring_group_launch(const RingWgrad *items, int count, hipStream_t s, bool bf16) {
  // cost of a workgroup of `steps` 32-row steps: ring_plan's constants for 64 x 64 tiles sharing a CU's matrix pipes
  const double FIXED = 5.0, STEP = 0.98, SLOTS = 512.0;
  for (int base = 0; base < count; base += RG_GROUP_MAX) {
    const int n = count - base < RG_GROUP_MAX ? count - base : RG_GROUP_MAX;
    const RingWgrad *it = items + base;
    // one step target for the whole grid: the largest chunk length whose modelled time is (nearly) the best one
    int best_target = 16;
    double best = 1e30;
    for (int target = 256; target >= 8; target /= 2) {
      double work = 0, bytes = 0;
      long long longest = 0;
      for (int i = 0; i < n; ++i) {
        const long long steps = it[i].P / RG_BK;
        const long long chunks = (steps + target - 1) / target, per = (steps + chunks - 1) / chunks;
        const long long tiles = (long long)((it[i].N + 63) / 64) * ((it[i].K + 63) / 64);
        work += (double)(tiles * chunks) * (FIXED + per * STEP);
        bytes += (double)(tiles * chunks) * 64 * 64 * 4;
        if (per > longest) longest = per;
      }
      const double t = work / SLOTS + bytes / 1.3e6 + 0.5 * (FIXED + longest * STEP);
      if (t < best * 0.97) { best = t; best_target = target; }
    }
    RingGroup grp = {};
    int order[RG_GROUP_MAX];
    long long per_of[RG_GROUP_MAX];
    for (int i = 0; i < n; ++i) {
      const long long steps = it[i].P / RG_BK, chunks = (steps + best_target - 1) / best_target;
      per_of[i] = (steps + chunks - 1) / chunks;
      order[i] = i;
    }
    for (int a = 1; a < n; ++a)   // longest workgroups first (insertion sort: n <= 63)
      for (int b = a; b > 0 && per_of[order[b]] > per_of[order[b - 1]]; --b) { const int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t; }
    unsigned first = 0;
    for (int k = 0; k < n; ++k) {
      const RingWgrad &w = it[order[k]];
      RingGroupItem &e = grp.it[k];
      const long long per = per_of[order[k]], chunks = (w.P / RG_BK + per - 1) / per;
      e.dy = w.dy; e.x = w.x; e.aff = w.aff; e.dw = w.dw;
      e.red = (int)w.P; e.N = w.N; e.K = w.K; e.ldw = w.ldw;
      e.kchunk = (int)(per * RG_BK);
      e.first = first;
      e.tiles_n = (w.K + 63) / 64;
      e.tiles = (unsigned)(((w.N + 63) / 64) * e.tiles_n);
      first += e.tiles * (unsigned)chunks;
    }
    grp.count = n;
    const size_t lds = (size_t)RG_STAGES * (64 + 64) * RG_BK * sizeof(float) + 2 * 256 * sizeof(float);
    static std::atomic<unsigned long long> attr_set[2] = {{0}, {0}};
    if (bf16) {
      allow_dynamic_lds(gemm_ring_group_kernel<true>, 160 * 1024, attr_set[1]);
      hipLaunchKernelGGL(gemm_ring_group_kernel<true>, dim3(first), dim3(RG_TPB), lds, s, grp);
    } else {
      allow_dynamic_lds(gemm_ring_group_kernel<false>, 160 * 1024, attr_set[0]);
      hipLaunchKernelGGL(gemm_ring_group_kernel<false>, dim3(first), dim3(RG_TPB), lds, s, grp);
    }
  }
}

template <int KA, int KB, int BM, int BN, int EPI, bool AFFA, bool AFFB, bool BF>
static void rg_launch(const RingArgs &g, long long tiles_m, unsigned chunks, hipStream_t s) {
  static std::atomic<unsigned long long> attr_set{0};
  auto kern = gemm_ring_kernel<KA, KB, BM, BN, EPI, AFFA, AFFB, BF>;
  allow_dynamic_lds(kern, 160 * 1024, attr_set);
  size_t lds = (size_t)RG_STAGES * (BM + BN) * RG_BK * sizeof(float);
  if (AFFA) lds += 2 * (size_t)((g.kchunk + 255) / 256 * 256) * sizeof(float);
  if (AFFB) lds += 2 * 256 * sizeof(float);
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * g.tiles_n), chunks), dim3(RG_TPB), lds, s, g);
}

template <int KA, int KB, int EPI, bool AFFA, bool AFFB>
static void rg_launch_tile(RingArgs &g, bool big, unsigned chunks, hipStream_t s, bool bf16) {
  if (big) {
    g.tiles_n = (int)((g.b.rows + 127) / 128);
    if (bf16) rg_launch<KA, KB, 128, 128, EPI, AFFA, AFFB, true>(g, (g.a.rows + 127) / 128, chunks, s);
    else rg_launch<KA, KB, 128, 128, EPI, AFFA, AFFB, false>(g, (g.a.rows + 127) / 128, chunks, s);
  } else {
    g.tiles_n = (int)((g.b.rows + 63) / 64);
    if (bf16) rg_launch<KA, KB, 64, 64, EPI, AFFA, AFFB, true>(g, (g.a.rows + 63) / 64, chunks, s);
    else rg_launch<KA, KB, 64, 64, EPI, AFFA, AFFB, false>(g, (g.a.rows + 63) / 64, chunks, s);
  }
}

// Tile and split of one product (M x N output, reduction `red`) by a small cost model (microseconds; constants from
// tools/ring_prof.sh on MI355X).  A workgroup costs a fixed ~5 us (launch, cold first loads, epilogue) + its reduction
// steps: 32 x 32 x 2 MFMAs of 64 cycles, 64 per wave and step for a 128 x 128 tile (one workgroup per CU: 136 KB of LDS),
// 16 for a 64 x 64 tile (two per CU, sharing the matrix pipes).  Splitting the reduction buys more workgroups at the price
// of a closing pass (forward / dgrad: partial products through the caller's workspace, `pass_bytes` per chunk and
// `pass_us` for the extra launch) or of more atomics (wgrad: 1.3 TB/s of float atomics).
void ring_plan(long long M, long long N, long long red, bool want_split, long long max_chunks, RingPlan *p, int atomics,
               bool bf16) {
  // (bf16: 2 matrix instructions per step and tile pair instead of 16, yet the same plans are the best ones - scaling the
  // step costs by 0.5 / 0.25 for bf16 moved configs[4] by nothing, 23.86 - 23.94 ms: the few-row products are bound by
  // what a workgroup waits for, not by its matrix instructions)
  (void)bf16;
  const double FIXED = 5.0, STEP_BIG = 1.95, STEP_SMALL = 0.49, CUS = 256.0;
  double best = 1e30;
  p->big = false; p->chunks = 1; p->kchunk = red;
  for (int big = 0; big < 2; ++big) {
    const long long T0 = big ? 128 : 64;
    if (big && (M < 128 || N < 128)) continue;
    const long long tiles = ((M + T0 - 1) / T0) * ((N + T0 - 1) / T0);
    for (long long chunks = 1; chunks <= 64; chunks *= 2) {
      if (chunks > 1 && (!want_split || chunks > max_chunks)) break;
      long long kc = (red + chunks - 1) / chunks;
      kc = (kc + 31) / 32 * 32;
      if (chunks > 1 && kc < 128) break;
      const long long nch = (red + kc - 1) / kc;
      const double steps = (double)(kc / 32);
      const double wgs = (double)(tiles * nch);
      double t;
      if (big) t = ceil(wgs / CUS) * (FIXED + steps * STEP_BIG);
      else t = ceil(wgs / (2 * CUS)) * (FIXED + steps * STEP_SMALL * (wgs > CUS ? 2.0 : 1.0));
      if (nch > 1) {
        const double bytes = (double)M * N * 4.0;
        t += atomics ? nch * bytes / 1.3e6 : 4.0 + (nch + 1) * bytes / 3.0e6;   // bytes / (TB/s) = 1e-6 us
      } else if (atomics) {
        t += (double)M * N * 4.0 / 1.3e6;
      }
      if (t < best) { best = t; p->big = big != 0; p->chunks = (int)nch; p->kchunk = kc; }
    }
  }
}

bool ring_gemm_try(int kind, const float *a, const float *b, const float *aff, float *d, long long P, int K, int N,
                   double *stats, int stat_slots, const float *epi_y, const float *epi_ab, const RingPlan &plan,
                   long long dchunk, hipStream_t s, bool bf16) {
  RingArgs g = {};
  g.d = d;
  g.stats = stats;
  g.stat_slots = stat_slots < 1 ? 1 : stat_slots;
  g.epi_y = epi_y;
  g.epi_ab = epi_ab;
  g.kchunk = plan.kchunk;
  g.dchunk = dchunk;
  const unsigned chunks = (unsigned)plan.chunks;
  if (!rg_aligned16(a) || !rg_aligned16(b) || !rg_aligned16(d) || (aff && !rg_aligned16(aff))) return false;
  if (plan.kchunk % RG_BK != 0 || plan.kchunk < RG_BK || chunks < 1 || chunks > 65535) return false;
  if (kind == RING_FWD) {
    // Y (P,N) = f(X (P,K)) W (N,K)^T: both reduction-contiguous
    if (K % RG_BK != 0 || K % 4 != 0 || P < 1 || N < 1) return false;
    if (aff && 2 * (size_t)plan.kchunk * sizeof(float) > 16 * 1024) return false;   // the table's share of the LDS
    g.a = {a, P, K, aff};
    g.b = {b, N, K, nullptr};
    g.ldd = N;
    g.red = K;
    if (chunks > 1 && (stats || !dchunk)) return false;   // partial products carry no statistics
#define GB_RF(EPI_)                                                                                   \
    do {                                                                                              \
      if (aff) rg_launch_tile<RK_KC, RK_KC, EPI_, true, false>(g, plan.big, chunks, s, bf16);               \
      else rg_launch_tile<RK_KC, RK_KC, EPI_, false, false>(g, plan.big, chunks, s, bf16);                  \
    } while (0)
    if (stats) GB_RF(RG_STATS); else GB_RF(RG_STORE);
#undef GB_RF
    return true;
  }
  if (kind == RING_DGRAD) {
    // dX (P,K) = dY (P,N) W (N,K): A reduction-contiguous, B row-contiguous (element (k, n) at w[n K + k])
    if (N % RG_BK != 0 || K % 4 != 0 || N % 4 != 0 || P < 1 || aff) return false;
    g.a = {a, P, N, nullptr};
    g.b = {b, K, K, nullptr};
    g.ldd = K;
    g.red = N;
    if (chunks > 1 && (stats || !dchunk)) return false;
    if (stats) rg_launch_tile<RK_KC, RK_RC, RG_BNBWD, false, false>(g, plan.big, chunks, s, bf16);
    else rg_launch_tile<RK_KC, RK_RC, RG_STORE, false, false>(g, plan.big, chunks, s, bf16);
    return true;
  }
  if (kind == RING_WGRAD) {
    // dW (N,K) += dY (P,N)^T f(X (P,K)): both row-contiguous, reduction over the P rows, fp32 atomics into dW
    if (P % RG_BK != 0 || K % 4 != 0 || N % 4 != 0 || K < 4 || N < 4 || stats) return false;
    g.a = {a, N, N, nullptr};
    g.b = {b, K, K, aff};
    g.ldd = K;
    g.red = P;
    if (aff) rg_launch_tile<RK_RC, RK_RC, RG_ATOMIC, false, true>(g, plan.big, chunks, s, bf16);
    else rg_launch_tile<RK_RC, RK_RC, RG_ATOMIC, false, false>(g, plan.big, chunks, s, bf16);
    return true;
  }
  return false;
}

// dgrad (dX (P,K) = dY (P,N) W (N,K), optional BatchNorm-backward sums of the previous layer) and wgrad
// (dW (N,K) += dY^T f(X (P,K))) of one layer as ONE launch of 64 x 64-tile workgroups (gemm_ring_pair_kernel).  The dgrad is
// not split (its reduction N is short where this pays); the wgrad splits the P reduction as ring_plan says.  Returns false
// (nothing launched) when either product does not suit the ring kernel in this form.
bool ring_pair_try(const float *dy, const float *w, float *dx, double *dstats, int stat_slots, const float *y_prev,
                   const float *ab_prev, const float *x, const float *x_aff, float *dw, long long P, int K, int N,
                   hipStream_t s, bool bf16) {
  if (!rg_aligned16(dy) || !rg_aligned16(w) || !rg_aligned16(dx) || !rg_aligned16(x) || !rg_aligned16(dw) ||
      (x_aff && !rg_aligned16(x_aff)))
    return false;
  if (N % RG_BK != 0 || P % RG_BK != 0 || K % 4 != 0 || N % 4 != 0 || K < 4 || N < 4 || P < 64) return false;
  RingPlan pw;
  ring_plan(N, K, P, true, 65535, &pw, 1, bf16);
  if (pw.big || pw.kchunk % RG_BK != 0 || pw.chunks < 1) return false;
  RingArgs gd = {}, gw = {};
  gd.a = {dy, P, N, nullptr};
  gd.b = {w, K, K, nullptr};
  gd.d = dx; gd.ldd = K; gd.red = N; gd.kchunk = N; gd.dchunk = 0;
  gd.stats = dstats; gd.stat_slots = stat_slots < 1 ? 1 : stat_slots; gd.epi_y = y_prev; gd.epi_ab = ab_prev;
  gd.tiles_n = (int)((K + 63) / 64);
  gw.a = {dy, N, N, nullptr};
  gw.b = {x, K, K, x_aff};
  gw.d = dw; gw.ldd = K; gw.red = P; gw.kchunk = pw.kchunk; gw.dchunk = 0; gw.stat_slots = 1;
  gw.tiles_n = (int)((K + 63) / 64);
  const long long nd_x = ((P + 63) / 64) * gd.tiles_n, nw_x = ((long long)(N + 63) / 64) * gw.tiles_n;
  const long long total = nd_x + nw_x * pw.chunks;
  if (total > 0x7fffffffLL || nd_x > 0x7fffffffLL) return false;
  size_t lds = (size_t)RG_STAGES * (64 + 64) * RG_BK * sizeof(float) + 2 * 256 * sizeof(float);   // (+ the wgrad's table)
#define GB_RP(EPI_D_, AFF_, BF_)                                                                                \
  do {                                                                                                          \
    static std::atomic<unsigned long long> attr_set{0};                                                         \
    auto kern = gemm_ring_pair_kernel<EPI_D_, AFF_, BF_>;                                                       \
    allow_dynamic_lds(kern, 160 * 1024, attr_set);                                                              \
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(RG_TPB), lds, s, gd, gw, (unsigned)nd_x, (unsigned)nd_x,  \
                       (unsigned)nw_x);                                                                        \
  } while (0)
#define GB_RP2(EPI_D_, AFF_)            \
  do {                                  \
    if (bf16) GB_RP(EPI_D_, AFF_, true); \
    else GB_RP(EPI_D_, AFF_, false);    \
  } while (0)
  if (dstats) {
    if (x_aff) GB_RP2(RG_BNBWD, true); else GB_RP2(RG_BNBWD, false);
  } else {
    if (x_aff) GB_RP2(RG_STORE, true); else GB_RP2(RG_STORE, false);
  }
#undef GB_RP2
#undef GB_RP
  return true;
}

}  // namespace gb

// fp32 MFMA GEMMs for the channel-last SharedMLP path (gfx950, v_mfma_f32_32x32x2_f32: exact fp32,
// bit-for-bit a k-ordered fmaf chain, 64 FLOP/clk/SIMD).  Replaces the cuBLAS/cuDNN 1x1-convolution
// calls the reference makes through torch (pytorch_utils.py:61-113) for
//
//   forward   Y[p,n]  = sum_k f(X[p,k]) W[n,k]     f = identity | relu(a_k x + b_k)  (previous layer's
//                                                   BatchNorm+ReLU applied while loading, so the
//                                                   normalised activation is never materialised)
//             + epilogue: per-column sum / sum of squares of Y (the BatchNorm batch statistics) as
//               fp64 atomics, one per column per workgroup
//   dgrad     dX[p,k] = sum_n dY[p,n] Wt[k,n]      (Wt = W^T, K x N row-major, made by the caller)
//   wgrad     dW[n,k] += sum_p dY[p,n] X[p,k]      reduction over the P rows split across workgroups,
//                                                   fp32 atomics into dW (N x K, tiny)
//
// Tiling: 128 x 128 output tile per 256-thread workgroup (4 waves as 2 x 2, 64 x 64 each = 2 x 2 MFMA
// tiles, 64 accumulator VGPRs), reduction step 16.  Both operand tiles live in LDS as [row][17] (pitch
// 17 floats: the 32 lanes of a half-wave read one column of 32 rows -> 32 distinct banks), double
// buffered, filled through registers one step ahead of the MFMAs.  The contraction is compute-bound
// (16 KB of operands per 4.2 MFLOP), so plain ds_read_b32 operand fetches are off the critical path.
#include <stdlib.h>


#include "gb_common.h"
#include "gemm_rs.h"
#include "gemm_ring.h"
#include "gemm_wg.h"

namespace gb {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GK = 16, GPITCH = GK + 1;  // block tile BM x BN (64 or 128 each) is a template parameter
constexpr int GTPB = 256;

// how an operand tile element (row r of the tile, reduction index k) is fetched from global memory
enum { OP_KC = 0,     // src[(row0 + r) * ld + k]      (reduction index contiguous)
       OP_RC = 1 };   // src[k * ld + (row0 + r)]      (tile-row index contiguous: transposed read)

struct Operand {
  const float *src;
  long long rows;   // valid tile-row indices  [0, rows)
  long long red;    // valid reduction indices [0, red)
  long long ld;
  const float *aff; // optional [a(red), b(red)]: value = relu(a_k * x + b_k)   (OP_KC only)
  const float *gen_x; // optional (OP_RC only): src is not read - element (tile row r, reduction index k) = gen_x[k] . gen_w[r],
  const float *gen_w; //   gen_x (red,3), gen_w (rows,3): the output of a 3-input first layer that was never stored (then aff)
};

// registers holding one thread's share (ROWS/16 floats... i.e. 4 or 8) of a ROWS x 16 operand tile
struct Frag { float v[8]; float ca[4], cb[4]; bool ok[2]; };  // data, affine (a,b) of its 4 channels, validity

template <int KIND, bool VEC, int ROWS, bool GEN = false>
__device__ __forceinline__ void load_frag(const Operand &op, long long row0, long long k0, Frag &f) {
  const int t = threadIdx.x;
  constexpr int TPK = ROWS / 4;        // OP_RC: threads covering the tile rows of one reduction index
  constexpr int KPP = GTPB / TPK;      // OP_RC: reduction indices per pass (8 or 16)
#pragma unroll
  for (int h = 0; h < ROWS / 64; ++h) {
    if constexpr (KIND == OP_KC) {
      const int r = (t >> 2) + 64 * h;          // 4 threads cover the 16 reduction indices of a row
      const long long row = row0 + r;
      const long long k = k0 + 4 * (t & 3);
      const float *p = op.src + row * op.ld + k;
      const bool rok = row < op.rows;
      f.ok[h] = rok;
      if (VEC && rok && k + 3 < op.red) {
        const float4 q = *reinterpret_cast<const float4 *>(p);
        f.v[4 * h + 0] = q.x; f.v[4 * h + 1] = q.y; f.v[4 * h + 2] = q.z; f.v[4 * h + 3] = q.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) f.v[4 * h + e] = (rok && k + e < op.red) ? p[e] : 0.f;
      }
      if (op.aff && h == 0) {  // channel = reduction index; the same 4 channels for every h
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = k + e < op.red;
          f.ca[e] = ok ? op.aff[k + e] : 0.f;
          f.cb[e] = ok ? op.aff[op.red + k + e] : 0.f;
        }
      }
    } else {
      const long long k = k0 + (t / TPK) + KPP * h;
      const long long row = row0 + 4 * (t % TPK);
      const float *p = op.src + k * op.ld + row;
      const bool kok = k < op.red;
      f.ok[h] = kok;
      if constexpr (GEN) {   // a template switch: the plain loaders must not carry this branch (it cost every wgrad 15-35 %)
        const float gx = kok ? op.gen_x[k * 3] : 0.f, gy = kok ? op.gen_x[k * 3 + 1] : 0.f, gz = kok ? op.gen_x[k * 3 + 2] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool rok = row + e < op.rows;
          const float *wr = op.gen_w + (rok ? (row + e) * 3 : 0);
          f.v[4 * h + e] = rok ? ((gx * wr[0]) + (gy * wr[1])) + (gz * wr[2]) : 0.f;  // == gemm_rs.hip's lin3
        }
      } else
      if (VEC && kok && row + 3 < op.rows) {
        const float4 q = *reinterpret_cast<const float4 *>(p);
        f.v[4 * h + 0] = q.x; f.v[4 * h + 1] = q.y; f.v[4 * h + 2] = q.z; f.v[4 * h + 3] = q.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) f.v[4 * h + e] = (kok && row + e < op.rows) ? p[e] : 0.f;
      }
      if (op.aff && h == 0) {  // channel = tile-row index here: aff = [a(rows), b(rows)]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = row + e < op.rows;
          f.ca[e] = ok ? op.aff[row + e] : 0.f;
          f.cb[e] = ok ? op.aff[op.rows + row + e] : 0.f;
        }
      }
    }
  }
}

// relu(a*x + b) on a fetched fragment.  Called AFTER the MFMA block of the step (right before the LDS
// store), so the global loads issued by load_frag stay in flight behind the MFMAs; padding elements have
// a = b = 0 and stay exact zeros.
template <int ROWS>
__device__ __forceinline__ void apply_aff(Frag &f) {
#pragma unroll
  for (int h = 0; h < ROWS / 64; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float z = f.ca[e] * f.v[4 * h + e] + f.cb[e];
      f.v[4 * h + e] = (f.ok[h] && z > 0.f) ? z : 0.f;
    }
}
template <int KIND, int ROWS>
__device__ __forceinline__ void store_frag(float *lds, const Frag &f) {
  const int t = threadIdx.x;
  constexpr int TPK = ROWS / 4;
  constexpr int KPP = GTPB / TPK;
#pragma unroll
  for (int h = 0; h < ROWS / 64; ++h) {
    if constexpr (KIND == OP_KC) {
      float *d = lds + ((t >> 2) + 64 * h) * GPITCH + 4 * (t & 3);
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = f.v[4 * h + e];
    } else {
      // row-contiguous operands keep the global order in LDS: [k][ROWS + 4], one 16-byte store
      float *d = lds + ((t / TPK) + KPP * h) * (ROWS + 4) + 4 * (t % TPK);
      *reinterpret_cast<float4 *>(d) = make_float4(f.v[4 * h + 0], f.v[4 * h + 1], f.v[4 * h + 2], f.v[4 * h + 3]);
    }
  }
}

enum { EPI_STORE = 0,
       EPI_STORE_STATS = 1,   // + column sums of D and D^2                     (BatchNorm batch statistics)
       EPI_ATOMIC = 2,        // D accumulated with fp32 atomics                 (split-K wgrad)
       EPI_STORE_BNBWD = 3 }; // + column sums of g and g*xhat, g = D*[a*y+b>0]  (BatchNorm-backward statistics
                              //   of the layer whose pre-BN output y has D's shape: dgrad of the next layer)

// D[i,j] = sum_k A[i,k] B[j,k] over k in [kbeg, kend);  D is (a.rows x b.rows) with leading dim ldd
// 128 x 128 tiles of the store / atomic kernels need 132-136 VGPRs as written: asking for four waves per SIMD
// (<= 128 registers) buys a fourth resident workgroup per CU
// XOP: 0 plain operands; XOP_GENB: B is generated (gb_gemm_wgrad_gen3).  Compile-time, so that the plain instantiations do
// not carry it: as a run-time branch in the loaders it cost every split-K product 15-35 % (registers: the 128 x 128 tile
// spilled under its 128-VGPR cap).
enum { XOP_NONE = 0, XOP_GENB = 1 };
template <int KA, int KB, bool VA, bool VB, int EPI, int GM, int GN, bool BF = false, int XOP = XOP_NONE>
__global__ __launch_bounds__(GTPB, ((GM == 128 && GN == 128 && (EPI == 0 || EPI == 2) && !BF) ? 4 : 1)) void gemm_cl_kernel(Operand a, Operand b, float *__restrict__ d, long long ldd,
                                                        double *__restrict__ stats, long long kchunk,
                                                        int tiles_n, int stat_slots,
                                                        const float *__restrict__ epi_y,
                                                        const float *__restrict__ epi_ab,
                                                        const uint16_t *__restrict__ epi_w16, long long dchunk,
                                                        const long long *__restrict__ red_dev) {
  // red_dev (split-K products over rows, EPI_ATOMIC): the reduction length is the caller's device-side row count (<= the
  // host-side capacity a.red the grid was sized for); the gridDim.y chunks re-divide it among themselves
  if constexpr (EPI == EPI_ATOMIC) {
    if (red_dev) {
      long long pd = *red_dev;
      pd = pd < a.red ? (pd > 0 ? pd : 0) : a.red;
      a.red = pd;
      b.red = pd;
      kchunk = ((pd + gridDim.y - 1) / gridDim.y + GK - 1) / GK * GK;
      if (kchunk < 256) kchunk = 256;
      if ((long long)blockIdx.y * kchunk >= pd) return;
    }
  }
  // dchunk != 0: reduction chunk blockIdx.y stores its partial product to its own copy of D (split_reduce_kernel sums
  // the copies in chunk order: same bits on every run, which accumulating with atomics does not give)
  d += (long long)blockIdx.y * dchunk;
  // LDS image per operand kind: OP_KC [row][17] (element (r,k) at r*17 + k), OP_RC [k][rows+4]
  constexpr int A_RS = KA == OP_KC ? GPITCH : 1, A_KS = KA == OP_KC ? 1 : GM + 4;  // row / k strides
  constexpr int B_RS = KB == OP_KC ? GPITCH : 1, B_KS = KB == OP_KC ? 1 : GN + 4;
  constexpr int A_SZ = KA == OP_KC ? GM * GPITCH : GK * (GM + 4);
  constexpr int B_SZ = KB == OP_KC ? GN * GPITCH : GK * (GN + 4);
  __shared__ __attribute__((aligned(16))) float lds_a[2][A_SZ];
  __shared__ __attribute__((aligned(16))) float lds_b[2][B_SZ];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long long m0 = (long long)(blockIdx.x / tiles_n) * GM;
  const long long n0 = (long long)(blockIdx.x % tiles_n) * GN;
  const long long kbeg = (long long)blockIdx.y * kchunk;
  long long kend = kbeg + kchunk;
  if (kend > a.red) kend = a.red;

  constexpr int MT = GM / 64, NT = GN / 64;  // 32x32 MFMA tiles per wave (waves are 2 x 2)
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Frag fa, fb;
  load_frag<KA, VA, GM>(a, m0, kbeg, fa);
  load_frag<KB, VB, GN, XOP == XOP_GENB>(b, n0, kbeg, fb);
  if (a.aff) apply_aff<GM>(fa);
  if (b.aff) apply_aff<GN>(fb);
  store_frag<KA, GM>(lds_a[0], fa);
  store_frag<KB, GN>(lds_b[0], fb);
  __syncthreads();
  int buf = 0;
  for (long long k0 = kbeg; k0 < kend; k0 += GK) {
    const bool more = k0 + GK < kend;
    if (more) {
      load_frag<KA, VA, GM>(a, m0, k0 + GK, fa);
      load_frag<KB, VB, GN, XOP == XOP_GENB>(b, n0, k0 + GK, fb);
    }
    if constexpr (BF) {
      // bf16 matrix cores: ONE v_mfma_f32_32x32x16_bf16 per 32x32 tile and step; a lane supplies the 8 reduction
      // indices 8*(lane>>5) .. +7 of its row, read as fp32 from the same LDS image and rounded here
      const float *pa8 = lds_a[buf] + (wm * (GM / 2) + (lane & 31)) * A_RS + 8 * (lane >> 5) * A_KS;
      const float *pb8 = lds_b[buf] + (wn * (GN / 2) + (lane & 31)) * B_RS + 8 * (lane >> 5) * B_KS;
      bf16x8 a8[MT], b8[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) a8[i][e] = (__bf16)pa8[i * 32 * A_RS + e * A_KS];
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) b8[j][e] = (__bf16)pb8[j * 32 * B_RS + e * B_KS];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8[j], acc[i][j], 0, 0, 0);
    } else {
    const float *pa = lds_a[buf] + (wm * (GM / 2) + (lane & 31)) * A_RS + (lane >> 5) * A_KS;
    const float *pb = lds_b[buf] + (wn * (GN / 2) + (lane & 31)) * B_RS + (lane >> 5) * B_KS;
#pragma unroll
    for (int s = 0; s < GK / 2; ++s) {
      float av[MT], bv[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) av[i] = pa[i * 32 * A_RS + 2 * s * A_KS];
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[j] = pb[j * 32 * B_RS + 2 * s * B_KS];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    }
    if (more) {
      if (a.aff) apply_aff<GM>(fa);
      if (b.aff) apply_aff<GN>(fb);
      store_frag<KA, GM>(lds_a[buf ^ 1], fa);
      store_frag<KB, GN>(lds_b[buf ^ 1], fb);
    }
    __syncthreads();
    buf ^= 1;
  }

  // epilogue: acc[mt][nt][reg] is D[m0 + wm*64 + mt*32 + (reg&3) + 8*(reg>>2) + 4*(lane>>5)][n0 + wn*64 + nt*32 + (lane&31)]
  const bool whole = m0 + GM <= a.rows && n0 + GN <= b.rows && (long long)ldd * GM < (1LL << 30);   // (wave-uniform)
  float csum[NT], csq[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { csum[j] = 0.f; csq[j] = 0.f; }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const long long col = n0 + wn * (GN / 2) + nt * 32 + (lane & 31);
      float ea = 0.f, eb = 0.f, emean = 0.f, erstd = 0.f;
      if constexpr (EPI == EPI_STORE_BNBWD) {
        if (col < b.rows) {
          ea = epi_ab[col]; eb = epi_ab[b.rows + col]; emean = epi_ab[2 * b.rows + col]; erstd = epi_ab[3 * b.rows + col];
        }
      }
      if (whole) {
        // (a whole tile: one wave-uniform pointer per row, a 32-bit lane offset - the generic form below spends ~12 vector
        // instructions per element on bounds checks and 64-bit index arithmetic: csrc/gemm_ring.hip has the account)
        const long long r0 = m0 + wm * (GM / 2) + mt * 32, c0 = n0 + wn * (GN / 2) + nt * 32;
        float *tp = d + r0 * ldd + c0;                                     // uniform
        const float *yp = EPI == EPI_STORE_BNBWD ? epi_y + r0 * ldd + c0 : nullptr;
        const unsigned pitch = (unsigned)ldd, lo = (unsigned)(4 * (lane >> 5)) * pitch + (unsigned)(lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[mt][nt][r];
          const unsigned ro = (unsigned)((r & 3) + 8 * (r >> 2)) * pitch;   // uniform
          if constexpr (EPI == EPI_ATOMIC) atomicAdd(tp + ro + lo, v);
          else (tp + ro)[lo] = v;
          if constexpr (EPI == EPI_STORE_BNBWD) {
            const float y = (yp + ro)[lo];
            const float g = (ea * y + eb) > 0.f ? v : 0.f;
            csum[nt] += g;
            csq[nt] += g * ((y - emean) * erstd);
          }
          if constexpr (EPI == EPI_STORE_STATS) {
            if (epi_w16) {
              const float wv = (float)epi_w16[r0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] * v;
              csum[nt] += wv;
              csq[nt] += wv * v;
            } else {
              csum[nt] += v;
              csq[nt] += v * v;
            }
          }
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long row = m0 + wm * (GM / 2) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float v = acc[mt][nt][r];
        if (row < a.rows && col < b.rows) {
          if constexpr (EPI == EPI_ATOMIC) atomicAdd(d + row * ldd + col, v);
          else d[row * ldd + col] = v;
          if constexpr (EPI == EPI_STORE_BNBWD) {
            const float y = epi_y[row * ldd + col];
            const float g = (ea * y + eb) > 0.f ? v : 0.f;
            csum[nt] += g;
            csq[nt] += g * ((y - emean) * erstd);
          }
        }
        if constexpr (EPI == EPI_STORE_STATS) {  // padded rows/cols are exact zeros
          if (epi_w16) {  // row multiplicities (de-duplicated rows): sums over the original batch
            const float wv = (row < a.rows ? (float)epi_w16[row] : 0.f) * v;
            csum[nt] += wv;
            csq[nt] += wv * v;
          } else {
            csum[nt] += v;
            csq[nt] += v * v;
          }
        }
      }
    }
  if constexpr (EPI == EPI_STORE_STATS || EPI == EPI_STORE_BNBWD) {
    // column partials: lanes l and l+32 hold the same column; then the two M-waves (wm) via LDS
    __shared__ float s_col[2][2][GN];  // [sum|sq][wm][col in tile]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      csum[nt] += __shfl_xor(csum[nt], 32);
      csq[nt] += __shfl_xor(csq[nt], 32);
      if (lane < 32) {
        s_col[0][wm][wn * (GN / 2) + nt * 32 + lane] = csum[nt];
        s_col[1][wm][wn * (GN / 2) + nt * 32 + lane] = csq[nt];
      }
    }
    __syncthreads();
    if (threadIdx.x < GN) {
      const long long col = n0 + threadIdx.x;
      if (col < b.rows) {
        // stats is [stat_slots][2N]: row tiles spread over the slots so that thousands of workgroups
        // do not serialise on the same 2N addresses (same-address atomics run ~14x slower on MI355X)
        double *st = stats + (size_t)((blockIdx.x / tiles_n) % stat_slots) * 2 * b.rows;
        atomicAdd(st + col, (double)s_col[0][0][threadIdx.x] + (double)s_col[0][1][threadIdx.x]);
        atomicAdd(st + b.rows + col, (double)s_col[1][0][threadIdx.x] + (double)s_col[1][1][threadIdx.x]);
      }
    }
  }
}

// wgrad for a handful of input channels (K <= 4: the xyz-only first layer of a grouped SharedMLP): dW[n, j] =
// sum_p dY[p, n] x[p, j] is a weighted column reduction of dY, HBM-bound on reading dY once - the MFMA tiling
// above would waste 60/64 of its reduction lanes on it.  A thread owns 4 columns n and all K inputs, 4 rows in
// flight; row lanes of a workgroup are combined through LDS, one fp32 atomic per (n, j) per workgroup.
constexpr int WSK_ROWS = 2048;  // rows per workgroup: few workgroups, so few same-address atomics per dW element
template <int K>
__global__ __launch_bounds__(GTPB) void wgrad_smallk_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                            float *__restrict__ dw, long long P, int N) {
  __shared__ float part[GTPB * 4 * K];
  const int tpr = N / 4, rpp = GTPB / tpr;
  const int cg = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  const long long r0 = (long long)blockIdx.x * WSK_ROWS;
  long long r1 = r0 + WSK_ROWS;
  if (r1 > P) r1 = P;
  float acc[4][K];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int j = 0; j < K; ++j) acc[t][j] = 0.f;
  if (rl < rpp)
    for (long long r = r0 + rl; r < r1; r += 4 * rpp) {
      float4 g[4];
      float xv[4][K];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long rr = r + (long long)u * rpp;
        if (rr < r1) {
          g[u] = *reinterpret_cast<const float4 *>(dy + rr * N + 4 * cg);
#pragma unroll
          for (int j = 0; j < K; ++j) xv[u][j] = x[rr * K + j];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r + (long long)u * rpp < r1) {
          const float gv[4] = {g[u].x, g[u].y, g[u].z, g[u].w};
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < K; ++j) acc[t][j] += gv[t] * xv[u][j];
        }
    }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int j = 0; j < K; ++j) part[(threadIdx.x * 4 + t) * K + j] = acc[t][j];
  __syncthreads();
  // element o = (cg*4 + t)*K + j  of row lane l lives at part[(l*tpr*4)*K + o]
  for (int o = threadIdx.x; o < N * K; o += GTPB) {
    float s = 0.f;
    for (int l = 0; l < rpp; ++l) s += part[l * tpr * 4 * K + o];
    atomicAdd(dw + o, s);  // dw is (N, K) row-major: o = n*K + j
  }
}

static inline bool aligned16(const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

template <int KA, int KB, int EPI, int BM, int BN, int XOP = XOP_NONE>
static void launch_tile(const Operand &a, const Operand &b, bool va, bool vb, float *d, long long ldd, double *stats,
                        long long kchunk, unsigned chunks, hipStream_t s, bool bf16, int stat_slots, const float *epi_y,
                        const float *epi_ab, const uint16_t *epi_w16, long long dchunk, const long long *red_dev) {
  const int tiles_n = (int)((b.rows + BN - 1) / BN);
  const long long tiles_m = (a.rows + BM - 1) / BM;
  const dim3 grid((unsigned)(tiles_m * tiles_n), chunks);
  const bool bf = bf16 && a.red >= 16;  // GB_PREC_BF16; short reductions (xyz-only first layers) stay fp32
#define GB_L(VA_, VB_, BF_)                                                                                       \
  hipLaunchKernelGGL((gemm_cl_kernel<KA, KB, VA_, VB_, EPI, BM, BN, BF_, XOP>), grid, dim3(GTPB), 0, s, a, b, d, ldd, \
                     stats, kchunk, tiles_n, stat_slots, epi_y, epi_ab, epi_w16, dchunk, red_dev)
  if constexpr (XOP == XOP_GENB) {  // B is generated (gb_gemm_wgrad_gen3): never a vector load of it
    if (bf) { if (va) GB_L(true, false, true); else GB_L(false, false, true); }
    else { if (va) GB_L(true, false, false); else GB_L(false, false, false); }
  } else
  if (bf) {
    if (va && vb) GB_L(true, true, true);
    else if (va) GB_L(true, false, true);
    else if (vb) GB_L(false, true, true);
    else GB_L(false, false, true);
  } else {
    if (va && vb) GB_L(true, true, false);
    else if (va) GB_L(true, false, false);
    else if (vb) GB_L(false, true, false);
    else GB_L(false, false, false);
  }
#undef GB_L
}

// tile choice: 64-wide where the dimension is <= 64 (a 128 tile would waste half of its MFMAs) and
// 64-tall when 128-tall tiles would leave most of the 256 CUs without a workgroup
template <int KA, int KB, int EPI, int XOP = XOP_NONE>
static void launch_gemm(const Operand &a, const Operand &b, bool va, bool vb, float *d, long long ldd, double *stats,
                        long long kchunk, unsigned chunks, hipStream_t s, bool bf16, int stat_slots = 1,
                        const float *epi_y = nullptr, const float *epi_ab = nullptr, const uint16_t *epi_w16 = nullptr,
                        long long dchunk = 0, const long long *red_dev = nullptr) {
  // ... and 64 x 64 tiles while those number at most four per CU: the pointwise C -> 4C -> C pairs on a few thousand
  // rows then run on 2-4x as many workgroups (measured: the ten such shapes of the step 735 -> 621 us in total)
  const bool bn64 = b.rows <= 64 || (((a.rows + 63) / 64) * ((b.rows + 63) / 64) * chunks <= 1024);
  const long long blocks128 = ((a.rows + 127) / 128) * ((b.rows + (bn64 ? 63 : 127)) / (bn64 ? 64 : 128)) * chunks;
  const bool bm64 = a.rows <= 64 || blocks128 < 512;
  if (bm64 && bn64) launch_tile<KA, KB, EPI, 64, 64, XOP>(a, b, va, vb, d, ldd, stats, kchunk, chunks, s, bf16, stat_slots, epi_y, epi_ab, epi_w16, dchunk, red_dev);
  else if (bm64) launch_tile<KA, KB, EPI, 64, 128, XOP>(a, b, va, vb, d, ldd, stats, kchunk, chunks, s, bf16, stat_slots, epi_y, epi_ab, epi_w16, dchunk, red_dev);
  else if (bn64) launch_tile<KA, KB, EPI, 128, 64, XOP>(a, b, va, vb, d, ldd, stats, kchunk, chunks, s, bf16, stat_slots, epi_y, epi_ab, epi_w16, dchunk, red_dev);
  else launch_tile<KA, KB, EPI, 128, 128, XOP>(a, b, va, vb, d, ldd, stats, kchunk, chunks, s, bf16, stat_slots, epi_y, epi_ab, epi_w16, dchunk, red_dev);
}

// out[i] = ((part0[i] + part1[i]) + part2[i]) + ...   (chunks copies of `elems` floats, summed in chunk order)
__global__ __launch_bounds__(GTPB) void split_reduce_kernel(const float *__restrict__ part, int chunks, long long elems,
                                                            float *__restrict__ out, int vec) {
  const long long i = ((long long)blockIdx.x * GTPB + threadIdx.x) * 4;
  if (i >= elems) return;
  if (vec && i + 3 < elems) {  // vec: elems % 4 == 0 and both base pointers 16-byte aligned (checked by the host)
    float4 acc = *reinterpret_cast<const float4 *>(part + i);
    for (int c = 1; c < chunks; ++c) {
      const float4 v = *reinterpret_cast<const float4 *>(part + (long long)c * elems + i);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4 *>(out + i) = acc;
  } else {
    for (long long e = i; e < elems && e < i + 4; ++e) {
      float acc = part[e];
      for (int c = 1; c < chunks; ++c) acc += part[(long long)c * elems + e];
      out[e] = acc;
    }
  }
}

static int split_reduce(const float *part, int chunks, long long elems, float *out, hipStream_t s) {
  const long long blocks = (elems / 4 + GTPB) / GTPB;
  const int vec = elems % 4 == 0 && aligned16(part) && aligned16(out);
  hipLaunchKernelGGL(split_reduce_kernel, dim3((unsigned)blocks), dim3(GTPB), 0, s, part, chunks, elems, out, vec);
  return check_launch("gb_gemm split reduce");
}

// Reduction split for forward / dgrad products with few output tiles and a long reduction (the C -> 4C -> C
// pointwise pairs on a few thousand rows: 32-128 tiles of 64 steps each leave most CUs idle and each busy one
// latency-bound).  Returns the number of reduction chunks (1 = no split) and the chunk length.
static int split_reduction(long long rows, long long cols, long long red, long long *kchunk) {
  const long long tiles = ((rows + 63) / 64) * ((cols + 127) / 128);
  long long chunks = 1;
  if ((red >= 1024 && tiles <= 160) || (red >= 512 && tiles <= 64)) {  // measured break-even (memset + column pass)
    chunks = 320 / tiles;
    if (chunks > red / 128) chunks = red / 128;
    if (chunks < 1) chunks = 1;
  }
  long long kc = (red + chunks - 1) / chunks;
  kc = (kc + GK - 1) / GK * GK;
  *kchunk = kc;
  return (int)((red + kc - 1) / kc);
}

// does the caller's workspace (GbGemmOpts.scratch) hold `chunks` partial products of `elems` floats?
static bool split_fits(const GbGemmOpts *o, int chunks, long long elems) {
  return o && o->scratch && (unsigned long long)chunks * (unsigned long long)elems * sizeof(float) <= o->scratch_bytes;
}

}  // namespace gb

using namespace gb;

// Y (P,N) = f(X (P,K)) W(N,K)^T ; aff (optional) = [a(K), b(K)] -> f = relu(a*x+b); stats (optional,
// fp64 [stat_slots][2N], caller-zeroed) += column sums / sums of squares of Y (summed over slots by gb_bn_finalize)
// the layer's gb_bn_finalize as a second launch of the same call (kernels that cannot finish the layer themselves)
static int finalize_after(int rc, const GbBnFinalize *fin, const double *stats, int slots, int N, void *stream) {
  if (rc != GB_OK || !fin) return rc;
  return gb_bn_finalize(stats, slots, fin->P, N, fin->gamma, fin->beta, fin->eps, fin->momentum, fin->running_mean,
                        fin->running_var, fin->ab, 1, stream);
}

// The PLAIN forward / dgrad products prefer the row-streaming kernel only from 65 536 rows: below, a wave gets <= 4 tiles and
// the workgroup's weight image (up to 128 KB) is staged for ~50 us of work - the tiled kernels run those products
// 1.5 - 2x faster (round 4, same-box A/B of the whole step with the threshold at 16 384 / 65 536 / 200 000: 17.73 / 17.42 /
// 17.60 ms; configs[4] 25.83 / 25.65).  Products that NEED that kernel - a device-side row count, the generated first-layer
// operand, the pooled epilogue - go there from 16 384 rows as before (rs_shape_ok).
static constexpr long long RS_PAYS_FROM = 65536;
// the register-direct wgrad (csrc/gemm_wg.hip): from WG_PAYS_FROM rows (or a device-side row count)
constexpr long long WG_PAYS_FROM = 65536;
static bool wg_pays(long long P, const GbGemmOpts *opts) {
  return !(opts && (opts->flags & GB_GEMM_NO_DIRECT)) && (P >= WG_PAYS_FROM || opts_rows(opts) != nullptr);
}
static bool rs_pays(long long P, const GbGemmOpts *opts) { return P >= RS_PAYS_FROM || opts_rows(opts) != nullptr; }

static int gemm_fwd_impl(const float *x, const float *w, const float *aff, const uint16_t *row_w16, float *y,
                         double *stats, int stat_slots, long long P, int K, int N, const GbBnFinalize *fin,
                         const GbGemmOpts *opts, void *stream) {
  if (P < 0 || K < 1 || N < 1 || !x || !w || !y || (stats && stat_slots < 1) || opts_bad(opts)) return GB_EINVAL;
  const bool bf16 = opts_bf16(opts);
  if (fin && (!stats || !fin->gamma || !fin->beta || !fin->ab || fin->P < 1 || fin->training != 1))
    return GB_EINVAL;
  if (P == 0) return fin ? GB_EINVAL : GB_OK;
  if (P / 64 * ((N + 63) / 64) > 0x7fffffffLL) return GB_ERANGE;
  if (rs_pays(P, opts) &&
      rs_gemm_try(x, w, y, aff, stats, stat_slots, nullptr, nullptr, P, K, N, 1, stats ? RS_STATS : RS_STORE,
                  as_stream(stream), bf16, opts_reserved(opts), nullptr, stats ? row_w16 : nullptr, nullptr,
                  opts_rows(opts), opts_split3(opts)))
    return finalize_after(check_launch("gb_gemm_fwd"), fin, stats, stat_slots, N, stream);
  if (opts_rows(opts)) return GB_EINVAL;  // a device-side row count: row-streaming kernel only
  if (!(row_w16 && stats) && !opts_no_ring(opts)) {
    // few-row products: the LDS-DMA ring kernel (csrc/gemm_ring.hip); a long reduction with few output tiles is split
    // over workgroups into the caller's workspace and closed by the same ordered reduce + column-sum pass as below
    RingPlan plan;
    const long long fit = (opts && opts->scratch && P > 0) ? (long long)(opts->scratch_bytes / sizeof(float)) / (P * N) : 1;
    ring_plan(P, N, K, fit > 1, fit, &plan, 0, bf16);
    if (plan.chunks > 1) {
      float *part = static_cast<float *>(opts->scratch);
      if (ring_gemm_try(RING_FWD, x, w, aff, part, P, K, N, nullptr, 1, nullptr, nullptr, plan, (long long)P * N,
                        as_stream(stream), bf16)) {
        int rc = check_launch("gb_gemm_fwd");
        if (rc != GB_OK) return rc;
        if (stats && N % 4 == 0 && aligned16(part) && aligned16(y) && (long long)P * N % 4 == 0)
          return gb_split_col_stats(part, plan.chunks, y, P, N, stats, fin, stream);
        rc = split_reduce(part, plan.chunks, (long long)P * N, y, as_stream(stream));
        if (rc != GB_OK || !stats) return rc;
        return gb_col_stats(y, P, N, stats, fin, stream);
      }
    } else if (ring_gemm_try(RING_FWD, x, w, aff, y, P, K, N, stats, stat_slots, nullptr, nullptr, plan, 0,
                             as_stream(stream), bf16)) {
      return finalize_after(check_launch("gb_gemm_fwd"), fin, stats, stat_slots, N, stream);
    }
  }
  Operand a = {x, P, K, K, aff};
  Operand b = {w, N, K, K, nullptr};
  const bool v = (K % 4 == 0) && aligned16(x) && aligned16(w);
  long long kchunk = 0;
  int chunks = split_reduction(P, N, K, &kchunk);
  // the column pass of the split path does not know row weights; no (or too small a) caller workspace: no split
  if (chunks > 1 && ((row_w16 && stats) || !split_fits(opts, chunks, (long long)P * N))) {
    chunks = 1;
    kchunk = (K + GK - 1) / GK * GK;
  }
  if (chunks > 1) {
    // split reduction: every chunk stores its partial product in the caller's workspace, split_reduce_kernel adds
    // them in chunk order (bit-reproducible, unlike fp32 atomics into a zeroed Y); the BatchNorm sums then need the
    // finished Y, i.e. a (small) column pass
    float *part = static_cast<float *>(opts->scratch);
    launch_gemm<OP_KC, OP_KC, EPI_STORE>(a, b, v, v, part, N, nullptr, kchunk, (unsigned)chunks, as_stream(stream), bf16, 1,
                                         nullptr, nullptr, nullptr, (long long)P * N);
    int rc = check_launch("gb_gemm_fwd");
    if (rc != GB_OK) return rc;
    if (stats && N % 4 == 0 && aligned16(part) && aligned16(y) && (long long)P * N % 4 == 0)
      return gb_split_col_stats(part, chunks, y, P, N, stats, fin, stream);   // reduce + column sums in one sweep
    rc = split_reduce(part, chunks, (long long)P * N, y, as_stream(stream));
    if (rc != GB_OK || !stats) return rc;
    return gb_col_stats(y, P, N, stats, fin, stream);
  }
  if (stats)
    launch_gemm<OP_KC, OP_KC, EPI_STORE_STATS>(a, b, v, v, y, N, stats, kchunk, 1, as_stream(stream), bf16, stat_slots,
                                               nullptr, nullptr, row_w16);
  else launch_gemm<OP_KC, OP_KC, EPI_STORE>(a, b, v, v, y, N, stats, kchunk, 1, as_stream(stream), bf16);
  return finalize_after(check_launch("gb_gemm_fwd"), fin, stats, stat_slots, N, stream);
}

extern "C" int gb_gemm_fwd(const float *x, const float *w, const float *aff, float *y, double *stats,
                           int stat_slots, long long P, int K, int N, const GbBnFinalize *fin, const GbGemmOpts *opts,
                           void *stream) {
  return gemm_fwd_impl(x, w, aff, nullptr, y, stats, stat_slots, P, K, N, fin, opts, stream);
}

// gb_gemm_fwd whose BatchNorm sums weight row p by row_w16[p] (uint16; the array must extend, zero-filled, to the
// next multiple of 32 rows): the rows are the DISTINCT rows of a batch with duplicates (csrc/cyl_rows.hip), the
// sums are those of the full batch.
extern "C" int gb_gemm_fwd_w(const float *x, const float *w, const float *aff, const uint16_t *row_w16, float *y,
                             double *stats, int stat_slots, long long P, int K, int N, const GbBnFinalize *fin,
                             const GbGemmOpts *opts, void *stream) {
  return gemm_fwd_impl(x, w, aff, row_w16, y, stats, stat_slots, P, K, N, fin, opts, stream);
}

// The last layer of a crop stack (reference modules.py:104-124: SharedMLP's final conv + BatchNorm + ReLU, then
// max_pool2d over each crop) WITHOUT storing its output: the weighted BatchNorm sums and per-(tile, seed, crop,
// column) extrema of sign(gamma)*y leave the GEMM's epilogue (csrc/gemm_rs.hip, RS_STATS_POOL_V); gb_pool_pairs finishes
// the pooling once the statistics are known.  Only the row-streaming kernel implements it: GB_EINVAL when the shape is
// not eligible (ask gb_gemm_uses_rs(P, K, N, 0, 3, has_aff)).
extern "C" int gb_gemm_fwd_pool(const float *x, const float *w, const float *aff, const int32_t *row_key,
                                const float *gamma, float *pairs, long long pairs_elems, long long seeds, float *y,
                                double *stats,
                                int stat_slots, long long P, int K, int N, int D, const GbBnFinalize *fin,
                                const GbGemmOpts *opts, void *stream) {
  if (P < 1 || K < 1 || N < 1 || D < 1 || D > 4 || !x || !w || !row_key || !gamma || !pairs || !stats || stat_slots < 1 ||
      opts_bad(opts) || reinterpret_cast<uintptr_t>(row_key) % 16 || reinterpret_cast<uintptr_t>(pairs) % 8)
    return GB_EINVAL;
  if (fin && (!fin->gamma || !fin->beta || !fin->ab || fin->P < 1 || fin->training != 1)) return GB_EINVAL;
  if (P > 0x7fffffffLL - 64) return GB_ERANGE;
  // slot (tile + seed) of D*N floats for every (32-row tile, seed id < seeds) pair the keys can name
  if (seeds < 1 || pairs_elems < ((P + 31) / 32 + seeds) * (long long)D * N) return GB_ERANGE;
  RsPool pool = {};
  pool.key = row_key;
  pool.gamma = gamma;
  pool.pairs = reinterpret_cast<float2 *>(pairs);
  pool.D = D;
  // no Y: a forward-only caller (inference) - nothing can find the arg-max rows afterwards
  if (!rs_gemm_try(x, w, y, aff, stats, stat_slots, nullptr, nullptr, P, K, N, 1, RS_STATS_POOL_V, as_stream(stream),
                   opts_bf16(opts), opts_reserved(opts), nullptr, nullptr, &pool, opts_rows(opts), opts_split3(opts)))
    return GB_EINVAL;
  return finalize_after(check_launch("gb_gemm_fwd_pool"), fin, stats, stat_slots, N, stream);
}

// ---- the 3-input first layer of a stack folded into its consumers (its output Y1 = x0 W1^T is never stored) ----------
// gb_gemm_fwd_gen3: the SECOND layer's forward, Y (P,N) = relu(a1 * (x0 W1^T) + b1) W^T with BatchNorm sums (optionally
// row-weighted), its operand formed in registers from x0 (P,3), W1 (K,3) and ab1 = [a1(K), b1(K)].  Row-streaming
// kernel only: GB_EINVAL when the shape is not eligible (gb_gemm_uses_rs(P, K, N, 0, 1, 1), N <= 128).
extern "C" int gb_gemm_fwd_gen3(const float *x0, const float *w1, const float *ab1, const float *w,
                                const uint16_t *row_w16, float *y, double *stats, int stat_slots, long long P, int K,
                                int N, const GbBnFinalize *fin, const GbGemmOpts *opts, void *stream) {
  if (P < 1 || K < 1 || N < 1 || !x0 || !w1 || !ab1 || !w || !y || (stats && stat_slots < 1) || opts_bad(opts))
    return GB_EINVAL;
  if (fin && (!stats || !fin->gamma || !fin->beta || !fin->ab || fin->P < 1 || fin->training != 1)) return GB_EINVAL;
  RsPool gen = {};
  gen.gen_x = x0;
  gen.gen_w = w1;
  if (!stats) return GB_EINVAL;  // (the eval-mode caller passes a scratch sum buffer: the kernel always forms the sums)
  if (!rs_gemm_try(nullptr, w, y, ab1, stats, stat_slots, nullptr, nullptr, P, K, N, 1, RS_STATS, as_stream(stream),
                   opts_bf16(opts), opts_reserved(opts), nullptr, row_w16, &gen, opts_rows(opts), opts_split3(opts)))
    return GB_EINVAL;
  return finalize_after(check_launch("gb_gemm_fwd_gen3"), fin, stats, stat_slots, N, stream);
}

// gb_gemm_wgrad_gen3: dW (N,K) += dY (P,N)^T relu(a1 * (x0 W1^T) + b1)  (gb_gemm_wgrad with the x operand generated)
extern "C" int gb_gemm_wgrad_gen3(const float *dy, const float *x0, const float *w1, const float *ab1, float *dw,
                                  long long P, int K, int N, const GbGemmOpts *opts, void *stream) {
  if (P < 0 || K < 1 || N < 1 || !dy || !x0 || !w1 || !ab1 || !dw || opts_bad(opts)) return GB_EINVAL;
  if (P == 0) return GB_OK;
  if (wg_pays(P, opts) && wg_wgrad_try(dy, nullptr, ab1, x0, w1, dw, P, K, N, opts_rows(opts), opts_reserved(opts),
                                       as_stream(stream), opts_bf16(opts), opts_split3(opts)))
    return check_launch("gb_gemm_wgrad_gen3");
  Operand a = {dy, N, P, N, nullptr, nullptr, nullptr};
  Operand b = {nullptr, K, P, K, ab1, x0, w1};
  const long long tiles = (long long)((N + (N <= 64 ? 63 : 127)) / (N <= 64 ? 64 : 128)) *
                          ((K + (K <= 64 ? 63 : 127)) / (K <= 64 ? 64 : 128));
  long long chunks = 1024 / tiles;
  if (chunks < 1) chunks = 1;
  long long kchunk = (P + chunks - 1) / chunks;
  kchunk = (kchunk + GK - 1) / GK * GK;
  if (kchunk < 256) kchunk = 256;
  chunks = (P + kchunk - 1) / kchunk;
  if (chunks > 65535) return GB_ERANGE;
  const bool va = (N % 4 == 0) && aligned16(dy);
  launch_gemm<OP_RC, OP_RC, EPI_ATOMIC, XOP_GENB>(a, b, va, false, dw, K, nullptr, kchunk, (unsigned)chunks,
                                              as_stream(stream), opts_bf16(opts), 1, nullptr, nullptr, nullptr, 0,
                                              opts_rows(opts));
  return check_launch("gb_gemm_wgrad_gen3");
}

// dX (P,K) = dY (P,N) W(N,K)   with W in its natural (N,K) row-major layout (no transposed copy).
// Optional fused BatchNorm-backward statistics of the PREVIOUS layer (whose post-ReLU activation is this
// GEMM's input, i.e. dX is its dZ): y_prev (P,K) pre-BN output, ab_prev = [a,b,mean,rstd](K),
// dstats fp64 [stat_slots][2K] (caller-zeroed) += [sum dA, sum dA*xhat],  dA = dX * [a*y+b > 0].
extern "C" int gb_gemm_dgrad(const float *dy, const float *w, float *dx, const float *y_prev,
                             const float *ab_prev, double *dstats, int stat_slots, long long P, int K, int N,
                             double *dstats_total, float *dbeta, float *dgamma, const GbGemmOpts *opts,
                             void *stream) {
  if (P < 0 || K < 1 || N < 1 || !dy || !w || !dx || opts_bad(opts)) return GB_EINVAL;
  const bool bf16 = opts_bf16(opts);
  if (dstats && (!y_prev || !ab_prev || stat_slots < 1)) return GB_EINVAL;
  if ((!dbeta != !dgamma) || (dbeta && !dstats)) return GB_EINVAL;
  // dbeta / dgamma (optional): the previous layer's gb_bn_bwd_reduce runs from this call (dstats_total: where the
  // slot rows' total goes when stat_slots > 1)
  auto done = [&](int rc) {
    if (rc != GB_OK || !dbeta) return rc;
    return gb_bn_bwd_reduce(dstats, stat_slots, K, stat_slots > 1 ? dstats_total : nullptr, dbeta, dgamma, stream);
  };
  if (P == 0) return GB_OK;
  if (P / 64 * ((K + 63) / 64) > 0x7fffffffLL) return GB_ERANGE;
  if (rs_pays(P, opts) &&
      rs_gemm_try(dy, w, dx, nullptr, dstats, stat_slots, y_prev, ab_prev, P, N, K, 0, dstats ? RS_BNBWD : RS_STORE,
                  as_stream(stream), bf16, opts_reserved(opts), nullptr, nullptr, nullptr, opts_rows(opts),
                  opts_split3(opts)))
    return done(check_launch("gb_gemm_dgrad"));
  if (opts_rows(opts)) return GB_EINVAL;  // a device-side row count: row-streaming kernel only
  if (!opts_no_ring(opts)) {
    RingPlan plan;
    const long long fit = (opts && opts->scratch && P > 0) ? (long long)(opts->scratch_bytes / sizeof(float)) / (P * K) : 1;
    ring_plan(P, K, N, fit > 1, fit, &plan, 0, bf16);
    if (plan.chunks > 1) {
      float *part = static_cast<float *>(opts->scratch);
      if (ring_gemm_try(RING_DGRAD, dy, w, nullptr, part, P, K, N, nullptr, 1, nullptr, nullptr, plan, (long long)P * K,
                        as_stream(stream), bf16)) {
        int rc = check_launch("gb_gemm_dgrad");
        if (rc != GB_OK) return rc;
        if (dstats && K % 4 == 0 && aligned16(part) && aligned16(dx) && aligned16(y_prev))
          return done(gb_split_bn_bwd_stats(part, plan.chunks, dx, y_prev, ab_prev, P, K, dstats, stream));
        rc = split_reduce(part, plan.chunks, (long long)P * K, dx, as_stream(stream));
        if (rc != GB_OK || !dstats) return rc;
        return done(gb_bn_bwd_stats(dx, y_prev, ab_prev, nullptr, P, K, 1, dstats, nullptr, nullptr, stream));
      }
    } else if (ring_gemm_try(RING_DGRAD, dy, w, nullptr, dx, P, K, N, dstats, stat_slots, y_prev, ab_prev, plan, 0,
                             as_stream(stream), bf16)) {
      return done(check_launch("gb_gemm_dgrad"));
    }
  }
  Operand a = {dy, P, N, N, nullptr};
  Operand b = {w, K, N, K, nullptr};  // tile rows = k, reduction = n, element (k,n) at w[n*K + k]
  const bool va = (N % 4 == 0) && aligned16(dy);
  const bool vb = (K % 4 == 0) && aligned16(w);
  long long kchunk = 0;
  int chunks = split_reduction(P, K, N, &kchunk);
  if (chunks > 1 && !split_fits(opts, chunks, (long long)P * K)) {  // no (or too small a) caller workspace: no split
    chunks = 1;
    kchunk = (N + GK - 1) / GK * GK;
  }
  if (chunks > 1) {
    float *part = static_cast<float *>(opts->scratch);
    launch_gemm<OP_KC, OP_RC, EPI_STORE>(a, b, va, vb, part, K, nullptr, kchunk, (unsigned)chunks, as_stream(stream), bf16, 1,
                                         nullptr, nullptr, nullptr, (long long)P * K);
    int rc = check_launch("gb_gemm_dgrad");
    if (rc != GB_OK) return rc;
    if (dstats && K % 4 == 0 && aligned16(part) && aligned16(dx) && aligned16(y_prev))   // reduce + BN sums in one sweep
      return done(gb_split_bn_bwd_stats(part, chunks, dx, y_prev, ab_prev, P, K, dstats, stream));
    rc = split_reduce(part, chunks, (long long)P * K, dx, as_stream(stream));
    if (rc != GB_OK || !dstats) return rc;
    return done(gb_bn_bwd_stats(dx, y_prev, ab_prev, nullptr, P, K, 1, dstats, nullptr, nullptr, stream));
  }
  if (dstats)
    launch_gemm<OP_KC, OP_RC, EPI_STORE_BNBWD>(a, b, va, vb, dx, K, dstats, kchunk, 1, as_stream(stream), bf16, stat_slots,
                                               y_prev, ab_prev);
  else
    launch_gemm<OP_KC, OP_RC, EPI_STORE>(a, b, va, vb, dx, K, nullptr, kchunk, 1, as_stream(stream), bf16);
  return done(check_launch("gb_gemm_dgrad"));
}

// dW (N,K) += dY (P,N)^T X (P,K) ; dW must be zeroed by the caller (accumulates with fp32 atomics)
// x_aff (optional) = [a(K), b(K)]: X is used as relu(a_k x + b_k) (the materialisation-free form of the
// previous layer's BatchNorm + ReLU, matching gb_gemm_fwd's prologue)
extern "C" int gb_gemm_wgrad(const float *dy, const float *x, const float *x_aff, float *dw, long long P, int K,
                             int N, const GbGemmOpts *opts, void *stream) {
  if (P < 0 || K < 1 || N < 1 || !dy || !x || !dw || opts_bad(opts)) return GB_EINVAL;
  if (P == 0) return GB_OK;
  if (K <= 4 && !x_aff && N % 4 == 0 && N / 4 <= GTPB && P >= 4096 && reinterpret_cast<uintptr_t>(dy) % 16 == 0 &&
      !opts_rows(opts)) {
    const dim3 grid((unsigned)((P + WSK_ROWS - 1) / WSK_ROWS));
    if (K == 1) hipLaunchKernelGGL(wgrad_smallk_kernel<1>, grid, dim3(GTPB), 0, as_stream(stream), dy, x, dw, P, N);
    else if (K == 2) hipLaunchKernelGGL(wgrad_smallk_kernel<2>, grid, dim3(GTPB), 0, as_stream(stream), dy, x, dw, P, N);
    else if (K == 3) hipLaunchKernelGGL(wgrad_smallk_kernel<3>, grid, dim3(GTPB), 0, as_stream(stream), dy, x, dw, P, N);
    else hipLaunchKernelGGL(wgrad_smallk_kernel<4>, grid, dim3(GTPB), 0, as_stream(stream), dy, x, dw, P, N);
    return check_launch("gb_gemm_wgrad");
  }
  // many rows: both operands straight from global memory into the matrix cores (csrc/gemm_wg.hip)
  if (wg_pays(P, opts) && wg_wgrad_try(dy, x, x_aff, nullptr, nullptr, dw, P, K, N, opts_rows(opts), opts_reserved(opts),
                                       as_stream(stream), opts_bf16(opts), opts_split3(opts)))
    return check_launch("gb_gemm_wgrad");
  if (!opts_rows(opts) && !opts_no_ring(opts) && P <= 131072) {
    // few-row products: split the P reduction for ~one round of workgroups (fp32 atomics into dW, as below)
    RingPlan plan;
    ring_plan(N, K, P, true, 65535, &plan, 1, opts_bf16(opts));
    if (ring_gemm_try(RING_WGRAD, dy, x, x_aff, dw, P, K, N, nullptr, 1, nullptr, nullptr, plan, 0, as_stream(stream),
                      opts_bf16(opts)))
      return check_launch("gb_gemm_wgrad");
  }
  Operand a = {dy, N, P, N, nullptr};  // tile rows = n, reduction = p, element (n,p) at dy[p*N + n]
  Operand b = {x, K, P, K, x_aff};     // tile rows = k, reduction = p, element (k,p) at x[p*K + k]
  const long long tiles = (long long)((N + (N <= 64 ? 63 : 127)) / (N <= 64 ? 64 : 128)) *
                          ((K + (K <= 64 ? 63 : 127)) / (K <= 64 ? 64 : 128));
  // split the P reduction so that ~1024 workgroups exist, chunks a multiple of the reduction step
  const int target_blocks = 1024;
  long long chunks = target_blocks / tiles;
  if (chunks < 1) chunks = 1;
  long long kchunk = (P + chunks - 1) / chunks;
  kchunk = (kchunk + GK - 1) / GK * GK;
  if (kchunk < 256) kchunk = 256;
  chunks = (P + kchunk - 1) / kchunk;
  if (chunks > 65535) return GB_ERANGE;
  const bool va = (N % 4 == 0) && aligned16(dy);
  const bool vb = (K % 4 == 0) && aligned16(x);
  launch_gemm<OP_RC, OP_RC, EPI_ATOMIC>(a, b, va, vb, dw, K, nullptr, kchunk, (unsigned)chunks, as_stream(stream),
                                        opts_bf16(opts), 1, nullptr, nullptr, nullptr, 0, opts_rows(opts));
  return check_launch("gb_gemm_wgrad");
}

// dgrad into the FIRST layer of a stack whose input x has 3 channels (xyz-only grouped rows): dZ = dY W is formed
// tile by tile but never stored - all that is needed downstream are five column sums,
//   sums[slot][0:K] += g, [K:2K] += g*xhat, [2K+jK : ...] += g*x_j (j < 3),   g = dZ*[a*y+b > 0], xhat = (y-mean)*rstd,
// from which the layer's BatchNorm gradients and its weight gradient follow in closed form (gb_la_wx_grad with
// u = 0 and the 12 moments of x).  Saves the dZ write, the bn_bwd_apply pass and the K=3 wgrad.
// Only the row-streaming kernel implements it: GB_EINVAL when the shape is not eligible (gb_gemm_uses_rs(.., 2, 0)).
extern "C" int gb_gemm_dgrad_first(const float *dy, const float *w, const float *y_prev, const float *ab_prev,
                                   const float *x_in, double *sums, int slots, long long P, int K, int N,
                                   const GbGemmOpts *opts, void *stream) {
  if (P < 0 || K < 1 || N < 1 || !dy || !w || !y_prev || !ab_prev || !x_in || !sums || slots < 1 || opts_bad(opts))
    return GB_EINVAL;
  if (P == 0) return GB_OK;
  if (!rs_gemm_try(dy, w, nullptr, nullptr, sums, slots, y_prev, ab_prev, P, N, K, 0, RS_BNBWD_X, as_stream(stream),
                   opts_bf16(opts), opts_reserved(opts), x_in, nullptr, nullptr, opts_rows(opts), opts_split3(opts)))
    return GB_EINVAL;
  return check_launch("gb_gemm_dgrad_first");
}

// gb_gemm_dgrad_first with the first layer's pre-BatchNorm output re-formed from x_in and its weight w_in (K,3) instead of
// read from y_prev (the layer was folded into its consumers: gb_gemm_fwd_gen3)
extern "C" int gb_gemm_dgrad_first_gen3(const float *dy, const float *w, const float *ab_prev, const float *x_in,
                                        const float *w_in, double *sums, int slots, long long P, int K, int N,
                                        const GbGemmOpts *opts, void *stream) {
  if (P < 0 || K < 1 || N < 1 || !dy || !w || !ab_prev || !x_in || !w_in || !sums || slots < 1 || opts_bad(opts))
    return GB_EINVAL;
  if (P == 0) return GB_OK;
  RsPool gen = {};
  gen.gen_w = w_in;
  if (!rs_gemm_try(dy, w, nullptr, nullptr, sums, slots, nullptr, ab_prev, P, N, K, 0, RS_BNBWD_X, as_stream(stream),
                   opts_bf16(opts), opts_reserved(opts), x_in, nullptr, &gen, opts_rows(opts), opts_split3(opts)))
    return GB_EINVAL;
  return check_launch("gb_gemm_dgrad_first_gen3");
}

// mom fp64 [12] += [sum_p w_p x (3), sum_p w_p x x^T (3x3)] of x (P,3), w = row_w or 1; caller-zeroed
namespace gb {
__global__ __launch_bounds__(GTPB) void moments3_kernel(const float *__restrict__ x, const float *__restrict__ rw,
                                                        long long P, double *__restrict__ mom,
                                                        const long long *__restrict__ rows_dev) {
  if (rows_dev) {   // the caller's device-side row count (<= the capacity P)
    const long long pd = *rows_dev;
    P = pd < P ? (pd > 0 ? pd : 0) : P;
  }
  // few workgroups, each striding over the rows: the 12 results are same-address fp64 atomics, which serialise
  // (one per wave of a 512-block grid cost 190 us; one per workgroup of a 64-block grid is free)
  __shared__ double part[GTPB / 64][12];
  double v[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) v[i] = 0.0;
  for (long long p0 = (long long)blockIdx.x * GTPB * 8; p0 < P; p0 += (long long)gridDim.x * GTPB * 8) {
    float s[3] = {0.f, 0.f, 0.f}, mm[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long p = p0 + (long long)u * GTPB + threadIdx.x;
      if (p < P) {
        const float d[3] = {x[p * 3], x[p * 3 + 1], x[p * 3 + 2]};
        const float wt = rw ? rw[p] : 1.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          s[t] += wt * d[t];
#pragma unroll
          for (int q = 0; q < 3; ++q) mm[3 * t + q] += (wt * d[t]) * d[q];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] += (double)s[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) v[3 + i] += (double)mm[i];
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v[i] += __shfl_xor(v[i], off);
  }
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int i = 0; i < 12; ++i) part[threadIdx.x >> 6][i] = v[i];
  __syncthreads();
  if (threadIdx.x < 12) {
    double t = 0.0;
    for (int w = 0; w < GTPB / 64; ++w) t += part[w][threadIdx.x];
    atomicAdd(mom + threadIdx.x, t);
  }
}
}  // namespace gb

extern "C" int gb_moments3(const float *x, const float *row_w, long long P, double *mom, const long long *rows_dev,
                           void *stream) {
  if (P < 0 || !x || !mom) return GB_EINVAL;
  if (P == 0) return GB_OK;
  long long blocks = (P + 8 * GTPB - 1) / (8 * GTPB);
  if (blocks > 128) blocks = 128;
  hipLaunchKernelGGL(moments3_kernel, dim3((unsigned)blocks), dim3(GTPB), 0, as_stream(stream), x, row_w, P, mom, rows_dev);
  return check_launch("gb_moments3");
}

// Both gradient products of a layer as one launch of the ring kernel: each alone would run unsplit on 64 x 64 tiles
// (what the pair kernel gives them; the wgrad's reduction split is the pair kernel's own)
static bool pair_shape(long long P, int K, int N, long long fit, bool bf16) {
  if (K <= 4 || P > 131072 || P / 64 * ((K + 63) / 64) > 0x7fffffffLL) return false;
  if (N % 32 != 0 || P % 32 != 0 || K % 4 != 0 || N % 4 != 0 || P < 64) return false;
  RingPlan pd, pw;
  ring_plan(P, K, N, fit > 1, fit, &pd, 0, bf16);
  ring_plan(N, K, P, true, 65535, &pw, 1, bf16);
  return !pd.big && pd.chunks == 1 && !pw.big && pw.kchunk % 32 == 0 && pw.chunks >= 1;
}

// gb_gemm_dgrad + gb_gemm_wgrad of ONE layer (they read the same dY and nothing of each other).  Where both are few-row
// products of the ring kernel they leave as ONE launch (csrc/gemm_ring.hip, gemm_ring_pair_kernel: the two grids resident
// together); otherwise exactly the two single calls, wgrad first.  Arguments as of the single entries.
extern "C" int gb_gemm_dgrad_wgrad(const float *dy, const float *w, float *dx, const float *y_prev, const float *ab_prev,
                                   double *dstats, int stat_slots, long long P, int K, int N, double *dstats_total,
                                   float *dbeta, float *dgamma, const float *x, const float *x_aff, float *dw,
                                   const GbGemmOpts *opts, void *stream) {
  if (P < 0 || K < 1 || N < 1 || !dy || !w || !dx || !x || !dw || opts_bad(opts)) return GB_EINVAL;
  if (dstats && (!y_prev || !ab_prev || stat_slots < 1)) return GB_EINVAL;
  if ((!dbeta != !dgamma) || (dbeta && !dstats)) return GB_EINVAL;
  if (P == 0) return GB_OK;
  const long long fit = (opts && opts->scratch) ? (long long)(opts->scratch_bytes / sizeof(float)) / (P * K) : 1;
  const bool pairable = !opts_rows(opts) && !opts_no_ring(opts) && !(opts && (opts->flags & GB_GEMM_NO_PAIR)) &&
                        !rs_pays(P, opts) && !wg_pays(P, opts) && pair_shape(P, K, N, fit, opts_bf16(opts));
  if (pairable && ring_pair_try(dy, w, dx, dstats, stat_slots, y_prev, ab_prev, x, x_aff, dw, P, K, N, as_stream(stream),
                                opts_bf16(opts))) {
    const int rc = check_launch("gb_gemm_dgrad_wgrad");
    if (rc != GB_OK || !dbeta) return rc;
    return gb_bn_bwd_reduce(dstats, stat_slots, K, stat_slots > 1 ? dstats_total : nullptr, dbeta, dgamma, stream);
  }
  const int rc = gb_gemm_wgrad(dy, x, x_aff, dw, P, K, N, opts, stream);
  if (rc != GB_OK) return rc;
  return gb_gemm_dgrad(dy, w, dx, y_prev, ab_prev, dstats, stat_slots, P, K, N, dstats_total, dbeta, dgamma, opts, stream);
}

// Many weight gradients in one call (round 6): items[i] is a gb_gemm_wgrad {dy, x, x_aff, dw, P, K, N} whose dW may sit in a
// wider matrix (ldw >= K floats between its rows).  The few-row ones that suit the LDS-DMA ring kernel leave as ONE grid
// per 63 products (csrc/gemm_ring.hip gemm_ring_group_kernel); anything else (tall products for the register-direct
// kernel, <= 4 input channels, unaligned shapes) runs exactly as its own gb_gemm_wgrad - those need ldw == K.
extern "C" int gb_gemm_wgrad_group(const GbWgradItem *items, int count, const GbGemmOpts *opts, void *stream) {
  if (count < 0 || (count && !items) || opts_bad(opts) || opts_rows(opts)) return GB_EINVAL;
  for (int i = 0; i < count; ++i) {
    const GbWgradItem &w = items[i];
    if (w.P < 0 || w.K < 1 || w.N < 1 || !w.dy || !w.x || !w.dw || w.ldw < w.K) return GB_EINVAL;
  }
  RingWgrad ring[126];
  int nr = 0;
  auto flush = [&]() {
    if (nr) ring_group_launch(ring, nr, as_stream(stream), opts_bf16(opts));
    nr = 0;
    return check_launch("gb_gemm_wgrad_group");
  };
  for (int i = 0; i < count; ++i) {
    const GbWgradItem &w = items[i];
    if (w.P == 0) continue;
    const bool smallk = w.K <= 4 && !w.x_aff && w.N % 4 == 0 && w.N / 4 <= GTPB && w.P >= 4096;
    const bool tall = wg_pays(w.P, opts) && wg_wgrad_suits(w.P, w.K, w.N, false, opts_bf16(opts) || opts_split3(opts),
                                                           opts_reserved(opts));
    if (!smallk && !tall && !opts_no_ring(opts) && w.P <= 131072 &&
        ring_group_suits(w.dy, w.x, w.x_aff, w.dw, w.P, w.K, w.N, w.ldw)) {
      ring[nr++] = {w.dy, w.x, w.x_aff, w.dw, w.P, w.K, w.N, w.ldw};
      if (nr == 126) {
        const int rc = flush();
        if (rc != GB_OK) return rc;
      }
      continue;
    }
    if (w.ldw != w.K) return GB_EINVAL;
    const int rc = gb_gemm_wgrad(w.dy, w.x, w.x_aff, w.dw, w.P, w.K, w.N, opts, stream);
    if (rc != GB_OK) return rc;
  }
  return flush();
}

// Would gb_gemm_wgrad_group put this product into a grouped launch (16-byte aligned operands assumed)?  What a caller
// needs to know before it relies on ldw != K.
extern "C" int gb_gemm_wgrad_groups(long long P, int K, int N, int precision, int reserved_cus, unsigned flags) {
  const bool skeleton16 = precision == GB_PREC_BF16 || precision == GB_PREC_F32_SPLIT3;
  if (P < 32 || P % 32 != 0 || K % 4 != 0 || N % 4 != 0 || K < 4 || N < 4 || P > 131072 || (flags & GB_GEMM_NO_RING)) return 0;
  if (K <= 4) return 0;
  if (!(flags & GB_GEMM_NO_DIRECT) && P >= WG_PAYS_FROM && wg_wgrad_suits(P, K, N, false, skeleton16, reserved_cus)) return 0;
  return 1;
}

// Which kernel a gb_gemm_fwd (kind 0) / gb_gemm_dgrad (1) / gb_gemm_wgrad (2) call of this shape launches for 16-byte
// aligned fp32 operands and default options: 0 = the register-staged tiles of this file, 1 = the row-streaming kernel
// (csrc/gemm_rs.hip), 2 = the LDS-DMA ring kernel (csrc/gemm_ring.hip), 3 = the column-reduction wgrad, 4 = the
// register-direct tall wgrad (csrc/gemm_wg.hip).  Pure host-side
// introspection (no launch), used by bench.py to attribute timings per kernel.
extern "C" int gb_gemm_kernel_for2(int kind, long long P, int K, int N, int fused_stats, int has_aff, int precision,
                                   int reserved_cus, unsigned flags) {
  const bool bf16 = precision == GB_PREC_BF16, skeleton16 = bf16 || precision == GB_PREC_F32_SPLIT3;
  const bool no_ring = flags & GB_GEMM_NO_RING, no_direct = flags & GB_GEMM_NO_DIRECT;
  if (kind == 3)  // gb_gemm_dgrad_wgrad: 2 = ONE launch of the ring kernel carries both products, 0 = the two single calls
    return (!(flags & GB_GEMM_NO_PAIR) && !no_ring && P < RS_PAYS_FROM && (no_direct || P < WG_PAYS_FROM) && P > 0 && K > 0 &&
            pair_shape(P, K, N, (long long)(GB_GEMM_SCRATCH_BYTES / sizeof(float)) / (P * K), bf16)) ? 2 : 0;
  if (kind == 2) {
    if (K <= 4 && !has_aff && N % 4 == 0 && N / 4 <= GTPB && P >= 4096) return 3;
    if (!no_direct && P >= WG_PAYS_FROM && wg_wgrad_suits(P, K, N, false, skeleton16, reserved_cus)) return 4;
    return (!no_ring && P % 32 == 0 && K % 4 == 0 && N % 4 == 0 && K >= 4 && N >= 4 && P <= 131072) ? 2 : 0;
  }
  if ((P >= RS_PAYS_FROM || fused_stats >= 2) && gb_gemm_uses_rs(P, K, N, kind, fused_stats, has_aff)) return 1;
  if (no_ring) return 0;
  if (kind == 0) return K % 32 == 0 ? 2 : 0;
  return (N % 32 == 0 && K % 4 == 0) ? 2 : 0;
}

extern "C" int gb_gemm_kernel_for(int kind, long long P, int K, int N, int fused_stats, int has_aff) {
  return gb_gemm_kernel_for2(kind, P, K, N, fused_stats, has_aff, GB_PREC_F32, 0, 0u);
}

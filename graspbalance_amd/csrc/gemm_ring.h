// LDS-DMA ring GEMM for the few-row products (csrc/gemm_ring.hip): internal interface used by the C entry points in
// gemm_cl.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gb {

enum { RK_KC = 0,     // operand element (tile row r, reduction index k) at src[r * ld + k]
       RK_RC = 1 };   // ... at src[k * ld + r]
enum { RG_STORE = 0, RG_STATS = 1, RG_ATOMIC = 2, RG_BNBWD = 3 };
enum { RING_FWD = 0, RING_DGRAD = 1, RING_WGRAD = 2 };

struct RingPlan {
  bool big;          // 128 x 128 tiles (else 64 x 64)
  int chunks;        // reduction chunks (grid.y)
  long long kchunk;  // reduction indices per chunk
};

// tile / split choice for an M x N output with reduction length `red`
void ring_plan(long long M, long long N, long long red, bool want_split, long long max_chunks, RingPlan *plan,
               int atomics = 0, bool bf16 = false);

// kind RING_FWD  : d (P,N) = f(a (P,K)) b (N,K)^T, aff = [a(K), b(K)] or NULL; stats: BatchNorm column sums of d
//      RING_DGRAD: d (P,K) = a (P,N) b (N,K);  stats (with epi_y, epi_ab): the previous layer's BatchNorm-backward sums
//      RING_WGRAD: d (N,K) += a (P,N)^T f(b (P,K)), aff = [a(K), b(K)] or NULL (fp32 atomics; plan.chunks splits P)
// dchunk != 0 (FWD / DGRAD with plan.chunks > 1): chunk c stores its partial product at d + c * dchunk.
// bf16: operands rounded to bf16 in registers, v_mfma_f32_32x32x16_bf16, fp32 accumulation (GbGemmOpts.precision).
// Returns false (nothing launched) when the shape / alignment does not suit the kernel.
bool ring_gemm_try(int kind, const float *a, const float *b, const float *aff, float *d, long long P, int K, int N,
                   double *stats, int stat_slots, const float *epi_y, const float *epi_ab, const RingPlan &plan,
                   long long dchunk, hipStream_t s, bool bf16 = false);

// one layer's dgrad + wgrad in one launch (gemm_ring_pair_kernel); false = nothing launched, use the two single calls
bool ring_pair_try(const float *dy, const float *w, float *dx, double *dstats, int stat_slots, const float *y_prev,
                   const float *ab_prev, const float *x, const float *x_aff, float *dw, long long P, int K, int N,
                   hipStream_t s, bool bf16);

// many wgrads in one launch (gemm_ring_group_kernel): dW (N, ldw) += dY (P,N)^T f(X (P,K)) per item
struct RingWgrad {
  const float *dy, *x, *aff;
  float *dw;
  long long P;
  int K, N, ldw;
};
bool ring_group_suits(const float *dy, const float *x, const float *aff, const float *dw, long long P, int K, int N, int ldw);
void ring_group_launch(const RingWgrad *items, int count, hipStream_t s, bool bf16);

}  // namespace gb

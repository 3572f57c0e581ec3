// Row-streaming GEMM (csrc/gemm_rs.hip): internal interface used by the C entry points in gemm_cl.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gb {

enum { RS_STORE = 0,   // D only
       RS_STATS = 1,   // + column sums of D and D^2                       (BatchNorm batch statistics)
       RS_BNBWD = 2,   // + column sums of g and g*xhat, g = D*[a*y+b > 0]  (BatchNorm-backward sums)
       RS_BNBWD_X = 3,   // those two + sum g*x_j (j < 3) for the layer's 3-channel input x; D is NOT stored
       RS_STATS_POOL_V = 6 }; // RS_STATS with per-row keys: besides D (optional) and the weighted sums, per (tile, seed, crop,
                            // column) the extreme VALUE of sign(gamma)*y leaves the tile (pairs = float[(tile + seed)][D][C]);
                            // the arg-max row is found by value (y == y*) where y is stored

// D (P,C) = f(A (P,R)) B (R,C);  w_kc = 1: B[r][c] = w[c*R + r], 0: B[r][c] = w[r*C + c].
// Returns false (nothing launched) when the shape does not suit the kernel; the caller then uses the
// LDS-tiled kernel.  bf16 / reserved_cus: GbGemmOpts.precision / .reserved_cus of the call.
// RS_STATS_POOL_V inputs / outputs (see gemm_rs.hip): row keys, BatchNorm weight, crops per seed, partial extrema
struct RsPool {
  const int32_t *key;
  const float *gamma;
  float2 *pairs;
  int D;
  const float *gen_x = nullptr;    // RS_STATS: generate the A operand from these (P,3) rows (a = nullptr) / RS_BNBWD_X: y
  const float *gen_w = nullptr;    // ... and this 3-input first-layer weight (R,3) / (C,3)
};

bool rs_gemm_try(const float *a, const float *w, float *d, const float *aff, double *stats, int slots,
                 const float *epi_y, const float *epi_ab, long long P, int R, int C, int w_kc, int epi,
                 hipStream_t s, bool bf16, int reserved_cus, const float *epi_x = nullptr,
                 const uint16_t *epi_w16 = nullptr, const RsPool *pool = nullptr, const long long *rows_dev = nullptr,
                 bool split3 = false);   // split3: GB_PREC_F32_SPLIT3 where an instantiation fits, fp32 MFMA otherwise

}  // namespace gb

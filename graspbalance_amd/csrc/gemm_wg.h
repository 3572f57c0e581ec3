// Register-direct tall wgrad (csrc/gemm_wg.hip): internal interface used by the C entry points in gemm_cl.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gb {

// dW (N,K) += dY (P,N)^T f(X (P,K)), reduction over MANY rows (the caller decides from where on it pays); bf16: both
// operands rounded to bf16 on their way into the matrix cores (fp32 in memory, fp32 accumulation).
//   x      : X (P,K), or nullptr with gen_x / gen_w: X[p][k] = gen_x[p] . gen_w[k] (the never-stored output of a 3-input
//            first layer: gen_x (P,3), gen_w (K,3)), evaluated as gemm_rs.hip's lin3
//   aff    : optional [a(K), b(K)]: f(x) = relu(a_k x + b_k) (required with gen_x)
//   rows_dev: optional device-side row count (<= P)
// Returns false (nothing launched) when the shape does not suit the kernel.
bool wg_wgrad_try(const float *dy, const float *x, const float *aff, const float *gen_x, const float *gen_w, float *dw,
                  long long P, int K, int N, const long long *rows_dev, int reserved_cus, hipStream_t s, bool bf16 = false,
                  bool split3 = false);   // split3: GB_PREC_F32_SPLIT3 (an fp32 mode on the bf16 skeleton)
// shape test only (no launch): what wg_wgrad_try accepts for 16-byte aligned operands - with the SAME wave count and grid
// it would launch (skeleton16: bf16 or split3, the 4-wave 16-row skeleton), so introspection and dispatch cannot disagree
bool wg_wgrad_suits(long long P, int K, int N, bool gen, bool skeleton16, int reserved_cus);

}  // namespace gb

// Shared helpers of libgraspbal_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/graspbal.h"

namespace gb {

void set_last_error(const char *what, hipError_t err);
void clear_last_error();

// returns GB_OK / GB_ELAUNCH after a kernel launch on `stream`
inline int check_launch(const char *what) {
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_last_error(what, err);
    return GB_ELAUNCH;
  }
  return GB_OK;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: set it once on every device a kernel is
// launched on (`done` = one bit per device ordinal, a function-local static of the caller).  Two threads racing on
// the first launch both set it - harmless, the call is idempotent - and the bit is published after the attribute.
template <typename K>
inline void allow_dynamic_lds(K kern, int bytes, std::atomic<unsigned long long> &done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  done.fetch_or(bit, std::memory_order_release);
}

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- wave64 reductions on DPP (no LDS): idempotent ops only (max / min) ----------------------
// One VOP2-DPP instruction per step: v = op(dpp(v), v).  quad_perm xor1, xor2, row_half_mirror,
// row_mirror leave the row result in all 16 lanes of each row; row_bcast:15 (rows 1,3) and
// row_bcast:31 (rows 2,3) fold the four rows into lane 63.  Lanes a step does not write keep
// their value (the register is tied in/out).  hipcc pads nothing inside asm, so every DPP read
// carries the two wait states it needs after the VALU write of its source (s_nop 1).
#define GB_DPP_STEP(OP, V, CTRL) \
  asm volatile("s_nop 1\n\t" OP " %0, %0, %0 " CTRL : "+v"(V))

__device__ __forceinline__ float row_max_f32(float v) {  // result in every lane of each 16-lane row
  GB_DPP_STEP("v_max_f32_dpp", v, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
  GB_DPP_STEP("v_max_f32_dpp", v, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
  GB_DPP_STEP("v_max_f32_dpp", v, "row_half_mirror row_mask:0xf bank_mask:0xf");
  GB_DPP_STEP("v_max_f32_dpp", v, "row_mirror row_mask:0xf bank_mask:0xf");
  return v;
}
__device__ __forceinline__ unsigned row_min_u32(unsigned v) {
  GB_DPP_STEP("v_min_u32_dpp", v, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
  GB_DPP_STEP("v_min_u32_dpp", v, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
  GB_DPP_STEP("v_min_u32_dpp", v, "row_half_mirror row_mask:0xf bank_mask:0xf");
  GB_DPP_STEP("v_min_u32_dpp", v, "row_mirror row_mask:0xf bank_mask:0xf");
  return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {  // wave-uniform result
  v = row_max_f32(v);
  GB_DPP_STEP("v_max_f32_dpp", v, "row_bcast:15 row_mask:0xa bank_mask:0xf");
  GB_DPP_STEP("v_max_f32_dpp", v, "row_bcast:31 row_mask:0xc bank_mask:0xf");
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  v = row_min_u32(v);
  GB_DPP_STEP("v_min_u32_dpp", v, "row_bcast:15 row_mask:0xa bank_mask:0xf");
  GB_DPP_STEP("v_min_u32_dpp", v, "row_bcast:31 row_mask:0xc bank_mask:0xf");
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ int lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int prefix_popc(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// ---- BatchNorm finalisation shared by the kernels / entry points that produce its sums ---------------------------
// (A "last workgroup finishes the layer" form - ticket after the atomics, sums read back with atomic adds of zero -
// was built for the column-sum kernels and measured: +8 us on a 13 us kernel, more than the 6 us launch it saves.
// The finalisation stays its own tiny kernel, launched from the same C call as the producer.)
struct BnFinalize {  // == GbBnFinalize (include/graspbal.h)
  const float *gamma, *beta;
  float *running_mean, *running_var;
  float *ab;
  long long P;
  float eps, momentum;
  int training;
};

// one column of gb_bn_finalize (training): s1 = sum y, s2 = sum y^2 over P rows
__device__ __forceinline__ void bn_finalize_column(const BnFinalize &f, double s1, double s2, int c, int C) {
  const double m = s1 / (double)f.P;
  double v = s2 / (double)f.P - m * m;  // biased variance (normalisation)
  if (v < 0.0) v = 0.0;
  const float mean = (float)m, var = (float)v;
  if (f.running_mean) {
    const double unbiased = f.P > 1 ? v * (double)f.P / (double)(f.P - 1) : v;
    f.running_mean[c] = (1.0f - f.momentum) * f.running_mean[c] + f.momentum * mean;
    f.running_var[c] = (1.0f - f.momentum) * f.running_var[c] + f.momentum * (float)unbiased;
  }
  const float rstd = 1.0f / sqrtf(var + f.eps);
  const float a = f.gamma[c] * rstd;
  f.ab[c] = a;
  f.ab[C + c] = f.beta[c] - mean * a;
  f.ab[2 * C + c] = mean;
  f.ab[3 * C + c] = rstd;
}

// ---- precision of the SharedMLP contractions ------------------------------------------------------------------------
// GbGemmOpts.precision = GB_PREC_BF16: every GEMM of the channel-last MLP path (forward, dgrad, wgrad) rounds both
// operands to bf16 as they enter the matrix cores (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate) and keeps
// accumulating in fp32; BatchNorm statistics, element-wise passes, geometry and the tensors in HBM stay fp32.
// Reductions shorter than 16 (the xyz-only first layers) stay on the fp32 instruction.  Per call.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
inline bool opts_bf16(const GbGemmOpts *o) { return o && o->precision == GB_PREC_BF16; }
inline bool opts_split3(const GbGemmOpts *o) { return o && o->precision == GB_PREC_F32_SPLIT3; }
inline int opts_reserved(const GbGemmOpts *o) { return o ? o->reserved_cus : 0; }
inline const long long *opts_rows(const GbGemmOpts *o) { return o ? o->rows_dev : nullptr; }
inline bool opts_no_ring(const GbGemmOpts *o) { return o && (o->flags & GB_GEMM_NO_RING); }
inline bool opts_bad(const GbGemmOpts *o) {
  return o && ((o->precision != GB_PREC_F32 && o->precision != GB_PREC_BF16 && o->precision != GB_PREC_F32_SPLIT3) || o->reserved_cus < 0 ||
               o->reserved_cus > 128 || (o->scratch && reinterpret_cast<uintptr_t>(o->scratch) % 16 != 0) ||
               (o->rows_dev && reinterpret_cast<uintptr_t>(o->rows_dev) % 8 != 0) ||
               (o->flags & ~(GB_GEMM_NO_RING | GB_GEMM_NO_PAIR | GB_GEMM_NO_DIRECT)));
}

struct __attribute__((packed, aligned(4))) f3 {
  float x, y, z;
};

}  // namespace gb

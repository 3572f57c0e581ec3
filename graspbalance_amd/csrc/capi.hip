// Error reporting + version of the C-ABI (include/graspbal.h).
#include <stdio.h>
#include <string.h>

#include "gb_common.h"

namespace gb {
static thread_local char g_err[256] = "";

void set_last_error(const char *what, hipError_t err) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(err));
}
void clear_last_error() { g_err[0] = 0; }
}  // namespace gb

namespace gb {
std::atomic<int> g_mlp_bf16{0};
}

extern "C" int gb_set_mlp_precision(int precision) {
  if (precision != GB_PREC_F32 && precision != GB_PREC_BF16) return GB_EINVAL;
  gb::g_mlp_bf16.store(precision == GB_PREC_BF16 ? 1 : 0, std::memory_order_relaxed);
  return GB_OK;
}
extern "C" int gb_get_mlp_precision(void) { return gb::mlp_bf16() ? GB_PREC_BF16 : GB_PREC_F32; }

extern "C" int gb_abi_version(void) { return GB_ABI_VERSION; }
extern "C" const char *gb_last_error(void) { return gb::g_err; }

// Streams restricted to a set of compute units (hipExtStreamCreateWithCUMask).  cu_mask: `words` 32-bit words, bit i =
// CU i may run this stream's kernels.  Used to split the chip between the training stream and the side stream that
// runs the next batch's furthest-point sampling (prefetch.py): one 1024-thread workgroup per cloud for ~2 ms gets CUs
// of its own instead of slowing down - and being slowed down by - the statically partitioned GEMMs.
extern "C" int gb_stream_create_cu_mask(const uint32_t *cu_mask, int words, void **stream) {
  if (!cu_mask || words < 1 || !stream) return GB_EINVAL;
  hipStream_t s = nullptr;
  hipError_t err = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, cu_mask);
  if (err != hipSuccess) {
    gb::set_last_error("gb_stream_create_cu_mask", err);
    return GB_ELAUNCH;
  }
  *stream = s;
  return GB_OK;
}

extern "C" int gb_stream_destroy(void *stream) {
  if (!stream) return GB_EINVAL;
  hipError_t err = hipStreamDestroy(reinterpret_cast<hipStream_t>(stream));
  if (err != hipSuccess) {
    gb::set_last_error("gb_stream_destroy", err);
    return GB_ELAUNCH;
  }
  return GB_OK;
}

extern "C" int gb_device_cu_count(int *count) {
  if (!count) return GB_EINVAL;
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return GB_ELAUNCH;
  *count = n;
  return GB_OK;
}

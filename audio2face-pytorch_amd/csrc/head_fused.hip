// Fused vertex head + loss + head backward (placeholder until the kernel lands; see include/a2f_hip.h).
#include "a2f_common.h"
extern "C" size_t a2f_head_fused_ws_bytes(int64_t B, int64_t n, int64_t K) { (void)B; (void)n; (void)K; return 16; }
extern "C" int a2f_head_fused(const float*, const float*, const float*, const float*, int64_t, const float*, int64_t, int64_t, int64_t,
                              int64_t, float, float, float*, float*, float*, float*, float*, void*, size_t, a2f_stream_t) {
  return a2f::set_err(A2F_EINVAL, "a2f_head_fused: not implemented yet");
}

// FaceFormer decoder recurrence (placeholder until the kernels land; see include/a2f_hip.h).
#include "a2f_common.h"
extern "C" size_t a2f_ff_save_bytes(int64_t S, int64_t T) { return (size_t)S * T * 64 * 4 * 16; }
extern "C" int a2f_ff_decoder_fwd(const float*, const float*, const float*, int64_t, int64_t, int32_t, float*, void*, size_t, a2f_stream_t) {
  return a2f::set_err(A2F_EINVAL, "a2f_ff_decoder_fwd: not implemented yet");
}
extern "C" int a2f_ff_decoder_bwd(const float*, const float*, const float*, const float*, int64_t, int64_t, int32_t, const void*, size_t,
                                  float*, float*, float*, a2f_stream_t) {
  return a2f::set_err(A2F_EINVAL, "a2f_ff_decoder_bwd: not implemented yet");
}

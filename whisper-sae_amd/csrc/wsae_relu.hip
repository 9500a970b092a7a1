// ReLU SAE dense path (model.py:260-322).  Scheduled after the TopK path (SURVEY.md section 7,
// stage 8); the entry points exist so the ABI is stable and fail loudly until the kernels land.
#include "wsae_common.h"

extern "C" int wsae_relu_forward(wsae_ctx*, const float*, const void*, int32_t, const int32_t*, int32_t, float, float*,
                                 float*, wsae_stats*, float*, void*) {
    wsae_set_error("wsae_relu_forward: the ReLU SAE kernels are not part of this build yet");
    return WSAE_ERR_INVALID;
}

extern "C" int wsae_relu_backward(wsae_ctx*, const float*, const void*, int32_t, const int32_t*, int32_t, float,
                                  const float*, const float*, float*, void*) {
    wsae_set_error("wsae_relu_backward: the ReLU SAE kernels are not part of this build yet");
    return WSAE_ERR_INVALID;
}

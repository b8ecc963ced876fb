// Internal (non-ABI) entry points shared between translation units of libllmie.so.
#pragma once
#include "device_utils.cuh"

namespace llmie {

enum : int { EPI_NONE_ = 0, EPI_SWIGLU_ = 1 };

// fp16 activations, fp16 W[N,K]; epi = 0 plain (bias/residual optional), 1 = SwiGLU over row
// pairs (i, N/2+i) writing y[M, N/2].  Defined in linear.hip.
int linear_f16_nk(const half_t *x, const half_t *W, half_t *y, int M, int K, int N, int epi,
                  const half_t *bias, const half_t *residual, hipStream_t st);

}  // namespace llmie

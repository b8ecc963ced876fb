// Entry points declared in include/llmie.h whose kernels are not written yet.  They fail loudly
// (LLMIE_ERR_UNSUPPORTED); nothing falls back to a CPU path.
#include "llmie_common.h"

using namespace llmie;

#define PENDING(name) \
    do { set_error(name ": not implemented in this build"); return LLMIE_ERR_UNSUPPORTED; } while (0)

extern "C" int llmie_linear_fp8(const void *, const uint8_t *, const float *, void *, int, int, int, const void *,
                                const void *, void *, size_t, llmie_stream) { PENDING("linear_fp8"); }
extern "C" size_t llmie_linear_fp8_workspace_bytes(int, int) { return 0; }
extern "C" int llmie_quantize_fp8(const void *, uint8_t *, float *, int, int, llmie_stream) { PENDING("quantize_fp8"); }

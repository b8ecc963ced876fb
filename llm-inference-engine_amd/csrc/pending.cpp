// Entry points declared in include/llmie.h whose kernels are not written yet.  They fail loudly
// (LLMIE_ERR_UNSUPPORTED); nothing falls back to a CPU path.
#include "llmie_common.h"

using namespace llmie;

#define PENDING(name) \
    do { set_error(name ": not implemented in this build"); return LLMIE_ERR_UNSUPPORTED; } while (0)

extern "C" int llmie_linear_w8a16(const void *, const int8_t *, const void *, void *, int, int, int, const void *,
                                  const void *, llmie_stream) { PENDING("linear_w8a16"); }
extern "C" int llmie_linear_w4a16(const void *, const uint8_t *, const void *, void *, int, int, int, int,
                                  const void *, const void *, llmie_stream) { PENDING("linear_w4a16"); }
extern "C" int llmie_linear_fp8(const void *, const uint8_t *, const float *, void *, int, int, int, const void *,
                                const void *, void *, size_t, llmie_stream) { PENDING("linear_fp8"); }
extern "C" size_t llmie_linear_fp8_workspace_bytes(int, int) { return 0; }
extern "C" int llmie_quantize_w8(const void *, int8_t *, void *, int, int, llmie_stream) { PENDING("quantize_w8"); }
extern "C" int llmie_quantize_w4(const void *, uint8_t *, void *, int, int, int, llmie_stream) { PENDING("quantize_w4"); }
extern "C" int llmie_quantize_fp8(const void *, uint8_t *, float *, int, int, llmie_stream) { PENDING("quantize_fp8"); }
extern "C" size_t llmie_decoder_workspace_bytes(const llmie_decoder_config *) { return 0; }
extern "C" llmie_decoder *llmie_decoder_create(const llmie_decoder_config *, const llmie_layer_weights *, void *, size_t) {
    set_error("decoder_create: not implemented in this build");
    return nullptr;
}
extern "C" void llmie_decoder_destroy(llmie_decoder *) {}
extern "C" int llmie_decoder_forward(llmie_decoder *, const void *, void *, void *, void *, int, int, const int32_t *,
                                     llmie_stream) { PENDING("decoder_forward"); }
extern "C" int llmie_lm_head_sample(llmie_decoder *, void *, const void *, const llmie_matrix *, llmie_weight_format,
                                    void *, int32_t *, void *, int32_t *, void *, int, int, int32_t *, uint8_t *,
                                    int32_t *, int, int, const int32_t *, int, llmie_stream) { PENDING("lm_head_sample"); }

// C++ API mirror, part 6: the 18 `launch*` templates of src/kernels/includes/*.cuh plus
// CublasWrapper (cublas_utils.cuh:17-75), as inline adaptors over the C ABI (include/llmie.h).
// Shapes are taken from the TensorWrappers exactly where the reference launchers take them;
// a failing C-ABI status becomes the reference's LLM_CHECK exception.
#pragma once
#include "params.hpp"
#include "tensor.hpp"
#include "weights.hpp"

typedef void *llmieBlasHandle_t;  // stands where the reference takes cublasHandle_t / cublasLtHandle_t (ignored)

// GEMM context: name and constructor shape of the reference's CublasWrapper; holds no vendor library.  Like the reference's
// (whose cuBLAS handle owns its workspace) it owns the scratch the GEMMs use: fp32 split-K slabs, allocated HERE, once -- the
// C ABI never allocates on the compute path.  A shape that needs more than the wrapper holds runs the non-split kernels.
class CublasWrapper {
private:
    llmie_dtype dtype_ = LLMIE_F32;
    void *ws_ = nullptr;
    size_t ws_bytes_ = 0;

public:
    explicit CublasWrapper(llmieBlasHandle_t = nullptr, llmieBlasHandle_t = nullptr, size_t workspace_bytes = size_t(128) << 20) {
        if (workspace_bytes && hipMalloc(&ws_, workspace_bytes) == hipSuccess) ws_bytes_ = workspace_bytes;
    }
    ~CublasWrapper() {
        if (ws_) (void)hipFree(ws_);
    }
    CublasWrapper(const CublasWrapper &) = delete;
    CublasWrapper &operator=(const CublasWrapper &) = delete;
    void setFP32GemmConfig() { dtype_ = LLMIE_F32; }
    void setFP16GemmConfig() { dtype_ = LLMIE_F16; }
    llmie_dtype dtype() const { return dtype_; }
    // the wrapper's scratch if it covers `need` bytes, else none
    void *workspace(size_t need) const { return need <= ws_bytes_ ? ws_ : nullptr; }
    size_t workspace_bytes(size_t need) const { return need <= ws_bytes_ ? ws_bytes_ : 0; }
};

namespace llmie_api {
inline hipStream_t st() { return launch_stream(); }

// grow-only device scratch for launchers whose C-ABI form takes a workspace (decode attention)
inline void *scratch(size_t bytes) {
    static thread_local void *buf = nullptr;
    static thread_local size_t cap = 0;
    if (bytes > cap) {
        if (buf) CHECK(hipFree(buf));
        CHECK(hipMalloc(&buf, bytes));
        cap = bytes;
    }
    return buf;
}
}  // namespace llmie_api

// input_embedding.cuh:8-12
template <typename T>
void launchInputEmbedding(TensorWrapper<int> *input_ids, TensorWrapper<T> *output, EmbeddingWeight<T> *embed_table) {
    const int num_tokens = output->shape[0], hidden = output->shape[1];
    LLM_CHECK_WITH_INFO(num_tokens == input_ids->shape[0], "Input ids 1st shape should equal to 1st shape of output");
    const int vocab = embed_table->shape.empty() ? (1 << 30) : embed_table->shape[0];
    LLMIE_CALL(llmie_input_embedding(input_ids->data, embed_table->data, output->data, num_tokens, hidden, vocab,
                                     llmie_api::dtype_of<T>(), llmie_api::st()));
}

// cal_padding_offset.cuh:16-20
inline void launchCalPaddingOffset(TensorWrapper<int> *padding_offset, TensorWrapper<int> *cum_seqlens,
                                   TensorWrapper<int> *input_lengths) {
    const int batch = padding_offset->shape[0], max_q_len = padding_offset->shape[1];
    LLM_CHECK_WITH_INFO(batch == input_lengths->shape[0],
                        "Input lengths numbers should equal to padding offset batch size dimension!");
    LLM_CHECK_WITH_INFO(batch == cum_seqlens->shape[0] - 1,
                        "Cumulative sequence length should equal to padding offset batch size dimension plus 1!");
    LLMIE_CALL(llmie_cal_padding_offset(padding_offset->data, cum_seqlens->data, input_lengths->data, batch, max_q_len,
                                        llmie_api::st()));
}

// build_causal_mask.cuh:10-14
template <typename T>
void launchBuildCausalMasks(TensorWrapper<T> *mask, TensorWrapper<int> *q_lens, TensorWrapper<int> *k_lens) {
    LLMIE_CALL(llmie_build_causal_mask(mask->data, q_lens->data, k_lens->data, mask->shape[0], mask->shape[1],
                                       mask->shape[2], llmie_api::dtype_of<T>(), llmie_api::st()));
}

// rmsnorm.cuh:10-15
template <typename T>
void launchRMSNorm(TensorWrapper<T> *decoder_out, TensorWrapper<T> *decoder_residual,
                   LayerNormWeight<T> *attention_norm_weight, float eps, bool is_last = false) {
    (void)is_last;
    LLMIE_CALL(llmie_rmsnorm(decoder_out->data, decoder_residual ? decoder_residual->data : nullptr,
                             attention_norm_weight->gamma, eps, decoder_out->shape[0], decoder_out->shape[1],
                             llmie_api::dtype_of<T>(), llmie_api::st()));
}

// add_residual_and_rmsnorm.cuh:12-18 (norm->bias is the bias; scale = gamma)
template <typename T>
void launchFusedAddBiasResidualAndRMSNorm(TensorWrapper<T> *residual, TensorWrapper<T> *decoder_out,
                                          BaseWeight<T> *norm, T *scale, float eps) {
    LLMIE_CALL(llmie_fused_add_bias_residual_rmsnorm(residual ? residual->data : nullptr, decoder_out->data,
                                                     norm ? norm->bias : nullptr, scale, eps, decoder_out->shape[0],
                                                     decoder_out->shape[1], llmie_api::dtype_of<T>(), llmie_api::st()));
}

// add_residual.cuh:10-14
template <typename T>
void launchAddResidual(TensorWrapper<T> *residual, TensorWrapper<T> *decoder_out, bool is_print = false) {
    (void)is_print;
    LLMIE_CALL(llmie_add_residual(residual->data, decoder_out->data, decoder_out->shape[0], decoder_out->shape[1],
                                  llmie_api::dtype_of<T>(), llmie_api::st()));
}

// linear.cuh:14-21.  3-D inputs/outputs are flattened on dims 1,2 (linear.cu:37-43); trans_b selects
// y = x.W^T (W [N,K]) vs y = x.W (W [K,N]) -- the intent of the reference.
template <typename T>
void launchLinearGemm(TensorWrapper<T> *input, BaseWeight<T> *weight, TensorWrapper<T> *output,
                      CublasWrapper *cublas_wrapper, bool trans_a = false, bool trans_b = false) {
    LLM_CHECK_WITH_INFO(!trans_a, "trans_a is not used by any caller of the reference and is not supported");
    const int Am = input->shape[0];
    const int An = input->shape.size() == 3 ? input->shape[1] * input->shape[2] : input->shape[1];
    const int Cm = output->shape[0];
    const int Cn = output->shape.size() == 3 ? output->shape[1] * output->shape[2] : output->shape[1];
    int opBm = weight->shape[0], opBn = weight->shape[1];
    if (trans_b) std::swap(opBm, opBn);
    LLM_CHECK_WITH_INFO(An == opBm, "2nd dim of weight MUST = 1st dim of input");
    LLM_CHECK_WITH_INFO(Am == Cm && opBn == Cn, "output shape should be equal to weight shape");
    const size_t need = (cublas_wrapper && trans_b && std::is_same<T, half>::value) ? llmie_linear_workspace_bytes(LLMIE_W_F16, Cm, An, Cn) : 0;
    LLMIE_CALL(llmie_linear(input->data, weight->data, output->data, Cm, An, Cn, trans_b ? 1 : 0, nullptr, nullptr,
                            llmie_api::dtype_of<T>(), need ? cublas_wrapper->workspace(need) : nullptr,
                            need ? cublas_wrapper->workspace_bytes(need) : 0, llmie_api::st()));
}

// linear.cuh:23-29: per (b,h): C = A.B or A.B^T
template <typename T>
void launchLinearStridedBatchGemm(TensorWrapper<T> *input1, TensorWrapper<T> *input2, TensorWrapper<T> *output,
                                  CublasWrapper *cublas_wrapper, bool trans_a = false, bool trans_b = false) {
    (void)cublas_wrapper;
    LLM_CHECK_WITH_INFO(!trans_a, "trans_a is not supported");
    const int Am = input1->shape[2], An = input1->shape[3];
    int opBm = input2->shape[2], opBn = input2->shape[3];
    if (trans_b) std::swap(opBm, opBn);
    const int Cm = output->shape[2], Cn = output->shape[3];
    LLM_CHECK_WITH_INFO(An == opBm, "2nd dim of weight MUST = 1st dim of input");
    LLM_CHECK_WITH_INFO(Am == Cm && opBn == Cn, "output shape should be equal to weight shape");
    const int batch = input1->shape[0] * input1->shape[1];
    LLM_CHECK_WITH_INFO(batch == input2->shape[0] * input2->shape[1], "dim 0 and dim 1 wrong!");
    LLMIE_CALL(llmie_batched_gemm(input1->data, input2->data, output->data, batch, Cm, Cn, An, trans_b ? 1 : 0,
                                  llmie_api::dtype_of<T>(), llmie_api::st()));
}

// qkv_bias_and_rope.cuh:13-23.  The reference never applies the bias (9-K10): neither do we here.
template <typename T>
void launchFusedQKVAddBiasAndTransposeAndRope(TensorWrapper<T> *q_buf, TensorWrapper<T> *k_buf, TensorWrapper<T> *v_buf,
                                              TensorWrapper<T> *QKV, BaseWeight<T> *qkv,
                                              TensorWrapper<int> *padding_offset, TensorWrapper<int> *history_length,
                                              TensorWrapper<int> *input_length,
                                              LlamaAttentionStaticParams *static_params) {
    (void)qkv;
    (void)input_length;
    const int token_num = QKV->shape[0], qkv_head_num = QKV->shape[1], head_size = QKV->shape[2];
    const int batch = q_buf->shape[0], head_num = q_buf->shape[1], seq_len = q_buf->shape[2];
    LLM_CHECK_WITH_INFO(k_buf->shape[1] == v_buf->shape[1], "k and v should have same head_num");
    LLM_CHECK_WITH_INFO(k_buf->shape[1] == (qkv_head_num - head_num) / 2, "k and v should have same head_num");
    LLM_CHECK_WITH_INFO(q_buf->shape[3] == head_size, "head_size does not match!");
    LLMIE_CALL(llmie_qkv_bias_transpose_rope(q_buf->data, k_buf->data, v_buf->data, QKV->data, nullptr,
                                             padding_offset->data, history_length->data, batch, seq_len, token_num,
                                             head_num, k_buf->shape[1], head_size, static_params->rotary_embedding_dim,
                                             static_params->rotary_embedding_base, llmie_api::dtype_of<T>(),
                                             llmie_api::st()));
}

// rope.cuh:13-17
template <typename T>
void launchRope(TensorWrapper<T> *qkv_buf, TensorWrapper<int> *step, LlamaAttentionStaticParams *static_params) {
    const int batch = qkv_buf->shape[0], qkv_head_num = qkv_buf->shape[1], head_size = qkv_buf->shape[2];
    const int head_num = static_params->head_num, kv_head_num = static_params->kv_head_num;
    LLM_CHECK_WITH_INFO(qkv_head_num == head_num + 2 * kv_head_num, "qkv_buf heads != head_num + 2*kv_head_num");
    LLMIE_CALL(llmie_rope_decode(qkv_buf->data, batch, head_num, kv_head_num, head_size, step->getVal(), nullptr,
                                 static_params->rotary_embedding_dim, static_params->rotary_embedding_base,
                                 llmie_api::dtype_of<T>(), llmie_api::st()));
}

// decoder_self_attention.cuh:12-22
template <typename T>
void launchDecoderMaskedMultiHeadAttention(TensorWrapper<T> *qkv_buf, BaseWeight<T> *qkv, TensorWrapper<int> *layer_id,
                                           TensorWrapper<T> *k_cache, TensorWrapper<T> *v_cache,
                                           TensorWrapper<bool> *finished, TensorWrapper<int> *step,
                                           TensorWrapper<T> *mha_output, LlamaAttentionStaticParams *static_params) {
    (void)finished;
    (void)static_params;
    const int batch = qkv_buf->shape[0], qkv_head_num = qkv_buf->shape[1], head_size = qkv_buf->shape[2];
    const int kv_head_num = k_cache->shape[2], max_seq_len = k_cache->shape[3];
    const int head_num = qkv_head_num - 2 * kv_head_num;
    const size_t ws = llmie_decoder_mha_workspace_bytes(batch, head_num, head_size, max_seq_len);
    LLMIE_CALL(llmie_decoder_mha(qkv_buf->data, qkv ? qkv->bias : nullptr, k_cache->data, v_cache->data,
                                 mha_output->data, layer_id->getVal(), batch, head_num, kv_head_num, head_size,
                                 max_seq_len, step->getVal(), nullptr, llmie_api::scratch(ws), ws,
                                 llmie_api::dtype_of<T>(), llmie_api::st()));
}

// concat_past_kv.cuh:10-18
template <typename T>
void launchConcatKVCache(TensorWrapper<T> *k_src, TensorWrapper<T> *v_src, TensorWrapper<int> *layer_id,
                         TensorWrapper<int> *cur_query_length, TensorWrapper<int> *history_length,
                         TensorWrapper<T> *k_dst, TensorWrapper<T> *v_dst) {
    const int batch = k_src->shape[0], kv_head_num = k_src->shape[1], max_q_len = k_src->shape[2];
    const int head_size = k_src->shape[3], max_seq_len = k_dst->shape[3], layer = layer_id->getVal();
    LLMIE_CALL(llmie_concat_kv(k_src->data, k_dst->data, cur_query_length->data, history_length->data, layer, batch,
                               kv_head_num, max_q_len, max_seq_len, head_size, llmie_api::dtype_of<T>(), llmie_api::st()));
    LLMIE_CALL(llmie_concat_kv(v_src->data, v_dst->data, cur_query_length->data, history_length->data, layer, batch,
                               kv_head_num, max_q_len, max_seq_len, head_size, llmie_api::dtype_of<T>(), llmie_api::st()));
}

// repeat_kv.cuh:10-17
template <typename T>
void launchRepeatKVCache(TensorWrapper<T> *k_cache_src, TensorWrapper<T> *v_cache_src,
                         TensorWrapper<int> *context_length, TensorWrapper<int> *layer_id,
                         TensorWrapper<T> *k_cache_dst, TensorWrapper<T> *v_cache_dst) {
    const int batch = context_length->shape[0], kv_head_num = k_cache_src->shape[2];
    const int max_seq_len = k_cache_src->shape[3], head_num = k_cache_dst->shape[1];
    const int max_k_len = k_cache_dst->shape[2], head_size = k_cache_dst->shape[3], layer = layer_id->getVal();
    LLMIE_CALL(llmie_repeat_kv(v_cache_src->data, v_cache_dst->data, context_length->data, layer, batch, head_num,
                               kv_head_num, max_k_len, max_seq_len, head_size, llmie_api::dtype_of<T>(), llmie_api::st()));
    LLMIE_CALL(llmie_repeat_kv(k_cache_src->data, k_cache_dst->data, context_length->data, layer, batch, head_num,
                               kv_head_num, max_k_len, max_seq_len, head_size, llmie_api::dtype_of<T>(), llmie_api::st()));
}

// scale_and_mask_and_softmax.cuh:11-16
template <typename T>
void launchFusedScaleMaskAndSoftmax(TensorWrapper<T> *qk, TensorWrapper<T> *mask, TensorWrapper<T> *attention_weights,
                                    float scale) {
    LLMIE_CALL(llmie_scale_mask_softmax(qk->data, mask->data, attention_weights->data, scale, qk->shape[0],
                                        qk->shape[1], qk->shape[2], qk->shape[3], llmie_api::dtype_of<T>(),
                                        llmie_api::st()));
}

// transpose_and_remove_padding.cuh:9-13
template <typename T>
void launchFusedTransposeAndRemovePadding(TensorWrapper<T> *padded_qkv_buf, TensorWrapper<int> *padding_offset,
                                          TensorWrapper<T> *lineared_qkv) {
    LLMIE_CALL(llmie_transpose_remove_padding(padded_qkv_buf->data, lineared_qkv->data, padding_offset->data,
                                              lineared_qkv->shape[0], padded_qkv_buf->shape[0], padded_qkv_buf->shape[2],
                                              padded_qkv_buf->shape[1], padded_qkv_buf->shape[3],
                                              llmie_api::dtype_of<T>(), llmie_api::st()));
}

// silu_and_mul.cuh:10-13
template <typename T> void launchSiluAndMul(TensorWrapper<T> *input, TensorWrapper<T> *output) {
    LLM_CHECK_WITH_INFO(input->shape.size() == 3 && input->shape[1] == 2, "SiluAndMul input must be [tokens, 2, inter]");
    LLMIE_CALL(llmie_silu_and_mul(input->data, output->data, input->shape[0], input->shape[2],
                                  llmie_api::dtype_of<T>(), llmie_api::st()));
}

// topk.cuh:45-51.  K = last dim of final_topk_ids (the reference hard-codes 5 and ignores the
// buffers' shapes); blocks per row = topk_ids->shape[2].
template <typename T>
void launchTopKForBeamSearch(TensorWrapper<T> *probs, TensorWrapper<int> *topk_ids, TensorWrapper<T> *topk_vals,
                             TensorWrapper<int> *final_topk_ids, TensorWrapper<T> *final_topk_vals) {
    const int rows = probs->shape[0], vocab = probs->shape[1];
    const int K = final_topk_ids->shape.back();
    const int bpr = topk_ids->shape.size() >= 4 ? topk_ids->shape[2] : 1;
    LLMIE_CALL(llmie_topk(probs->data, topk_ids->data, topk_vals->data, final_topk_ids->data, final_topk_vals->data,
                          rows, vocab, K, bpr, llmie_api::dtype_of<T>(), llmie_api::st()));
}

// sampling.cuh:12-19
template <typename T>
void launchSampling(TensorWrapper<int> *topk_id, TensorWrapper<T> *topk_val, TensorWrapper<int> *seqlen,
                    TensorWrapper<bool> *is_finished, TensorWrapper<int> *output_id, MapStringToInt *params) {
    static_assert(sizeof(bool) == 1, "finished flags are one byte each");
    LLMIE_CALL(llmie_sampling(topk_id->data, topk_val->data, seqlen->data,
                              reinterpret_cast<uint8_t *>(is_finished->data), output_id->data, topk_id->shape[0],
                              topk_id->shape[1], params->at("step"), nullptr, params->at("end_id"),
                              params->at("vocab_size"), llmie_api::dtype_of<T>(), llmie_api::st()));
}

// C++ API mirror, part 7: the five layer classes of src/layers/includes/*.h with the reference's
// constructor arguments, method names, forward() signatures and TensorMap keys:
//   LlamaSelfAttentionLayer   self_attention.h:17-62    keys in : attention_input, step, finished, layer_id
//                                                        keys out: attention_output, all_k_cache, all_v_cache
//   LlamaFFNLayer             ffn.h:14-54                ffn_input -> ffn_output (may alias)
//   LlamaSelfDecoder          self_decoder.h:14-86       decoder_input, step, finished, layer_id -> decoder_output, caches
//   LlamaContextAttentionLayer context_attention.h       attention_input, padding_offset, history_length, input_length,
//                                                        context_length, attention_mask, layer_id
//   LlamaContextDecoder       context_decoder.h:10-90    decoder_input, history_length, input_length, context_length, layer_id
// Bodies are new: scratch buffers are grow-only members (no malloc/free/sync per call, reference:
// self_attention.cpp:29-45,150), every launch goes to llmie_api::launch_stream() (null stream by default,
// like the reference; the `stream` constructor argument is stored but, as in the reference, never used
// because callers pass it uninitialised: self_decoder_example.cpp:41), no printf on the hot path.
#pragma once
#include <cmath>

#include "allocator.hpp"
#include "kernels.hpp"

namespace llmie_api {
// grow-only device buffer wrapped as TensorWrapper<T> with a per-call shape
template <typename T> class Scratch {
    BaseAllocator *alloc_;
    T *ptr_ = nullptr;
    size_t cap_ = 0;
    std::unique_ptr<TensorWrapper<T>> view_;

public:
    explicit Scratch(BaseAllocator *a) : alloc_(a) {}
    ~Scratch() { release(); }
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;
    TensorWrapper<T> *get(const std::vector<int> &shape, Device dev = Device::GPU) {
        size_t n = 1;
        for (int d : shape) n *= static_cast<size_t>(d);
        if (n > cap_) {
            release();
            if (alloc_) alloc_->malloc(&ptr_, sizeof(T) * n, false);
            else CHECK(hipMalloc(reinterpret_cast<void **>(&ptr_), sizeof(T) * n));
            cap_ = n;
        }
        view_.reset(new TensorWrapper<T>(dev, getTensorType<T>(), shape, ptr_));
        return view_.get();
    }
    void release() {
        if (ptr_) {
            if (alloc_) alloc_->free(ptr_, false);
            else CHECK(hipFree(ptr_));
        }
        ptr_ = nullptr;
        cap_ = 0;
        view_.reset();
    }
};
}  // namespace llmie_api

// Lazily created fused engine (include/llmie.h section 3) shared by the two decoder classes: eligible when every
// matrix of every layer is in HF layout ([N,K], is_transposed == true).
template <typename T> class EngineHolder {
    llmie_decoder *engine = nullptr;
    void *ws = nullptr;
    const void *key = nullptr;
    int batch_cap = 0, max_seq = 0;
    void *pf_ws = nullptr;
    size_t pf_cap = 0;

public:
    ~EngineHolder() { destroy(); }
    EngineHolder() = default;
    EngineHolder(const EngineHolder &) = delete;
    EngineHolder &operator=(const EngineHolder &) = delete;
    static bool usable(std::vector<LlamaLayerWeight<T> *> *lw) {
        for (auto *w : *lw)
            if (!(w->self_attention_weight.qkv.is_transposed && w->self_attention_weight.output.is_transposed &&
                  w->ffn_weight.gate_and_up.is_transposed && w->ffn_weight.down.is_transposed))
                return false;
        return true;
    }
    void destroy() {
        if (engine) llmie_decoder_destroy(engine);
        if (ws) (void)hipFree(ws);
        if (pf_ws) (void)hipFree(pf_ws);
        engine = nullptr;
        ws = pf_ws = nullptr;
        pf_cap = 0;
    }
    llmie_decoder *get(std::vector<LlamaLayerWeight<T> *> *lw, int num_layer, int head_num, int kv_head_num, int head_size,
                       int inter_size, const LlamaAttentionStaticParams &sp, float eps, int batch, int max_seq_len) {
        if (engine && key == lw->data() && batch <= batch_cap && max_seq_len == max_seq) return engine;
        destroy();
        llmie_decoder_config cfg{};
        cfg.head_num = head_num;
        cfg.kv_head_num = kv_head_num;
        cfg.head_size = head_size;
        cfg.inter_size = inter_size;
        cfg.num_layers = num_layer;
        cfg.vocab_size = 0;
        cfg.max_seq_len = max_seq_len;
        cfg.max_batch = batch;
        cfg.rotary_dim = sp.rotary_embedding_dim;
        cfg.rotary_base = sp.rotary_embedding_base;
        cfg.rms_eps = eps;
        cfg.dtype = llmie_api::dtype_of<T>();
        cfg.wfmt = std::is_same<T, half>::value ? LLMIE_W_F16 : LLMIE_W_F32;
        cfg.int4_group = 128;
        std::vector<llmie_layer_weights> layers(num_layer);
        for (int l = 0; l < num_layer; ++l) {
            LlamaLayerWeight<T> *w = lw->at(l);
            layers[l].attn_norm_gamma = w->attention_norm_weight.gamma;
            layers[l].ffn_norm_gamma = w->ffn_norm_weight.gamma;
            layers[l].qkv = {w->self_attention_weight.qkv.data, nullptr, w->self_attention_weight.qkv.bias};
            layers[l].o = {w->self_attention_weight.output.data, nullptr, w->self_attention_weight.output.bias};
            layers[l].gate_up = {w->ffn_weight.gate_and_up.data, nullptr, nullptr};
            layers[l].down = {w->ffn_weight.down.data, nullptr, nullptr};
        }
        const size_t bytes = llmie_decoder_workspace_bytes(&cfg);
        LLM_CHECK_WITH_INFO(bytes > 0, "invalid decoder configuration");
        CHECK(hipMalloc(&ws, bytes));
        engine = llmie_decoder_create(&cfg, layers.data(), ws, bytes);
        LLM_CHECK_WITH_INFO(engine != nullptr, std::string(llmie_last_error()));
        key = lw->data();
        batch_cap = batch;
        max_seq = max_seq_len;
        return engine;
    }
    void *prefill_workspace(const llmie_decoder_config *cfg, int tokens, int batch, size_t *bytes) {
        *bytes = llmie_decoder_prefill_workspace_bytes(cfg, tokens, batch);
        if (*bytes > pf_cap) {
            if (pf_ws) CHECK(hipFree(pf_ws));
            CHECK(hipMalloc(&pf_ws, *bytes));
            pf_cap = *bytes;
        }
        return pf_ws;
    }
};

// ------------------------------------------------------------------------------------------------
template <typename T> class LlamaSelfAttentionLayer {
private:
    int head_num, head_size, hidden_units, repeats_per_kv, kv_head_num;
    float scale;
    LlamaAttentionStaticParams *attention_static_params;
    hipStream_t stream;
    BaseAllocator *allocator;
    CublasWrapper *cublas_wrapper;
    llmie_api::Scratch<T> qkv_scratch, mha_scratch;
    TensorWrapper<T> *qkv_buf = nullptr, *mha_output = nullptr;

public:
    LlamaSelfAttentionLayer(int head_num, int kv_head_num, int head_size, LlamaAttentionStaticParams *attention_params,
                            hipStream_t stream, CublasWrapper *cublas_wrapper, BaseAllocator *allocator)
        : head_num(head_num), head_size(head_size), hidden_units(head_num * head_size),
          repeats_per_kv(kv_head_num ? head_num / kv_head_num : 0), kv_head_num(kv_head_num),
          scale(1.0f / std::sqrt(static_cast<float>(head_size))), attention_static_params(attention_params),
          stream(stream), allocator(allocator), cublas_wrapper(cublas_wrapper), qkv_scratch(allocator),
          mha_scratch(allocator) {
        LLM_CHECK_WITH_INFO(kv_head_num > 0 && head_num % kv_head_num == 0, "kv_head_num must be a factor of head_num");
        // the kernels read head geometry from the static params (rope.cu:68-69)
        attention_static_params->head_num = head_num;
        attention_static_params->kv_head_num = kv_head_num;
        attention_static_params->head_size = head_size;
    }
    LlamaAttentionStaticParams *getAttentionStaticParams() { return attention_static_params; }

    void allocateMemory(LlamaAttentionDynamicParams *dynamic_params) {
        const int bs = dynamic_params->batch_size;
        qkv_buf = qkv_scratch.get({bs, head_num + 2 * kv_head_num, head_size});
        mha_output = mha_scratch.get({bs, hidden_units});
    }
    void freeBuf() {
        qkv_scratch.release();
        mha_scratch.release();
        qkv_buf = mha_output = nullptr;
    }
    // self_attention.cpp:63-151
    void forward(TensorMap *inputs, TensorMap *outputs, LlamaAttentionWeights<T> *weights,
                 LlamaAttentionDynamicParams *dynamic_params) {
        allocateMemory(dynamic_params);
        Tensor *attention_input = inputs->at("attention_input");
        launchLinearGemm(attention_input->wrap<T>(), &weights->qkv, qkv_buf, cublas_wrapper, false,
                         weights->qkv.is_transposed);
        Tensor *attention_output = outputs->at("attention_output");
        Tensor *key_cache = outputs->at("all_k_cache");
        Tensor *value_cache = outputs->at("all_v_cache");
        Tensor *finished = inputs->at("finished");
        Tensor *step = inputs->at("step");
        Tensor *layer_id = inputs->at("layer_id");
        launchRope(qkv_buf, step->wrap<int>(), attention_static_params);
        launchDecoderMaskedMultiHeadAttention<T>(qkv_buf, &weights->qkv, layer_id->wrap<int>(), key_cache->wrap<T>(),
                                                 value_cache->wrap<T>(), finished->wrap<bool>(), step->wrap<int>(),
                                                 mha_output, attention_static_params);
        launchLinearGemm(mha_output, &weights->output, attention_output->wrap<T>(), cublas_wrapper, false,
                         weights->output.is_transposed);
    }
};

// ------------------------------------------------------------------------------------------------
template <typename T> class LlamaFFNLayer {
private:
    int head_num, head_size, intermediate_size, hidden_units;
    int count = -1;
    hipStream_t stream;
    BaseAllocator *allocator;
    CublasWrapper *cublas_wrapper;
    llmie_api::Scratch<T> swiglu_scratch, down_scratch;
    TensorWrapper<T> *swiglu_input = nullptr, *down_proj_input = nullptr;

public:
    LlamaFFNLayer(int head_num, int head_size, int intermediate_size, hipStream_t stream, CublasWrapper *cublas_wrapper,
                  BaseAllocator *allocator)
        : head_num(head_num), head_size(head_size), intermediate_size(intermediate_size),
          hidden_units(head_num * head_size), stream(stream), allocator(allocator), cublas_wrapper(cublas_wrapper),
          swiglu_scratch(allocator), down_scratch(allocator) {}

    void allocateMemory(LlamaAttentionDynamicParams *dynamic_params) { allocateMemory(dynamic_params->num_tokens); }
    void allocateMemory(const int &rows) {
        swiglu_input = swiglu_scratch.get({rows, 2, intermediate_size});
        down_proj_input = down_scratch.get({rows, intermediate_size});
    }
    void freeBuf() {
        swiglu_scratch.release();
        down_scratch.release();
        swiglu_input = down_proj_input = nullptr;
    }
    // ffn.cpp:76-144; rows = num_tokens if > 0 else batch_size (ffn.cpp:84-88)
    void forward(TensorMap *inputs, TensorMap *outputs, LlamaFFNWeights<T> *weights,
                 LlamaAttentionDynamicParams *dynamic_params) {
        Tensor *ffn_input = inputs->at("ffn_input");
        Tensor *ffn_output = outputs->at("ffn_output");
        const int rows = ffn_input->shape[0];
        LLM_CHECK_WITH_INFO(rows == (dynamic_params->num_tokens > 0 ? dynamic_params->num_tokens : dynamic_params->batch_size),
                            "ffn_input rows must equal num_tokens (context) or batch_size (decode)");
        allocateMemory(rows);
        ++count;
        if (std::is_same<T, half>::value && weights->gate_and_up.is_transposed && rows <= 64 &&
            hidden_units % 32 == 0 && intermediate_size % 16 == 0) {
            // decode: gate/up GEMV with the SwiGLU epilogue (one kernel instead of GEMM + launchSiluAndMul)
            const size_t need = cublas_wrapper ? llmie_linear_workspace_bytes(LLMIE_W_F16, rows, hidden_units, 2 * intermediate_size) : 0;
            LLMIE_CALL(llmie_linear_swiglu(ffn_input->wrap<T>()->data, weights->gate_and_up.data, down_proj_input->data,
                                           rows, hidden_units, 2 * intermediate_size, LLMIE_F16,
                                           need ? cublas_wrapper->workspace(need) : nullptr,
                                           need ? cublas_wrapper->workspace_bytes(need) : 0, llmie_api::st()));
        } else {
            launchLinearGemm(ffn_input->wrap<T>(), &weights->gate_and_up, swiglu_input, cublas_wrapper, false,
                             weights->gate_and_up.is_transposed);
            launchSiluAndMul(swiglu_input, down_proj_input);
        }
        launchLinearGemm(down_proj_input, &weights->down, ffn_output->wrap<T>(), cublas_wrapper, false,
                         weights->down.is_transposed);
    }
};

// ------------------------------------------------------------------------------------------------
template <typename T> class LlamaSelfDecoder {
private:
    int head_num, kv_head_num, head_size, intermediate_size, num_layer, hidden_units;
    float rmsnorm_eps;
    LlamaAttentionStaticParams static_params;
    hipStream_t stream;
    CublasWrapper *cublas_wrapper;
    BaseAllocator *allocator;
    llmie_api::Scratch<T> residual_scratch;
    TensorWrapper<T> *decoder_residual = nullptr;
    LlamaSelfAttentionLayer<T> *self_attention;
    LlamaFFNLayer<T> *ffn;
    DataType data_type;
    EngineHolder<T> engine_holder;

public:
    LlamaSelfDecoder(const int &head_num, const int &kv_head_num, const int &head_size, const int &intermediate_size,
                     const int &num_layer, const LlamaAttentionStaticParams &attn_params, const float &rmsnorm_eps,
                     const hipStream_t &stream, CublasWrapper *const &cublas_wrapper, BaseAllocator *const &allocator)
        : head_num(head_num), kv_head_num(kv_head_num), head_size(head_size), intermediate_size(intermediate_size),
          num_layer(num_layer), hidden_units(head_num * head_size), rmsnorm_eps(rmsnorm_eps), static_params(attn_params),
          stream(stream), cublas_wrapper(cublas_wrapper), allocator(allocator), residual_scratch(allocator),
          data_type(getTensorType<T>()) {
        self_attention = new LlamaSelfAttentionLayer<T>(head_num, kv_head_num, head_size, &static_params, stream,
                                                        cublas_wrapper, allocator);
        ffn = new LlamaFFNLayer<T>(head_num, head_size, intermediate_size, stream, cublas_wrapper, allocator);
    }
    ~LlamaSelfDecoder() {
        delete self_attention;
        delete ffn;
    }
    LlamaSelfDecoder(const LlamaSelfDecoder &) = delete;
    LlamaSelfDecoder &operator=(const LlamaSelfDecoder &) = delete;

    void allocateMemory(LlamaAttentionDynamicParams *dynamic_params) {
        decoder_residual = residual_scratch.get({dynamic_params->batch_size, hidden_units});
    }
    void freeBuf() {
        residual_scratch.release();
        decoder_residual = nullptr;
    }
    // set false to force the per-kernel loop even for HF-layout weights (used by the parity tests)
    bool use_fused_engine = true;

    // self_decoder.cpp:24-122
    void forward(TensorMap *input_tensors, std::vector<LlamaLayerWeight<T> *> *layer_weights, TensorMap *output_tensors,
                 LlamaAttentionDynamicParams *dynamic_params) {
        Tensor *decoder_input = input_tensors->at("decoder_input");
        Tensor *step = input_tensors->at("step");
        Tensor *finished = input_tensors->at("finished");
        Tensor *decoder_output = output_tensors->at("decoder_output");
        Tensor *all_k_cache = output_tensors->at("all_k_cache");
        Tensor *all_v_cache = output_tensors->at("all_v_cache");
        Tensor *layer_id = input_tensors->at("layer_id");
        LLM_CHECK_WITH_INFO(decoder_input->wrap<T>()->data != nullptr, "The data pointer of tensor inserted into TensorMap is nullptr!");
        LLM_CHECK_WITH_INFO(step->wrap<int>()->data != nullptr, "The data pointer of tensor inserted into TensorMap is nullptr!");
        LLM_CHECK_WITH_INFO(finished->wrap<bool>()->data != nullptr, "The data pointer of tensor inserted into TensorMap is nullptr!");
        LLM_CHECK_WITH_INFO(static_cast<int>(layer_weights->size()) >= num_layer, "not enough layer weights");
        const int batch = dynamic_params->batch_size;

        if (use_fused_engine && EngineHolder<T>::usable(layer_weights)) {
            llmie_decoder *engine = engine_holder.get(layer_weights, num_layer, head_num, kv_head_num, head_size,
                                                      intermediate_size, static_params, rmsnorm_eps, batch,
                                                      all_k_cache->shape[3]);
            LLMIE_CALL(llmie_decoder_forward(engine, decoder_input->wrap<T>()->data, decoder_output->wrap<T>()->data,
                                             all_k_cache->wrap<T>()->data, all_v_cache->wrap<T>()->data, batch,
                                             step->wrap<int>()->getVal(), nullptr, llmie_api::st()));
            return;
        }
        // generic path: the reference's kernel sequence, launcher by launcher
        allocateMemory(dynamic_params);
        int cur_layer = 0;
        TensorWrapper<int> layer_tensor(Device::CPU, getTensorType<int>(), {1}, &cur_layer);
        (void)layer_id;
        TensorMap self_attention_inputs{{"attention_input", decoder_input}, {"layer_id", &layer_tensor},
                                        {"step", step}, {"finished", finished}};
        TensorMap self_attention_outputs{{"attention_output", decoder_output}, {"all_k_cache", all_k_cache},
                                         {"all_v_cache", all_v_cache}};
        for (cur_layer = 0; cur_layer < num_layer; ++cur_layer) {
            LlamaLayerWeight<T> *w = layer_weights->at(cur_layer);
            decoder_input = self_attention_inputs.at("attention_input");
            launchRMSNorm(decoder_input->wrap<T>(), decoder_residual, &w->attention_norm_weight, rmsnorm_eps);
            self_attention->forward(&self_attention_inputs, &self_attention_outputs, &w->self_attention_weight, dynamic_params);
            launchFusedAddBiasResidualAndRMSNorm(decoder_residual, decoder_output->wrap<T>(),
                                                 &w->self_attention_weight.output, w->ffn_norm_weight.gamma, rmsnorm_eps);
            TensorMap ffn_inputs{{"ffn_input", decoder_output}};
            TensorMap ffn_outputs{{"ffn_output", decoder_output}};
            LlamaAttentionDynamicParams ffn_params = *dynamic_params;
            ffn_params.num_tokens = 0;  // decode: rows = batch_size (ffn.cpp:84-88)
            ffn->forward(&ffn_inputs, &ffn_outputs, &w->ffn_weight, &ffn_params);
            launchAddResidual(decoder_residual, decoder_output->wrap<T>(), true);
            self_attention_inputs.insert({"attention_input", decoder_output});
        }
    }
};

// ------------------------------------------------------------------------------------------------
template <typename T> class LlamaContextAttentionLayer {
private:
    int head_num, head_size, hidden_units, repeats_per_kv, kv_head_num;
    float scale;
    LlamaAttentionStaticParams *attention_static_params;
    hipStream_t stream;
    BaseAllocator *allocator;
    CublasWrapper *cublas_wrapper;
    llmie_api::Scratch<T> s_qkv, s_padded, s_kv, s_qk, s_qkv_out, s_unpadded;
    std::unique_ptr<TensorWrapper<T>> padded_k_view, padded_v_view, v_cache_view;
    TensorWrapper<T> *lineared_qkv = nullptr, *padded_q = nullptr, *padded_k = nullptr, *padded_v = nullptr;
    TensorWrapper<T> *k_cache = nullptr, *v_cache = nullptr, *qkT = nullptr, *padded_qkTv = nullptr;
    TensorWrapper<T> *transposed_unpadded_qkv = nullptr;

public:
    LlamaContextAttentionLayer(int head_num, int kv_head_num, int head_size, LlamaAttentionStaticParams *attention_params,
                               hipStream_t stream, CublasWrapper *cublas_wrapper, BaseAllocator *allocator)
        : head_num(head_num), head_size(head_size), hidden_units(head_num * head_size),
          repeats_per_kv(kv_head_num ? head_num / kv_head_num : 0), kv_head_num(kv_head_num),
          scale(1.0f / std::sqrt(static_cast<float>(head_size))), attention_static_params(attention_params),
          stream(stream), allocator(allocator), cublas_wrapper(cublas_wrapper), s_qkv(allocator), s_padded(allocator),
          s_kv(allocator), s_qk(allocator), s_qkv_out(allocator), s_unpadded(allocator) {
        LLM_CHECK_WITH_INFO(kv_head_num > 0 && head_num % kv_head_num == 0, "kv_head_num must be a factor of head_num");
        attention_static_params->head_num = head_num;
        attention_static_params->kv_head_num = kv_head_num;
        attention_static_params->head_size = head_size;
    }
    LlamaAttentionStaticParams *getAttentionStaticParams() { return attention_static_params; }

    // context_attention.cpp:29-90 (same buffers; q|k|v padded and k|v repeated share one slab each)
    void allocateMemory(LlamaAttentionDynamicParams *p) {
        const int bs = p->batch_size, T_ = p->num_tokens, mq = p->max_q_len, mk = p->max_k_len;
        const int qkv_heads = head_num + 2 * kv_head_num;
        const DataType type = getTensorType<T>();
        lineared_qkv = s_qkv.get({T_, qkv_heads, head_size});
        padded_q = s_padded.get({bs, qkv_heads, mq, head_size});
        padded_q->shape = {bs, head_num, mq, head_size};
        T *kptr = padded_q->data + static_cast<size_t>(bs) * mq * head_num * head_size;
        padded_k_view.reset(new TensorWrapper<T>(Device::GPU, type, {bs, kv_head_num, mq, head_size}, kptr));
        padded_v_view.reset(new TensorWrapper<T>(Device::GPU, type, {bs, kv_head_num, mq, head_size},
                                                 kptr + static_cast<size_t>(bs) * mq * kv_head_num * head_size));
        padded_k = padded_k_view.get();
        padded_v = padded_v_view.get();
        k_cache = s_kv.get({2 * bs, head_num, mk, head_size});
        k_cache->shape = {bs, head_num, mk, head_size};
        v_cache_view.reset(new TensorWrapper<T>(Device::GPU, type, {bs, head_num, mk, head_size},
                                                k_cache->data + static_cast<size_t>(bs) * head_num * mk * head_size));
        v_cache = v_cache_view.get();
        qkT = s_qk.get({bs, head_num, mq, mk});
        padded_qkTv = s_qkv_out.get({bs, head_num, mq, head_size});
        transposed_unpadded_qkv = s_unpadded.get({T_, head_num, head_size});
    }
    void freeBuf() {
        s_qkv.release(); s_padded.release(); s_kv.release(); s_qk.release(); s_qkv_out.release(); s_unpadded.release();
    }
    // context_attention.cpp:143-312
    void forward(TensorMap *inputs, TensorMap *outputs, LlamaAttentionWeights<T> *weights,
                 LlamaAttentionDynamicParams *dynamic_params, LlamaAttentionStaticParams *static_params) {
        allocateMemory(dynamic_params);
        Tensor *attention_input = inputs->at("attention_input");
        launchLinearGemm(attention_input->wrap<T>(), &weights->qkv, lineared_qkv, cublas_wrapper, false,
                         weights->qkv.is_transposed);
        Tensor *padding_offset = inputs->at("padding_offset");
        Tensor *history_length = inputs->at("history_length");
        Tensor *input_length = inputs->at("input_length");
        Tensor *layer_id = inputs->at("layer_id");
        // padded slots of q/k/v are never written by the scatter: zero them so QK^T stays finite
        CHECK(hipMemsetAsync(padded_q->data, 0,
                             sizeof(T) * static_cast<size_t>(dynamic_params->batch_size) * (head_num + 2 * kv_head_num) *
                                 dynamic_params->max_q_len * head_size, llmie_api::st()));
        launchFusedQKVAddBiasAndTransposeAndRope(padded_q, padded_k, padded_v, lineared_qkv, &weights->qkv,
                                                 padding_offset->wrap<int>(), history_length->wrap<int>(),
                                                 input_length->wrap<int>(), static_params);
        Tensor *all_k_cache = outputs->at("all_k_cache");
        Tensor *all_v_cache = outputs->at("all_v_cache");
        launchConcatKVCache(padded_k, padded_v, layer_id->wrap<int>(), input_length->wrap<int>(),
                            history_length->wrap<int>(), all_k_cache->wrap<T>(), all_v_cache->wrap<T>());
        Tensor *context_length = inputs->at("context_length");
        CHECK(hipMemsetAsync(k_cache->data, 0,
                             sizeof(T) * 2 * static_cast<size_t>(dynamic_params->batch_size) * head_num *
                                 dynamic_params->max_k_len * head_size, llmie_api::st()));
        launchRepeatKVCache(all_k_cache->wrap<T>(), all_v_cache->wrap<T>(), context_length->wrap<int>(),
                            layer_id->wrap<int>(), k_cache, v_cache);
        launchLinearStridedBatchGemm(padded_q, k_cache, qkT, cublas_wrapper, false, true);
        Tensor *attention_mask = inputs->at("attention_mask");
        launchFusedScaleMaskAndSoftmax(qkT, attention_mask->wrap<T>(), qkT, scale);
        launchLinearStridedBatchGemm(qkT, v_cache, padded_qkTv, cublas_wrapper, false, false);
        launchFusedTransposeAndRemovePadding(padded_qkTv, padding_offset->wrap<int>(), transposed_unpadded_qkv);
        Tensor *attention_output = outputs->at("attention_output");
        launchLinearGemm(transposed_unpadded_qkv, &weights->output, attention_output->wrap<T>(), cublas_wrapper, false,
                         weights->output.is_transposed);
    }
};

// ------------------------------------------------------------------------------------------------
template <typename T> class LlamaContextDecoder {
private:
    int head_num, kv_head_num, head_size, intermediate_size, num_layer, hidden_units;
    float rmsnorm_eps;
    hipStream_t stream;
    CublasWrapper *cublas_wrapper;
    BaseAllocator *allocator;
    llmie_api::Scratch<T> s_residual, s_mask;
    llmie_api::Scratch<int> s_padding, s_cum;
    TensorWrapper<T> *attention_mask = nullptr, *decoder_residual = nullptr;
    TensorWrapper<int> *padding_offset = nullptr, *cum_seqlens = nullptr;
    LlamaContextAttentionLayer<T> *context_attention = nullptr;
    LlamaFFNLayer<T> *ffn = nullptr;
    DataType data_type;
    EngineHolder<T> engine_holder;

public:
    // set false to force the reference's kernel sequence (padded q/k/v, batched GEMMs, softmax) even when the
    // flash-attention engine path (fp16, head_size 128, HF-layout weights) is eligible
    bool use_fused_engine = true;

    LlamaContextDecoder(const int &head_num, const int &kv_head_num, const int &head_size, const int &intermediate_size,
                        const int &num_layer, LlamaAttentionStaticParams *const &attention_static_params,
                        const float &rmsnorm_eps, const hipStream_t &stream, CublasWrapper *const &cublas_wrapper,
                        BaseAllocator *const &allocator)
        : head_num(head_num), kv_head_num(kv_head_num), head_size(head_size), intermediate_size(intermediate_size),
          num_layer(num_layer), hidden_units(head_num * head_size), rmsnorm_eps(rmsnorm_eps), stream(stream),
          cublas_wrapper(cublas_wrapper), allocator(allocator), s_residual(allocator), s_mask(allocator),
          s_padding(allocator), s_cum(allocator), data_type(getTensorType<T>()) {
        context_attention = new LlamaContextAttentionLayer<T>(head_num, kv_head_num, head_size, attention_static_params,
                                                              stream, cublas_wrapper, allocator);
        ffn = new LlamaFFNLayer<T>(head_num, head_size, intermediate_size, stream, cublas_wrapper, allocator);
    }
    ~LlamaContextDecoder() {
        delete context_attention;
        delete ffn;
    }
    LlamaContextDecoder(const LlamaContextDecoder &) = delete;
    LlamaContextDecoder &operator=(const LlamaContextDecoder &) = delete;

    void allocateMemory(LlamaAttentionDynamicParams *p) {
        decoder_residual = s_residual.get({p->num_tokens, hidden_units});
        attention_mask = s_mask.get({p->batch_size, p->max_q_len, p->max_k_len});
        padding_offset = s_padding.get({p->batch_size, p->max_q_len});
        cum_seqlens = s_cum.get({p->batch_size + 1});
    }
    // Unlike the reference (context_decoder.cpp:34-56) this does not delete the sub-layers: the decoder is reusable.
    void freeBuf() {
        s_residual.release(); s_mask.release(); s_padding.release(); s_cum.release();
    }
    // context_decoder.cpp:58-199
    void forward(TensorMap *input_tensors, std::vector<LlamaLayerWeight<T> *> *layer_weights, TensorMap *output_tensors,
                 LlamaAttentionDynamicParams *attention_dynamic_params) {
        if (use_fused_engine && std::is_same<T, half>::value && head_size == 128 && EngineHolder<T>::usable(layer_weights)) {
            // engine prefill: packed tokens, RoPE + KV append + flash attention (include/llmie.h llmie_decoder_prefill)
            Tensor *seq_lens = input_tensors->at("input_length");
            Tensor *history_length = input_tensors->at("history_length");
            Tensor *decoder_input = input_tensors->at("decoder_input");
            Tensor *decoder_output = output_tensors->at("decoder_output");
            Tensor *all_k_cache = output_tensors->at("all_k_cache");
            Tensor *all_v_cache = output_tensors->at("all_v_cache");
            const int batch = attention_dynamic_params->batch_size, tokens = attention_dynamic_params->num_tokens;
            LlamaAttentionStaticParams *sp = context_attention->getAttentionStaticParams();
            llmie_decoder *engine = engine_holder.get(layer_weights, num_layer, head_num, kv_head_num, head_size,
                                                      intermediate_size, *sp, rmsnorm_eps, batch, all_k_cache->shape[3]);
            llmie_decoder_config cfg{};
            cfg.head_num = head_num; cfg.kv_head_num = kv_head_num; cfg.head_size = head_size;
            cfg.inter_size = intermediate_size; cfg.num_layers = num_layer; cfg.max_seq_len = all_k_cache->shape[3];
            cfg.max_batch = batch; cfg.rotary_dim = sp->rotary_embedding_dim; cfg.rotary_base = sp->rotary_embedding_base;
            cfg.rms_eps = rmsnorm_eps; cfg.dtype = LLMIE_F16; cfg.wfmt = LLMIE_W_F16; cfg.int4_group = 128;
            size_t bytes = 0;
            void *pws = engine_holder.prefill_workspace(&cfg, tokens, batch, &bytes);
            LLMIE_CALL(llmie_decoder_prefill(engine, decoder_input->wrap<T>()->data, decoder_output->wrap<T>()->data,
                                             all_k_cache->wrap<T>()->data, all_v_cache->wrap<T>()->data,
                                             seq_lens->wrap<int>()->data, history_length->wrap<int>()->data, batch, tokens,
                                             attention_dynamic_params->max_q_len, pws, bytes, llmie_api::st()));
            return;
        }
        allocateMemory(attention_dynamic_params);
        Tensor *seq_lens = input_tensors->at("input_length");
        launchCalPaddingOffset(padding_offset, cum_seqlens, seq_lens->wrap<int>());
        Tensor *context_length = input_tensors->at("context_length");
        launchBuildCausalMasks<T>(attention_mask, seq_lens->wrap<int>(), context_length->wrap<int>());
        Tensor *history_length = input_tensors->at("history_length");
        Tensor *decoder_output = output_tensors->at("decoder_output");
        Tensor *all_k_cache = output_tensors->at("all_k_cache");
        Tensor *all_v_cache = output_tensors->at("all_v_cache");
        Tensor *decoder_input = input_tensors->at("decoder_input");
        LLM_CHECK_WITH_INFO(decoder_input->wrap<T>()->data != nullptr, "The data pointer of tensor inserted into TensorMap is nullptr!");
        LLM_CHECK_WITH_INFO(history_length->wrap<int>()->data != nullptr, "The data pointer of tensor inserted into TensorMap is nullptr!");
        int cur_layer = 0;
        TensorWrapper<int> layer_tensor(Device::CPU, getTensorType<int>(), {1}, &cur_layer);
        TensorMap context_attention_inputs{{"attention_input", decoder_input}, {"padding_offset", padding_offset},
                                           {"history_length", history_length}, {"input_length", seq_lens},
                                           {"context_length", context_length}, {"attention_mask", attention_mask},
                                           {"layer_id", &layer_tensor}};
        TensorMap context_attention_outputs{{"attention_output", decoder_output}, {"all_k_cache", all_k_cache},
                                            {"all_v_cache", all_v_cache}};
        for (cur_layer = 0; cur_layer < num_layer; ++cur_layer) {
            LlamaLayerWeight<T> *w = layer_weights->at(cur_layer);
            decoder_input = context_attention_inputs.at("attention_input");
            launchRMSNorm(decoder_input->wrap<T>(), decoder_residual, &w->attention_norm_weight, rmsnorm_eps);
            context_attention->forward(&context_attention_inputs, &context_attention_outputs, &w->self_attention_weight,
                                       attention_dynamic_params, context_attention->getAttentionStaticParams());
            launchFusedAddBiasResidualAndRMSNorm(decoder_residual, decoder_output->wrap<T>(),
                                                 &w->self_attention_weight.output, w->ffn_norm_weight.gamma, rmsnorm_eps);
            TensorMap ffn_inputs{{"ffn_input", decoder_output}};
            TensorMap ffn_outputs{{"ffn_output", decoder_output}};
            attention_dynamic_params->is_context = true;
            ffn->forward(&ffn_inputs, &ffn_outputs, &w->ffn_weight, attention_dynamic_params);
            launchAddResidual(decoder_residual, decoder_output->wrap<T>());
            context_attention_inputs.insert({"attention_input", decoder_output});
        }
    }
};

// C++ API mirror, part 8: BaseModel (src/models/basemodel.h:14-61), LlamaModel<T>
// (src/models/llama/llama.h:14-214, llama.cpp) and llm::create*LLMModel (src/utils/model_utils.h:16-94),
// i.e. everything user_entry.cpp touches.  The reference's model layer never compiled;
// it is used as the specification of names and control flow: prefill once through the context decoder,
// then <= output_token_limit decode steps, LM head on the last token, top-k, sampling, callback.
#pragma once
#include <functional>

#include "layers.hpp"
#include "tokenizer.hpp"

using CallBack = std::function<void(int, const char *)>;

class BaseModel {
public:
    std::string model_name;
    hipStream_t stream;
    CublasWrapper *cublas_wrapper;
    BaseAllocator *allocator;
    hipDeviceProp_t *device_prop;

    BaseModel(hipStream_t stream, CublasWrapper *cublas_wrapper, BaseAllocator *allocator,
              hipDeviceProp_t *device_prop = nullptr)
        : stream(stream), cublas_wrapper(cublas_wrapper), allocator(allocator), device_prop(device_prop) {}
    virtual ~BaseModel() = default;

    virtual void loadTokenizer(const std::string &file) = 0;
    virtual void loadWeights(const std::string &file) = 0;
    virtual void loadWeightsFromDummy() = 0;
    virtual std::vector<std::string> makeInput(const std::string &history, int round, const std::string &input) const = 0;
    virtual std::string makeHistory(const std::string &history, int round, const std::string &input,
                                    const std::string &output) const = 0;
    virtual std::string response(const std::vector<std::string> &input, CallBack printRes) = 0;

    // user_entry.cpp:25,39 spells these with capitals (basemodel.h:36-60 with lower case): both exist.
    std::vector<std::string> MakeInput(const std::string &h, int r, const std::string &i) const { return makeInput(h, r, i); }
    std::string MakeHistory(const std::string &h, int r, const std::string &i, const std::string &o) const {
        return makeHistory(h, r, i, o);
    }
    std::string Response(const std::vector<std::string> &input, CallBack printRes) { return response(input, printRes); }
};

template <typename T> class LlamaModel : public BaseModel {
private:
    int head_num, kv_head_num, head_size, inter_size, num_layers, vocab_size, vocab_size_padded;
    float rmsnorm_eps = 1e-5f;  // llama.h:23
    int hidden_units, max_seq_len;
    int pad_token_id = 0, bos_token_id = 1, eos_token_id = 2;
    int layer_id = 0, batch_size = 1, beamwidth = 1, blocks_per_beam = 8, K = 4;
    std::string prompt;
    Tokenizer tokenizer;
    LlamaAttentionStaticParams static_params;
    std::unique_ptr<CublasWrapper> owned_cublas;
    std::unique_ptr<BaseAllocator> owned_allocator;
    std::unique_ptr<LlamaWeight<T>> llama_weights;
    std::unique_ptr<LlamaContextDecoder<T>> context_decoder;
    std::unique_ptr<LlamaSelfDecoder<T>> self_decoder;
    std::vector<LlamaLayerWeight<T> *> layer_ptrs;
    MapStringToInt int_params_of_sample;
    int h_step = 0;

    template <typename U> struct DevBuf {
        U *p = nullptr;
        size_t n = 0;
        ~DevBuf() { if (p) (void)hipFree(p); }
        U *ensure(size_t count) {
            if (count > n) {
                if (p) CHECK(hipFree(p));
                CHECK(hipMalloc(reinterpret_cast<void **>(&p), sizeof(U) * count));
                n = count;
            }
            return p;
        }
    };
    DevBuf<T> d_ctx_in, d_ctx_out, d_dec_in, d_dec_out, d_kcache, d_vcache, d_probs, d_topk_val, d_final_val, d_unused;
    DevBuf<int> d_ids, d_in_len, d_hist_len, d_ctx_len, d_seq_len, d_token, d_topk_id, d_final_id;
    DevBuf<bool> d_finished;

    int lmHeadAndSample(T *hidden_row /*[1,H] device*/) {
        const DataType ty = getTensorType<T>(), ti = getTensorType<int>();
        TensorWrapper<T> x(Device::GPU, ty, {batch_size, hidden_units}, hidden_row);
        TensorWrapper<T> unused(Device::GPU, ty, {batch_size, hidden_units}, d_unused.ensure(hidden_units));
        launchRMSNorm(&x, &unused, &llama_weights->out_rmsnorm_weight, rmsnorm_eps, true);      // llama.cpp:247
        TensorWrapper<T> probs(Device::GPU, ty, {batch_size, vocab_size}, d_probs.ensure(vocab_size));
        launchLinearGemm(&x, &llama_weights->post_decoder_embedding_weight, &probs, cublas_wrapper, false, true);  // :282
        TensorWrapper<int> topk_id(Device::GPU, ti, {batch_size, beamwidth, blocks_per_beam, K}, d_topk_id.ensure(blocks_per_beam * K));
        TensorWrapper<T> topk_val(Device::GPU, ty, {batch_size, beamwidth, blocks_per_beam, K}, d_topk_val.ensure(blocks_per_beam * K));
        TensorWrapper<int> final_id(Device::GPU, ti, {batch_size * beamwidth, K}, d_final_id.ensure(K));
        TensorWrapper<T> final_val(Device::GPU, ty, {batch_size * beamwidth, K}, d_final_val.ensure(K));
        launchTopKForBeamSearch(&probs, &topk_id, &topk_val, &final_id, &final_val);                // :293
        int_params_of_sample["step"] = h_step;
        TensorWrapper<int> seq(Device::GPU, ti, {batch_size}, d_seq_len.ensure(1));
        TensorWrapper<bool> fin(Device::GPU, getTensorType<bool>(), {batch_size}, d_finished.ensure(1));
        TensorWrapper<int> tok(Device::GPU, ti, {batch_size}, d_token.ensure(1));
        launchSampling(&final_id, &final_val, &seq, &fin, &tok, &int_params_of_sample);            // :304
        int h_tok = 0;
        CHECK(hipMemcpyAsync(&h_tok, tok.data, sizeof(int), hipMemcpyDeviceToHost, llmie_api::st()));  // :314
        CHECK(hipStreamSynchronize(llmie_api::st()));
        return h_tok;
    }

public:
    int output_token_limit = 20;  // llama.h:26

    LlamaModel(int head_num, int kv_head_num, int head_size, int inter_size, int num_layers, int vocab_size,
               const LlamaAttentionStaticParams &attention_static_params, int max_seq_len, hipStream_t stream,
               CublasWrapper *cublas_wrapper, BaseAllocator *allocator, hipDeviceProp_t *device_prop = nullptr)
        : BaseModel(stream, cublas_wrapper, allocator, device_prop), head_num(head_num), kv_head_num(kv_head_num),
          head_size(head_size), inter_size(inter_size), num_layers(num_layers), vocab_size(vocab_size),
          vocab_size_padded(vocab_size), hidden_units(head_num * head_size), max_seq_len(max_seq_len),
          static_params(attention_static_params) {
        model_name = "llama";
        int_params_of_sample["vocab_size"] = vocab_size;
        int_params_of_sample["end_id"] = eos_token_id;
        llama_weights = std::make_unique<LlamaWeight<T>>(head_num, kv_head_num, head_size, inter_size, vocab_size,
                                                         num_layers, false, getWeightType<T>());
        layer_ptrs = llama_weights->layerPointers();
        self_decoder = std::make_unique<LlamaSelfDecoder<T>>(head_num, kv_head_num, head_size, inter_size, num_layers,
                                                             static_params, rmsnorm_eps, stream, cublas_wrapper, allocator);
        context_decoder = std::make_unique<LlamaContextDecoder<T>>(head_num, kv_head_num, head_size, inter_size,
                                                                   num_layers, &static_params, rmsnorm_eps, stream,
                                                                   cublas_wrapper, allocator);
        const size_t kv = static_cast<size_t>(num_layers) * batch_size * kv_head_num * max_seq_len * head_size;
        CHECK(hipMemset(d_kcache.ensure(kv), 0, sizeof(T) * kv));
        CHECK(hipMemset(d_vcache.ensure(kv), 0, sizeof(T) * kv));
    }
    // model_utils.h creates the GEMM context and allocator on the stack and releases the model: keep them alive here
    void adopt(std::unique_ptr<CublasWrapper> c, std::unique_ptr<BaseAllocator> a) {
        owned_cublas = std::move(c);
        owned_allocator = std::move(a);
    }
    void loadTokenizer(const std::string &file) override { tokenizer.Initialize(file); }
    void loadWeights(const std::string &dir) override { llama_weights->loadWeightsFromFile(dir); }
    void loadWeightsFromDummy() override { llama_weights->loadWeightsFromDummy(); }
    LlamaWeight<T> *weights() { return llama_weights.get(); }

    // llama.cpp:128-150
    std::vector<std::string> makeInput(const std::string &history, int round, const std::string &input) const override {
        return {(round == 0 ? "" : history) + input, history, input};
    }
    std::string makeHistory(const std::string &history, int round, const std::string &input,
                            const std::string &output) const override {
        return (round == 0 ? prompt : history) + input + output;
    }

    // llama.cpp:165-217: prefill of `ids` on top of `history_len` cached tokens; returns the first new token
    int generateFirstToken(const std::vector<int> &ids, int history_len) {
        const int n = static_cast<int>(ids.size());
        LLM_CHECK_WITH_INFO(n > 0 && history_len + n < max_seq_len, "prompt does not fit max_seq_len");
        const DataType ty = getTensorType<T>(), ti = getTensorType<int>();
        const int ctx = history_len + n;
        CHECK(hipMemcpy(d_ids.ensure(n), ids.data(), sizeof(int) * n, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_in_len.ensure(1), &n, sizeof(int), hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_hist_len.ensure(1), &history_len, sizeof(int), hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_ctx_len.ensure(1), &ctx, sizeof(int), hipMemcpyHostToDevice));
        bool f = false;
        CHECK(hipMemcpy(d_finished.ensure(1), &f, sizeof(bool), hipMemcpyHostToDevice));
        CHECK(hipMemset(d_seq_len.ensure(1), 0, sizeof(int)));
        TensorWrapper<int> input_ids(Device::GPU, ti, {n}, d_ids.p);
        TensorWrapper<T> ctx_in(Device::GPU, ty, {n, hidden_units}, d_ctx_in.ensure(static_cast<size_t>(n) * hidden_units));
        TensorWrapper<T> ctx_out(Device::GPU, ty, {n, hidden_units}, d_ctx_out.ensure(static_cast<size_t>(n) * hidden_units));
        launchInputEmbedding<T>(&input_ids, &ctx_in, &llama_weights->pre_decoder_embedding_weight);
        TensorWrapper<int> in_len(Device::GPU, ti, {1}, d_in_len.p), hist(Device::GPU, ti, {1}, d_hist_len.p);
        TensorWrapper<int> ctx_len(Device::GPU, ti, {1}, d_ctx_len.p);
        TensorWrapper<int> layer(Device::CPU, ti, {1}, &layer_id);
        TensorWrapper<T> kc(Device::GPU, ty, {num_layers, batch_size, kv_head_num, max_seq_len, head_size}, d_kcache.p);
        TensorWrapper<T> vc(Device::GPU, ty, {num_layers, batch_size, kv_head_num, max_seq_len, head_size}, d_vcache.p);
        TensorMap decoder_inputs{{"decoder_input", &ctx_in}, {"history_length", &hist}, {"input_length", &in_len},
                                 {"context_length", &ctx_len}, {"layer_id", &layer}};
        TensorMap decoder_outputs{{"decoder_output", &ctx_out}, {"all_k_cache", &kc}, {"all_v_cache", &vc}};
        LlamaAttentionDynamicParams dyn{};
        dyn.batch_size = 1;
        dyn.num_tokens = n;
        dyn.max_q_len = n;
        dyn.max_k_len = ctx;
        dyn.num_layers = num_layers;
        context_decoder->forward(&decoder_inputs, &layer_ptrs, &decoder_outputs, &dyn);
        h_step = ctx;
        return lmHeadAndSample(ctx_out.data + static_cast<size_t>(n - 1) * hidden_units);  // last token only (:262-279)
    }

    // llama.cpp:219-257: one decode step for token `id`; h_step = context length including it
    int generateNextToken(int id) {
        const DataType ty = getTensorType<T>(), ti = getTensorType<int>();
        CHECK(hipMemcpy(d_ids.ensure(1), &id, sizeof(int), hipMemcpyHostToDevice));
        TensorWrapper<int> input_ids(Device::GPU, ti, {1}, d_ids.p);
        TensorWrapper<T> dec_in(Device::GPU, ty, {1, hidden_units}, d_dec_in.ensure(hidden_units));
        TensorWrapper<T> dec_out(Device::GPU, ty, {1, hidden_units}, d_dec_out.ensure(hidden_units));
        launchInputEmbedding<T>(&input_ids, &dec_in, &llama_weights->pre_decoder_embedding_weight);
        TensorWrapper<int> step(Device::CPU, ti, {1}, &h_step);
        TensorWrapper<int> layer(Device::CPU, ti, {1}, &layer_id);
        TensorWrapper<bool> fin(Device::GPU, getTensorType<bool>(), {1}, d_finished.ensure(1));
        TensorWrapper<T> kc(Device::GPU, ty, {num_layers, batch_size, kv_head_num, max_seq_len, head_size}, d_kcache.p);
        TensorWrapper<T> vc(Device::GPU, ty, {num_layers, batch_size, kv_head_num, max_seq_len, head_size}, d_vcache.p);
        TensorMap decoder_inputs{{"decoder_input", &dec_in}, {"step", &step}, {"finished", &fin}, {"layer_id", &layer}};
        TensorMap decoder_outputs{{"decoder_output", &dec_out}, {"all_k_cache", &kc}, {"all_v_cache", &vc}};
        LlamaAttentionDynamicParams dyn{};
        dyn.batch_size = 1;
        dyn.num_layers = num_layers;
        self_decoder->forward(&decoder_inputs, &layer_ptrs, &decoder_outputs, &dyn);
        return lmHeadAndSample(dec_out.data);
    }

    // llama.cpp:322-398.  Returns the generated text; printRes(index, piece), index -1 = end of reply.
    std::vector<int> last_token_ids;
    const Tokenizer &getTokenizer() const { return tokenizer; }
    std::string response(const std::vector<std::string> &input, CallBack printRes) override {
        const std::vector<int> history_ids = input.size() > 1 && !input[1].empty() ? tokenizer.Encode(input[1]) : std::vector<int>();
        const std::vector<int> cur_ids = tokenizer.Encode(input.empty() ? std::string() : input[0]);
        (void)history_ids;  // batch-1, single round: the whole context is re-prefilled, as the reference does
        last_token_ids.clear();
        std::string ret_string;
        int ret = 0;
        for (int index = 0; index < output_token_limit; ++index) {
            if (index == 0) {
                ret = generateFirstToken(cur_ids, 0);
            } else {
                if (h_step + 1 >= max_seq_len) break;
                ++h_step;  // the token generated last round becomes part of the context
                ret = generateNextToken(ret);
                if (ret == eos_token_id) break;
            }
            last_token_ids.push_back(ret);
            const std::string piece = tokenizer.Decode({ret});
            ret_string += piece;
            if (printRes) printRes(index, piece.c_str());
        }
        if (printRes) printRes(-1, ret_string.c_str());
        return ret_string;
    }
};

namespace llm {
// Replaces the JSON file read from a hard-coded home path (model_utils.h:22-40, llama_config.json):
// Llama-2-7B defaults, editable before the first create* call.
struct ModelConfig {
    int head_num = 32, kv_head_num = 32, head_size = 128, inter_size = 11008, num_layers = 32, max_seq_len = 2048;
    int vocab_size = 32000;
    int rotary_embedding_dim = 128, max_position_embeddings = 2048;
    float rotary_embedding_base = 10000.0f;
    bool use_dynamic_ntk = false, attn_bias = false;
};
inline ModelConfig &config() {
    static ModelConfig c;
    return c;
}

template <typename T> BaseModel *createModelWithName(const std::string &model_name) {
    LLM_CHECK_WITH_INFO(model_name == "llama", "Currently, only llama models are supported!");
    const ModelConfig &c = config();
    LlamaAttentionStaticParams sp{};
    sp.rotary_embedding_dim = c.rotary_embedding_dim;
    sp.rotary_embedding_base = c.rotary_embedding_base;
    sp.max_position_embeddings = c.max_position_embeddings;
    sp.use_dynamic_ntk = c.use_dynamic_ntk;
    auto cublas_wrapper = std::make_unique<CublasWrapper>(nullptr, nullptr);
    if (std::is_same<T, half>::value) cublas_wrapper->setFP16GemmConfig();
    else cublas_wrapper->setFP32GemmConfig();
    std::unique_ptr<BaseAllocator> allocator = std::make_unique<CudaAllocator>();
    static hipDeviceProp_t device_prop;
    CHECK(hipGetDeviceProperties(&device_prop, 0));
    auto model = std::make_unique<LlamaModel<T>>(c.head_num, c.kv_head_num, c.head_size, c.inter_size, c.num_layers,
                                                 c.vocab_size, sp, c.max_seq_len, nullptr, cublas_wrapper.get(),
                                                 allocator.get(), &device_prop);
    model->adopt(std::move(cublas_wrapper), std::move(allocator));
    return model.release();
}
template <typename T> BaseModel *createDummyLLMModel(const std::string &tokenizer_file) {
    auto model = std::unique_ptr<BaseModel>(createModelWithName<T>("llama"));
    model->loadTokenizer(tokenizer_file);
    model->loadWeightsFromDummy();
    return model.release();
}
template <typename T> BaseModel *createRealLLMModel(const std::string &model_dir, const std::string &tokenizer_file) {
    auto model = std::unique_ptr<BaseModel>(createModelWithName<T>("llama"));
    std::cout << "Start creating model..." << std::endl;
    model->loadTokenizer(tokenizer_file);
    model->loadWeights(model_dir);
    std::cout << "Finish creating model..." << std::endl;
    return model.release();
}
}  // namespace llm

// C++ API mirror, part 4: weight structs and owners
// (src/weights/includes/*.h, src/weights/layer_weights.cpp, llama_weights.cpp, src/utils/weight_utils.cu).
// Layout = reference/HF: qkv [(nh+2kvh)*hs, H], o [H,H], gate_and_up [2I,H], down [H,I], embeddings [V,H].
#pragma once
#include <cstdint>
#include <fstream>
#include <type_traits>

#include "runtime.hpp"

enum class WeightType { FP32_W, FP16_W, INT8_W, UNSUPPORTED_W };

template <typename T> inline WeightType getWeightType() {
    using U = typename std::remove_const<T>::type;
    if (std::is_same<U, float>::value) return WeightType::FP32_W;
    if (std::is_same<U, half>::value) return WeightType::FP16_W;
    if (std::is_same<U, int8_t>::value) return WeightType::INT8_W;
    return WeightType::UNSUPPORTED_W;
}

template <typename T> struct BaseWeight {
    WeightType type = getWeightType<T>();
    std::vector<int> shape;
    T *data = nullptr, *bias = nullptr;
    bool is_transposed = false;
};
template <typename T> struct LayerNormWeight { T *gamma = nullptr; };
template <typename T> struct EmbeddingWeight : public BaseWeight<T> {};
template <typename T> struct LlamaAttentionWeights {
    BaseWeight<T> qkv;
    BaseWeight<T> output;
};
template <typename T> struct LlamaFFNWeights {
    BaseWeight<T> gate;
    BaseWeight<T> up;
    BaseWeight<T> down;
    BaseWeight<T> gate_and_up;
};

class Weight {
public:
    virtual ~Weight() = default;
    virtual void loadWeightsFromFile(const std::string &weight_path) = 0;
};

// ---- device allocation helpers (weight_utils.h: GPUMalloc takes an ELEMENT count) ----
template <typename T> inline void GPUMalloc(T **ptr, size_t count) {
    LLM_CHECK_WITH_INFO(count > 0, "GPUMalloc: count must be positive");
    CHECK(hipMalloc(reinterpret_cast<void **>(ptr), sizeof(T) * count));
}
template <typename T> inline void GPUFree(T *ptr) {
    if (ptr) CHECK(hipFree(ptr));
}

namespace llmie_api {
template <typename T> inline T host_cast(float v);
template <> inline float host_cast<float>(float v) { return v; }
template <> inline half host_cast<half>(float v) { return __float2half(v); }

template <typename T> inline void upload(T *dst, const std::vector<T> &host) {
    CHECK(hipMemcpy(dst, host.data(), sizeof(T) * host.size(), hipMemcpyHostToDevice));
}
}  // namespace llmie_api

// raw per-tensor .bin reader with convert-on-load (weight_utils.cu:95-224): the file holds
// FileType elements; missing / short files print a message and leave the tensor untouched.
template <typename OutputType, typename FileType> class loadWeightFromBin {
public:
    static void loadFromFileToDevice(OutputType *ptr, const std::vector<int> &shape, const std::string &filename) {
        if (shape.empty() || shape.size() > 2) {
            std::cerr << "[ERROR] shape should have one or two dims\n";
            return;
        }
        const size_t n = static_cast<size_t>(shape[0]) * (shape.size() == 2 ? shape[1] : 1);
        if (n == 0) return;
        std::ifstream in(filename, std::ios::in | std::ios::binary);
        if (!in.is_open()) {
            std::cerr << "File " << filename << " cannot be opened, loading model fails!" << std::endl;
            return;
        }
        in.seekg(0, std::ios::end);
        const std::streamsize have = in.tellg();
        in.seekg(0, std::ios::beg);
        if (have < static_cast<std::streamsize>(sizeof(FileType) * n)) {
            std::cerr << "File " << filename << " is too small, expected " << sizeof(FileType) * n << " bytes but got "
                      << have << " bytes" << std::endl;
            return;
        }
        std::vector<FileType> host(n);
        in.read(reinterpret_cast<char *>(host.data()), sizeof(FileType) * n);
        if (!in) {
            std::cerr << "Error reading from file " << filename << std::endl;
            return;
        }
        if (std::is_same<OutputType, FileType>::value) {
            CHECK(hipMemcpy(ptr, host.data(), sizeof(FileType) * n, hipMemcpyHostToDevice));
        } else {
            std::vector<OutputType> conv(n);
            for (size_t i = 0; i < n; ++i) conv[i] = llmie_api::host_cast<OutputType>(static_cast<float>(host[i]));
            llmie_api::upload(ptr, conv);
        }
    }
};

template <typename T> class LlamaLayerWeight {
private:
    int head_num, kv_head_num, head_size, hidden_units, intermediate_size;
    WeightType weight_type;
    bool attention_bias;

public:
    LlamaLayerWeight() = delete;
    // layer_weights.cpp:6-47: allocates every tensor of one decoder layer
    LlamaLayerWeight(int head_num, int kv_head_num, int head_size, int intermediate_size, WeightType weight_type,
                     bool attention_bias)
        : head_num(head_num), kv_head_num(kv_head_num), head_size(head_size), hidden_units(head_num * head_size),
          intermediate_size(intermediate_size), weight_type(weight_type), attention_bias(attention_bias) {
        const int qkv_units = (head_num + 2 * kv_head_num) * head_size;
        GPUMalloc(&attention_norm_weight.gamma, hidden_units);
        GPUMalloc(&ffn_norm_weight.gamma, hidden_units);
        self_attention_weight.qkv.type = weight_type;
        self_attention_weight.qkv.shape = {qkv_units, hidden_units};
        GPUMalloc(&self_attention_weight.qkv.data, static_cast<size_t>(qkv_units) * hidden_units);
        self_attention_weight.output.type = weight_type;
        self_attention_weight.output.shape = {hidden_units, hidden_units};
        GPUMalloc(&self_attention_weight.output.data, static_cast<size_t>(hidden_units) * hidden_units);
        if (attention_bias) {
            GPUMalloc(&self_attention_weight.qkv.bias, qkv_units);
            GPUMalloc(&self_attention_weight.output.bias, hidden_units);
            GPUMalloc(&ffn_weight.down.bias, hidden_units);
        }
        ffn_weight.gate_and_up.type = weight_type;
        ffn_weight.down.type = weight_type;
        ffn_weight.gate_and_up.shape = {2 * intermediate_size, hidden_units};
        ffn_weight.down.shape = {hidden_units, intermediate_size};
        GPUMalloc(&ffn_weight.gate_and_up.data, static_cast<size_t>(2) * intermediate_size * hidden_units);
        GPUMalloc(&ffn_weight.down.data, static_cast<size_t>(hidden_units) * intermediate_size);
    }
    ~LlamaLayerWeight() {
        GPUFree(attention_norm_weight.gamma);
        GPUFree(ffn_norm_weight.gamma);
        freeWeights(&self_attention_weight.qkv);
        freeWeights(&self_attention_weight.output);
        freeWeights(&ffn_weight.gate_and_up);
        freeWeights(&ffn_weight.down);
    }
    LlamaLayerWeight(const LlamaLayerWeight &) = delete;
    LlamaLayerWeight &operator=(const LlamaLayerWeight &) = delete;

    // layer_weights.cpp:49-81: per-tensor fp32 .bin files, HF ([N,K]) layout => is_transposed = true
    void loadWeightsFromFile(const std::string &weight_path, WeightType) {
        const int qkv_units = (head_num + 2 * kv_head_num) * head_size;
        auto load = [&](const std::string &suffix, const std::vector<int> &shape, T *ptr) {
            loadWeightFromBin<T, float>::loadFromFileToDevice(ptr, shape, weight_path + suffix);
        };
        load(".input_layernorm.weight.bin", {hidden_units}, attention_norm_weight.gamma);
        load(".post_attention_layernorm.weight.bin", {hidden_units}, ffn_norm_weight.gamma);
        load(".self_attn.qkv.weight.bin", {qkv_units, hidden_units}, self_attention_weight.qkv.data);
        load(".self_attn.o_proj.weight.bin", {hidden_units, hidden_units}, self_attention_weight.output.data);
        load(".mlp.gate_up_proj.weight.bin", {2 * intermediate_size, hidden_units}, ffn_weight.gate_and_up.data);
        load(".mlp.down_proj.weight.bin", {hidden_units, intermediate_size}, ffn_weight.down.data);
        if (attention_bias) {
            load(".attention.wqkv.bias.bin", {qkv_units}, self_attention_weight.qkv.bias);
            load(".attention.wo.bias.bin", {hidden_units}, self_attention_weight.output.bias);
        }
        self_attention_weight.qkv.is_transposed = true;
        self_attention_weight.output.is_transposed = true;
        ffn_weight.gate_and_up.is_transposed = true;
        ffn_weight.down.is_transposed = true;
    }

    // layer_weights.cpp:83-156: dummy init.  Same value recipe (rand() % 10000 / 100000.f, glibc rand,
    // same fill order) and the same flags as the reference (output.is_transposed == false there).
    void loadWeightsFromFile() {
        const size_t qkv_n = static_cast<size_t>(hidden_units) * (head_num + 2 * kv_head_num) * head_size;
        const size_t o_n = static_cast<size_t>(hidden_units) * hidden_units;
        const size_t gu_n = static_cast<size_t>(hidden_units) * 2 * intermediate_size;
        const size_t down_n = static_cast<size_t>(hidden_units) * intermediate_size;
        auto fill = [](size_t n) {
            std::vector<T> v(n);
            for (size_t i = 0; i < n; ++i) v[i] = llmie_api::host_cast<T>(static_cast<float>(rand() % 10000 / 100000.0f));
            return v;
        };
        const std::vector<T> h_attn_norm = fill(hidden_units), h_ffn_norm = fill(hidden_units);
        const std::vector<T> h_o_bias = fill(hidden_units), h_down_bias = fill(hidden_units);
        const std::vector<T> h_down = fill(down_n), h_gu = fill(gu_n), h_o = fill(o_n), h_qkv = fill(qkv_n);
        llmie_api::upload(attention_norm_weight.gamma, h_attn_norm);
        llmie_api::upload(ffn_norm_weight.gamma, h_ffn_norm);
        llmie_api::upload(self_attention_weight.qkv.data, h_qkv);
        llmie_api::upload(self_attention_weight.output.data, h_o);
        if (!self_attention_weight.output.bias) GPUMalloc(&self_attention_weight.output.bias, hidden_units);
        llmie_api::upload(self_attention_weight.output.bias, h_o_bias);
        llmie_api::upload(ffn_weight.down.data, h_down);
        if (!ffn_weight.down.bias) GPUMalloc(&ffn_weight.down.bias, hidden_units);
        llmie_api::upload(ffn_weight.down.bias, h_down_bias);
        llmie_api::upload(ffn_weight.gate_and_up.data, h_gu);
        if (self_attention_weight.qkv.bias) {
            GPUFree(self_attention_weight.qkv.bias);
            self_attention_weight.qkv.bias = nullptr;  // the reference's dummy model has no qkv bias
        }
        self_attention_weight.qkv.is_transposed = true;
        self_attention_weight.output.is_transposed = false;
        ffn_weight.gate_and_up.is_transposed = true;
        ffn_weight.down.is_transposed = true;
    }

    void freeWeights(BaseWeight<T> *w) {
        GPUFree(w->data);
        GPUFree(w->bias);
        w->data = nullptr;
        w->bias = nullptr;
    }

    LayerNormWeight<T> attention_norm_weight;
    LayerNormWeight<T> ffn_norm_weight;
    LlamaAttentionWeights<T> self_attention_weight;
    LlamaFFNWeights<T> ffn_weight;
};

template <typename T> class LlamaWeight : public Weight {
private:
    int hidden_units = 0, intermediate_size = 0, vocab_size = 0, vocab_size_padded = 0, num_layer = 0;
    WeightType weight_type = WeightType::UNSUPPORTED_W;

public:
    std::vector<std::unique_ptr<LlamaLayerWeight<T>>> llama_layer_weight;
    LayerNormWeight<T> out_rmsnorm_weight;
    EmbeddingWeight<T> post_decoder_embedding_weight;  // lm_head [V,H]
    EmbeddingWeight<T> pre_decoder_embedding_weight;   // embed_tokens [V,H]

    LlamaWeight() = default;
    // llama_weights.cpp:6-46
    LlamaWeight(int head_num, int kv_head_num, int head_size, int intermediate_size, int vocab_size, int num_layer,
                bool attention_bias, WeightType weight_type)
        : hidden_units(head_num * head_size), intermediate_size(intermediate_size), vocab_size(vocab_size),
          vocab_size_padded(vocab_size), num_layer(num_layer), weight_type(weight_type) {
        llama_layer_weight.reserve(num_layer);
        for (int l = 0; l < num_layer; ++l)
            llama_layer_weight.push_back(std::make_unique<LlamaLayerWeight<T>>(head_num, kv_head_num, head_size,
                                                                              intermediate_size, weight_type, attention_bias));
        GPUMalloc(&out_rmsnorm_weight.gamma, hidden_units);
        GPUMalloc(&post_decoder_embedding_weight.data, static_cast<size_t>(vocab_size) * hidden_units);
        GPUMalloc(&pre_decoder_embedding_weight.data, static_cast<size_t>(vocab_size) * hidden_units);
        pre_decoder_embedding_weight.shape = {vocab_size, hidden_units};
        post_decoder_embedding_weight.shape = {vocab_size, hidden_units};
        pre_decoder_embedding_weight.type = weight_type;
        post_decoder_embedding_weight.type = weight_type;
        post_decoder_embedding_weight.is_transposed = true;
    }
    ~LlamaWeight() override {
        GPUFree(pre_decoder_embedding_weight.data);
        GPUFree(out_rmsnorm_weight.gamma);
        GPUFree(post_decoder_embedding_weight.data);
    }
    // llama_weights.cpp:48-74: file names of the converted checkpoint
    void loadWeightsFromFile(const std::string &weight_path) override {
        loadWeightFromBin<T, float>::loadFromFileToDevice(out_rmsnorm_weight.gamma, {hidden_units},
                                                          weight_path + "model.norm.weight.bin");
        loadWeightFromBin<T, float>::loadFromFileToDevice(post_decoder_embedding_weight.data, {vocab_size, hidden_units},
                                                          weight_path + "lm_head.weight.bin");
        loadWeightFromBin<T, float>::loadFromFileToDevice(pre_decoder_embedding_weight.data, {vocab_size, hidden_units},
                                                          weight_path + "model.embed_tokens.weight.bin");
        for (int l = 0; l < num_layer; ++l)
            llama_layer_weight[l]->loadWeightsFromFile(weight_path + "model.layers." + std::to_string(l), weight_type);
    }
    void loadWeights(const std::string &weight_path) { loadWeightsFromFile(weight_path); }
    // llama_weights.cpp:76-127: norm gamma and both embeddings = 1, layers = rand()-dummy
    void loadWeightsFromDummy() {
        llmie_api::upload(out_rmsnorm_weight.gamma, std::vector<T>(hidden_units, llmie_api::host_cast<T>(1.0f)));
        const std::vector<T> ones(static_cast<size_t>(hidden_units) * vocab_size, llmie_api::host_cast<T>(1.0f));
        llmie_api::upload(post_decoder_embedding_weight.data, ones);
        llmie_api::upload(pre_decoder_embedding_weight.data, ones);
        for (int l = 0; l < num_layer; ++l) llama_layer_weight[l]->loadWeightsFromFile();
    }
    std::vector<LlamaLayerWeight<T> *> layerPointers() const {
        std::vector<LlamaLayerWeight<T> *> v;
        for (const auto &p : llama_layer_weight) v.push_back(p.get());
        return v;
    }
};

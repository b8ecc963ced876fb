// C++ API mirror, part 3: parameter structs (src/models/llama/llama_params.h:3-21, src/utils/params.h:6-7).
#pragma once
#include <string>
#include <unordered_map>

struct LlamaAttentionStaticParams {
    int rotary_embedding_dim;
    float rotary_embedding_base;
    int max_position_embeddings;
    bool use_dynamic_ntk;  // placeholder in the reference as well
    int head_size = 128;
    int head_num = 32;
    int kv_head_num = 32;
};

// per-call shapes; not every field is needed by every layer
struct LlamaAttentionDynamicParams {
    int batch_size;
    int num_tokens;
    int max_q_len;
    int max_k_len;
    int num_layers;
    bool is_context = false;
};

using MapStringToInt = std::unordered_map<std::string, int>;
using MapStringToFloat = std::unordered_map<std::string, float>;

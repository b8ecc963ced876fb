// Shared by test_layers_api.cpp and test_examples_replay.cpp: a random Llama layer stack on host (oracle view) and device
// (LlamaLayerWeight view), and the oracle composition of LlamaContextDecoder::forward.
#pragma once
#include "../src/layers/includes/context_decoder.h"
#include "../src/layers/includes/self_decoder.h"
#include "../src/utils/model_utils.h"
#include <cstdlib>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>

#include "test_common.hpp"

struct HostLayer {
    std::vector<float> attn_norm, qkv, o, o_bias, ffn_norm, gate_up, down;
};

template <typename T> struct Model {
    int nh, kvh, hs, I, L, H, QKV;
    std::vector<HostLayer> host;
    std::vector<std::unique_ptr<LlamaLayerWeight<T>>> dev;
    std::vector<LlamaLayerWeight<T> *> ptrs;
    std::vector<orc_layer_weights> orc;

    Model(int nh_, int kvh_, int hs_, int I_, int L_, bool o_bias, bool hf_layout_flags, uint64_t seed)
        : nh(nh_), kvh(kvh_), hs(hs_), I(I_), L(L_), H(nh_ * hs_), QKV((nh_ + 2 * kvh_) * hs_) {
        std::mt19937_64 rng(seed);
        for (int l = 0; l < L; ++l) {
            HostLayer h;
            h.attn_norm = randu(rng, H, 0.2f);
            h.ffn_norm = randu(rng, H, 0.2f);
            for (auto &v : h.attn_norm) v += 1.f;
            for (auto &v : h.ffn_norm) v += 1.f;
            h.attn_norm = storage_round<T>(h.attn_norm);
            h.ffn_norm = storage_round<T>(h.ffn_norm);
            h.qkv = storage_round<T>(randu(rng, static_cast<size_t>(QKV) * H, 2.f / std::sqrt(static_cast<float>(H))));
            h.o = storage_round<T>(randu(rng, static_cast<size_t>(H) * H, 2.f / std::sqrt(static_cast<float>(H))));
            if (o_bias) h.o_bias = storage_round<T>(randu(rng, H, 0.1f));
            h.gate_up = storage_round<T>(randu(rng, static_cast<size_t>(2 * I) * H, 2.f / std::sqrt(static_cast<float>(H))));
            h.down = storage_round<T>(randu(rng, static_cast<size_t>(H) * I, 2.f / std::sqrt(static_cast<float>(I))));
            host.push_back(std::move(h));
        }
        for (int l = 0; l < L; ++l) {
            auto w = std::make_unique<LlamaLayerWeight<T>>(nh, kvh, hs, I, getWeightType<T>(), o_bias);
            const HostLayer &h = host[l];
            llmie_api::upload(w->attention_norm_weight.gamma, cast_vec<T>(h.attn_norm));
            llmie_api::upload(w->ffn_norm_weight.gamma, cast_vec<T>(h.ffn_norm));
            llmie_api::upload(w->self_attention_weight.qkv.data, cast_vec<T>(h.qkv));
            llmie_api::upload(w->self_attention_weight.output.data, cast_vec<T>(h.o));
            if (o_bias) llmie_api::upload(w->self_attention_weight.output.bias, cast_vec<T>(h.o_bias));
            if (w->self_attention_weight.qkv.bias) {  // allocated by attention_bias=true but Llama has no qkv bias
                GPUFree(w->self_attention_weight.qkv.bias);
                w->self_attention_weight.qkv.bias = nullptr;
            }
            llmie_api::upload(w->ffn_weight.gate_and_up.data, cast_vec<T>(h.gate_up));
            llmie_api::upload(w->ffn_weight.down.data, cast_vec<T>(h.down));
            w->self_attention_weight.qkv.is_transposed = true;
            w->self_attention_weight.output.is_transposed = true;
            w->ffn_weight.gate_and_up.is_transposed = true;
            w->ffn_weight.down.is_transposed = hf_layout_flags;  // all four true -> the fused engine is eligible
            ptrs.push_back(w.get());
            dev.push_back(std::move(w));
            orc_layer_weights ow{};
            ow.attn_norm = host[l].attn_norm.data();
            ow.qkv = host[l].qkv.data();
            ow.o = host[l].o.data();
            ow.o_bias = o_bias ? host[l].o_bias.data() : nullptr;
            ow.ffn_norm = host[l].ffn_norm.data();
            ow.gate_up = host[l].gate_up.data();
            ow.down = host[l].down.data();
            orc.push_back(ow);
        }
    }
};

// oracle composition of LlamaContextDecoder::forward (context_decoder.cpp:58-199, context_attention.cpp:143-312)
template <typename T>
static void oracle_context_decoder(const Model<T> &m, std::vector<float> &hidden /*[Tn,H] in/out*/, std::vector<float> &kc,
                                   std::vector<float> &vc, const std::vector<int> &lens, const std::vector<int> &hist,
                                   int max_seq, int rot_dim, float rot_base, float eps) {
    const int bs = static_cast<int>(lens.size());
    int Tn = 0, mq = 0, mk = 0;
    std::vector<int> ctx(bs);
    for (int b = 0; b < bs; ++b) {
        Tn += lens[b];
        mq = std::max(mq, lens[b]);
        ctx[b] = lens[b] + hist[b];
        mk = std::max(mk, ctx[b]);
    }
    const int nh = m.nh, kvh = m.kvh, hs = m.hs, H = m.H, QKV = m.QKV, I = m.I;
    std::vector<int> off(static_cast<size_t>(bs) * mq, 0), cum(bs + 1);
    orc_cal_padding_offset(off.data(), cum.data(), lens.data(), bs, mq);
    std::vector<float> mask(static_cast<size_t>(bs) * mq * mk);
    orc_build_causal_mask(mask.data(), lens.data(), ctx.data(), bs, mq, mk);
    std::vector<float> resid(hidden.size()), qkv(static_cast<size_t>(Tn) * QKV), attn(static_cast<size_t>(Tn) * H);
    std::vector<float> gu(static_cast<size_t>(Tn) * 2 * I), act(static_cast<size_t>(Tn) * I);
    for (int l = 0; l < m.L; ++l) {
        const orc_layer_weights &w = m.orc[l];
        orc_rmsnorm(hidden.data(), resid.data(), w.attn_norm, eps, Tn, H);
        orc_linear(hidden.data(), w.qkv, qkv.data(), Tn, H, QKV, 1);
        std::vector<float> q(static_cast<size_t>(bs) * nh * mq * hs, 0.f), k(static_cast<size_t>(bs) * kvh * mq * hs, 0.f), v(k.size(), 0.f);
        orc_qkv_bias_transpose_rope(q.data(), k.data(), v.data(), qkv.data(), nullptr, off.data(), hist.data(), bs, mq, Tn,
                                    nh, kvh, hs, rot_dim, rot_base);
        orc_concat_kv(k.data(), kc.data(), lens.data(), hist.data(), l, bs, kvh, mq, max_seq, hs);
        orc_concat_kv(v.data(), vc.data(), lens.data(), hist.data(), l, bs, kvh, mq, max_seq, hs);
        std::vector<float> kr(static_cast<size_t>(bs) * nh * mk * hs, 0.f), vr(kr.size(), 0.f);
        orc_repeat_kv(kc.data(), kr.data(), ctx.data(), l, bs, nh, kvh, mk, max_seq, hs);
        orc_repeat_kv(vc.data(), vr.data(), ctx.data(), l, bs, nh, kvh, mk, max_seq, hs);
        std::vector<float> qk(static_cast<size_t>(bs) * nh * mq * mk), pv(static_cast<size_t>(bs) * nh * mq * hs);
        orc_batched_gemm(q.data(), kr.data(), qk.data(), bs * nh, mq, mk, hs, 1);
        orc_scale_mask_softmax(qk.data(), mask.data(), qk.data(), 1.0f / std::sqrt(static_cast<float>(hs)), bs, nh, mq, mk);
        orc_batched_gemm(qk.data(), vr.data(), pv.data(), bs * nh, mq, hs, mk, 0);
        orc_transpose_remove_padding(pv.data(), attn.data(), off.data(), Tn, bs, mq, nh, hs);
        orc_linear(attn.data(), w.o, hidden.data(), Tn, H, H, 1);
        orc_fused_add_bias_residual_rmsnorm(resid.data(), hidden.data(), w.o_bias, w.ffn_norm, eps, Tn, H);
        orc_linear(hidden.data(), w.gate_up, gu.data(), Tn, H, 2 * I, 1);
        orc_silu_and_mul(gu.data(), act.data(), Tn, I);
        orc_linear(act.data(), w.down, hidden.data(), Tn, I, H, 1);
        orc_add_residual(resid.data(), hidden.data(), Tn, H);
    }
}


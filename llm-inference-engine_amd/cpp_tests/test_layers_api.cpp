// Layer-level tests through the C++ API mirror (LlamaSelfDecoder / LlamaContextDecoder / LlamaFFNLayer /
// LlamaModel with the reference's constructor arguments and TensorMap keys), checked against compositions of
// the oracle's kernels:   ./test_layers_api [1 = fp16]
#include "../src/layers/includes/context_decoder.h"
#include "../src/layers/includes/self_decoder.h"
#include "../src/utils/model_utils.h"
#include <cstdlib>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>

#include "test_layer_models.hpp"

template <typename T> static void run(bool fp16) {
    const DataType ty = getTensorType<T>(), ti = getTensorType<int>();
    const float rt = fp16 ? 2e-2f : 2e-4f, at = fp16 ? 2e-2f : 2e-4f;
    const int nh = fp16 ? 8 : 4, kvh = fp16 ? 4 : 4, hs = fp16 ? 128 : 32, I = fp16 ? 1376 : 344, L = 2;
    const int max_seq = 48, rot = hs;
    const float eps = 1e-5f, base = 10000.f;
    CublasWrapper gemm;
    CudaAllocator alloc;
    LlamaAttentionStaticParams sp{};
    sp.rotary_embedding_dim = rot;
    sp.rotary_embedding_base = base;
    sp.max_position_embeddings = 2048;
    sp.use_dynamic_ntk = false;
    hipStream_t stream = nullptr;

    for (int variant = 0; variant < 2; ++variant) {
        // variant 0: HF layout flags -> fused engine; variant 1: force the per-kernel loop (the reference's sequence)
        Model<T> m(nh, kvh, hs, I, L, /*o_bias=*/true, /*hf flags*/ true, 77);
        const int H = m.H, bs = 2, step = 9;
        std::mt19937_64 rng(5);
        std::vector<float> x = storage_round<T>(randn(rng, static_cast<size_t>(bs) * H, 1.f));
        std::vector<float> kc = storage_round<T>(randn(rng, static_cast<size_t>(L) * bs * kvh * max_seq * hs, 0.5f));
        std::vector<float> vc = storage_round<T>(randn(rng, kc.size(), 0.5f));
        DeviceArray<T> din(cast_vec<T>(x)), dout(x.size()), dk(cast_vec<T>(kc)), dv(cast_vec<T>(vc));
        DeviceArray<bool> dfin(bs);
        int h_step = step, h_layer = 0;
        TensorWrapper<T> tin(Device::GPU, ty, {bs, H}, din.d), tout(Device::GPU, ty, {bs, H}, dout.d);
        TensorWrapper<T> tk(Device::GPU, ty, {L, bs, kvh, max_seq, hs}, dk.d), tv(Device::GPU, ty, {L, bs, kvh, max_seq, hs}, dv.d);
        TensorWrapper<int> tstep(Device::CPU, ti, {1}, &h_step), tlayer(Device::CPU, ti, {1}, &h_layer);
        TensorWrapper<bool> tfin(Device::GPU, getTensorType<bool>(), {bs}, dfin.d);
        TensorMap inputs{{"decoder_input", &tin}, {"step", &tstep}, {"finished", &tfin}, {"layer_id", &tlayer}};
        TensorMap outputs{{"decoder_output", &tout}, {"all_k_cache", &tk}, {"all_v_cache", &tv}};
        LlamaAttentionDynamicParams dyn{};
        dyn.batch_size = bs;
        LlamaSelfDecoder<T> dec(nh, kvh, hs, I, L, sp, eps, stream, &gemm, &alloc);
        dec.use_fused_engine = (variant == 0);
        dec.forward(&inputs, &m.ptrs, &outputs, &dyn);
        orc_llama_cfg oc{nh, kvh, hs, I, L, 0, max_seq, rot, base, eps};
        std::vector<float> scratch(static_cast<size_t>(bs) * (2 * H + m.QKV + 3 * I));
        orc_self_decoder(&oc, m.orc.data(), x.data(), kc.data(), vc.data(), bs, step, scratch.data());
        check_close(variant == 0 ? "LlamaSelfDecoder (fused engine)" : "LlamaSelfDecoder (per-kernel loop)",
                    to_float(dout.download()), x, rt, at);
        check_close("LlamaSelfDecoder k cache", to_float(dk.download()), kc, fp16 ? rt : 1e-5f, fp16 ? at : 1e-5f);
    }

    {   // prefill: ragged batch with history, against the oracle composition; then prefill/decode consistency
        Model<T> m(nh, kvh, hs, I, L, /*o_bias=*/false, true, 99);
        const int H = m.H;
        const std::vector<int> lens{5, 3}, hist{0, 2};
        const int bs = 2, Tn = 8, mq = 5, mk = 5;
        std::mt19937_64 rng(6);
        std::vector<float> x = storage_round<T>(randn(rng, static_cast<size_t>(Tn) * H, 1.f));
        std::vector<float> kc = storage_round<T>(randn(rng, static_cast<size_t>(L) * bs * kvh * max_seq * hs, 0.5f));
        std::vector<float> vc = storage_round<T>(randn(rng, kc.size(), 0.5f));
        DeviceArray<T> din(cast_vec<T>(x)), dout(x.size()), dk(cast_vec<T>(kc)), dv(cast_vec<T>(vc));
        std::vector<int> ctx{5, 5};
        DeviceArray<int> dlen(lens), dhist(hist), dctx(ctx);
        int h_layer = 0;
        TensorWrapper<T> tin(Device::GPU, ty, {Tn, H}, din.d), tout(Device::GPU, ty, {Tn, H}, dout.d);
        TensorWrapper<T> tk(Device::GPU, ty, {L, bs, kvh, max_seq, hs}, dk.d), tv(Device::GPU, ty, {L, bs, kvh, max_seq, hs}, dv.d);
        TensorWrapper<int> tlen(Device::GPU, ti, {bs}, dlen.d), thist(Device::GPU, ti, {bs}, dhist.d), tctx(Device::GPU, ti, {bs}, dctx.d);
        TensorWrapper<int> tlayer(Device::CPU, ti, {1}, &h_layer);
        TensorMap inputs{{"decoder_input", &tin}, {"history_length", &thist}, {"input_length", &tlen},
                         {"context_length", &tctx}, {"layer_id", &tlayer}};
        TensorMap outputs{{"decoder_output", &tout}, {"all_k_cache", &tk}, {"all_v_cache", &tv}};
        LlamaAttentionDynamicParams dyn{};
        dyn.batch_size = bs;
        dyn.num_tokens = Tn;
        dyn.max_q_len = mq;
        dyn.max_k_len = mk;
        LlamaContextDecoder<T> cdec(nh, kvh, hs, I, L, &sp, eps, stream, &gemm, &alloc);
        cdec.forward(&inputs, &m.ptrs, &outputs, &dyn);
        {   // the reference's kernel sequence (padded q/k/v, batched GEMMs, softmax) gives the same hidden state
            DeviceArray<T> din2(cast_vec<T>(x)), dout2(x.size()), dk2(cast_vec<T>(kc)), dv2(cast_vec<T>(vc));
            TensorWrapper<T> tin2(Device::GPU, ty, {Tn, H}, din2.d), tout2(Device::GPU, ty, {Tn, H}, dout2.d);
            TensorWrapper<T> tk2(Device::GPU, ty, {L, bs, kvh, max_seq, hs}, dk2.d), tv2(Device::GPU, ty, {L, bs, kvh, max_seq, hs}, dv2.d);
            TensorMap inputs2{{"decoder_input", &tin2}, {"history_length", &thist}, {"input_length", &tlen},
                              {"context_length", &tctx}, {"layer_id", &tlayer}};
            TensorMap outputs2{{"decoder_output", &tout2}, {"all_k_cache", &tk2}, {"all_v_cache", &tv2}};
            LlamaContextDecoder<T> cdec2(nh, kvh, hs, I, L, &sp, eps, stream, &gemm, &alloc);
            cdec2.use_fused_engine = false;
            cdec2.forward(&inputs2, &m.ptrs, &outputs2, &dyn);
            check_close("LlamaContextDecoder engine path == per-kernel path", to_float(dout.download()),
                        to_float(dout2.download()), rt, at);
        }
        oracle_context_decoder(m, x, kc, vc, lens, hist, max_seq, rot, base, eps);
        check_close("LlamaContextDecoder", to_float(dout.download()), x, rt, at);
        // layer >= 1 K rows come out of a full fp16 layer: same tolerance as the hidden state
        check_close("LlamaContextDecoder k cache", to_float(dk.download()), kc, fp16 ? rt : 1e-5f, fp16 ? at : 1e-5f);
        // the decoder object is reusable (the reference deletes its sub-layers in freeBuf)
        cdec.freeBuf();
        din.upload(cast_vec<T>(storage_round<T>(x)));
        cdec.forward(&inputs, &m.ptrs, &outputs, &dyn);
        std::printf("LlamaContextDecoder reuse passed\n");
    }
    {   // prefill(n+1 tokens) last row == prefill(n tokens) then one decode step   (same kernels family, two paths)
        Model<T> m(nh, kvh, hs, I, L, false, true, 123);
        const int H = m.H, n = 6;
        std::mt19937_64 rng(7);
        std::vector<float> xs = storage_round<T>(randn(rng, static_cast<size_t>(n + 1) * H, 1.f));
        const size_t kvn = static_cast<size_t>(L) * kvh * max_seq * hs;
        auto prefill = [&](int tokens, DeviceArray<T> &dk, DeviceArray<T> &dv) {
            std::vector<float> in(xs.begin(), xs.begin() + static_cast<size_t>(tokens) * H);
            DeviceArray<T> din(cast_vec<T>(in)), dout(in.size());
            std::vector<int> lens{tokens}, hist{0};
            DeviceArray<int> dlen(lens), dhist(hist), dctx(lens);
            int h_layer = 0;
            TensorWrapper<T> tin(Device::GPU, ty, {tokens, H}, din.d), tout(Device::GPU, ty, {tokens, H}, dout.d);
            TensorWrapper<T> tk(Device::GPU, ty, {L, 1, kvh, max_seq, hs}, dk.d), tv(Device::GPU, ty, {L, 1, kvh, max_seq, hs}, dv.d);
            TensorWrapper<int> tlen(Device::GPU, ti, {1}, dlen.d), thist(Device::GPU, ti, {1}, dhist.d), tctx(Device::GPU, ti, {1}, dctx.d);
            TensorWrapper<int> tlayer(Device::CPU, ti, {1}, &h_layer);
            TensorMap inputs{{"decoder_input", &tin}, {"history_length", &thist}, {"input_length", &tlen},
                             {"context_length", &tctx}, {"layer_id", &tlayer}};
            TensorMap outputs{{"decoder_output", &tout}, {"all_k_cache", &tk}, {"all_v_cache", &tv}};
            LlamaAttentionDynamicParams dyn{};
            dyn.batch_size = 1;
            dyn.num_tokens = tokens;
            dyn.max_q_len = tokens;
            dyn.max_k_len = tokens;
            LlamaContextDecoder<T> cdec(nh, kvh, hs, I, L, &sp, eps, stream, &gemm, &alloc);
            cdec.forward(&inputs, &m.ptrs, &outputs, &dyn);
            std::vector<float> all = to_float(dout.download());
            return std::vector<float>(all.end() - H, all.end());
        };
        DeviceArray<T> k1(kvn), v1(kvn), k2(kvn), v2(kvn);
        CHECK(hipMemset(k1.d, 0, sizeof(T) * kvn)); CHECK(hipMemset(v1.d, 0, sizeof(T) * kvn));
        CHECK(hipMemset(k2.d, 0, sizeof(T) * kvn)); CHECK(hipMemset(v2.d, 0, sizeof(T) * kvn));
        const std::vector<float> full = prefill(n + 1, k1, v1);
        (void)prefill(n, k2, v2);
        std::vector<float> last(xs.end() - H, xs.end());
        DeviceArray<T> din(cast_vec<T>(last)), dout(H);
        DeviceArray<bool> dfin(1);
        int h_step = n + 1, h_layer = 0;
        TensorWrapper<T> tin(Device::GPU, ty, {1, H}, din.d), tout(Device::GPU, ty, {1, H}, dout.d);
        TensorWrapper<T> tk(Device::GPU, ty, {L, 1, kvh, max_seq, hs}, k2.d), tv(Device::GPU, ty, {L, 1, kvh, max_seq, hs}, v2.d);
        TensorWrapper<int> tstep(Device::CPU, ti, {1}, &h_step), tlayer(Device::CPU, ti, {1}, &h_layer);
        TensorWrapper<bool> tfin(Device::GPU, getTensorType<bool>(), {1}, dfin.d);
        TensorMap inputs{{"decoder_input", &tin}, {"step", &tstep}, {"finished", &tfin}, {"layer_id", &tlayer}};
        TensorMap outputs{{"decoder_output", &tout}, {"all_k_cache", &tk}, {"all_v_cache", &tv}};
        LlamaAttentionDynamicParams dyn{};
        dyn.batch_size = 1;
        LlamaSelfDecoder<T> dec(nh, kvh, hs, I, L, sp, eps, stream, &gemm, &alloc);
        dec.forward(&inputs, &m.ptrs, &outputs, &dyn);
        check_close("prefill(n+1) last row == prefill(n) + decode step", to_float(dout.download()), full, rt, at);
        check_close("KV caches agree after both paths", to_float(k2.download()), to_float(k1.download()), fp16 ? rt : 1e-5f, fp16 ? at : 1e-5f);
    }
    {   // checkpoint loader (layer_weights.cpp:49-81, llama_weights.cpp:48-74, weight_utils.cu:189-224): per-tensor fp32 .bin
        // files under the reference's names and HF shapes, converted to T on load
        const int lnh = 4, lkvh = 2, lhs = 32, lI = 96, lV = 50, lL = 2, lH = lnh * lhs, lQKV = (lnh + 2 * lkvh) * lhs;
        char tmpl[] = "/tmp/llmie_ckpt_XXXXXX";
        const char *dir = mkdtemp(tmpl);
        if (!dir) throw std::runtime_error("mkdtemp failed");
        const std::string root = std::string(dir) + "/";
        std::mt19937 wrng(77);
        std::vector<std::pair<std::string, std::vector<float>>> files;
        auto emit = [&](const std::string &name, size_t n) {
            std::vector<float> v(n);
            std::uniform_real_distribution<float> u(-1.f, 1.f);
            for (auto &x : v) x = u(wrng);
            FILE *f = std::fopen((root + name).c_str(), "wb");
            if (!f || std::fwrite(v.data(), sizeof(float), n, f) != n) throw std::runtime_error("cannot write " + name);
            std::fclose(f);
            files.emplace_back(name, v);
            return v;
        };
        const auto f_norm = emit("model.norm.weight.bin", lH);
        const auto f_head = emit("lm_head.weight.bin", static_cast<size_t>(lV) * lH);
        const auto f_emb = emit("model.embed_tokens.weight.bin", static_cast<size_t>(lV) * lH);
        std::vector<std::vector<float>> f_qkv, f_o, f_gu, f_down, f_in, f_post;
        for (int l = 0; l < lL; ++l) {
            const std::string pre = "model.layers." + std::to_string(l);
            f_in.push_back(emit(pre + ".input_layernorm.weight.bin", lH));
            f_post.push_back(emit(pre + ".post_attention_layernorm.weight.bin", lH));
            f_qkv.push_back(emit(pre + ".self_attn.qkv.weight.bin", static_cast<size_t>(lQKV) * lH));
            f_o.push_back(emit(pre + ".self_attn.o_proj.weight.bin", static_cast<size_t>(lH) * lH));
            f_gu.push_back(emit(pre + ".mlp.gate_up_proj.weight.bin", static_cast<size_t>(2 * lI) * lH));
            f_down.push_back(emit(pre + ".mlp.down_proj.weight.bin", static_cast<size_t>(lH) * lI));
        }
        LlamaWeight<T> w(lnh, lkvh, lhs, lI, lV, lL, /*attention_bias=*/false, getWeightType<T>());
        w.loadWeightsFromFile(root);
        auto fetch = [&](const T *dptr, size_t n) {
            std::vector<T> h(n);
            CHECK(hipMemcpy(h.data(), dptr, sizeof(T) * n, hipMemcpyDeviceToHost));
            return to_float(h);
        };
        auto same = [&](const char *what, const T *dptr, const std::vector<float> &file) {
            check_close(what, fetch(dptr, file.size()), storage_round<T>(file), 0.f, 0.f);  // exact: one conversion
        };
        same("loader: model.norm", w.out_rmsnorm_weight.gamma, f_norm);
        same("loader: lm_head", w.post_decoder_embedding_weight.data, f_head);
        same("loader: embed_tokens", w.pre_decoder_embedding_weight.data, f_emb);
        for (int l = 0; l < lL; ++l) {
            LlamaLayerWeight<T> *lw = w.llama_layer_weight[l].get();
            same("loader: input_layernorm", lw->attention_norm_weight.gamma, f_in[l]);
            same("loader: post_attention_layernorm", lw->ffn_norm_weight.gamma, f_post[l]);
            same("loader: qkv", lw->self_attention_weight.qkv.data, f_qkv[l]);
            same("loader: o_proj", lw->self_attention_weight.output.data, f_o[l]);
            same("loader: gate_up_proj", lw->ffn_weight.gate_and_up.data, f_gu[l]);
            same("loader: down_proj", lw->ffn_weight.down.data, f_down[l]);
            const bool flags = lw->self_attention_weight.qkv.is_transposed && lw->self_attention_weight.output.is_transposed &&
                               lw->ffn_weight.gate_and_up.is_transposed && lw->ffn_weight.down.is_transposed;
            if (!flags) { std::printf("FAIL loader: HF-layout flags\n"); ++g_failures; }
        }
        std::printf("checkpoint loader: %zu files round-tripped\n", files.size());
        for (const auto &f : files) std::remove((root + f.first).c_str());
        std::remove(dir);
    }
    {   // the user_entry.cpp flow: llm::createDummyLLMModel -> MakeInput -> Response(callback) -> MakeHistory
        llm::ModelConfig &c = llm::config();
        c.head_num = 4; c.kv_head_num = 4; c.head_size = 32; c.inter_size = 344; c.num_layers = 2;
        c.max_seq_len = 64; c.vocab_size = 30000; c.rotary_embedding_dim = 32;
        srand(42);
        std::unique_ptr<BaseModel> model(llm::createDummyLLMModel<T>("/nonexistent/tokenizer.bin"));
        int calls = 0, last_index = -2;
        bool ended = false;
        auto cb = [&](int index, const char *content) {
            (void)content;
            if (index == -1) ended = true; else { ++calls; last_index = index; }
        };
        const std::string reply = model->Response(model->MakeInput("", 0, "Hey, are you conscious? Can you talk to me?"), cb);
        const std::string hist = model->MakeHistory("", 0, "Hey", reply);
        const bool ok = ended && calls >= 1 && calls <= 20 && last_index == calls - 1 && !reply.empty() && hist.size() >= reply.size() && model->model_name == "llama";
        std::printf(ok ? "LlamaModel Response/MakeInput/MakeHistory passed (%d tokens)\n" : "FAIL LlamaModel chat flow (%d tokens)\n", calls);
        if (!ok) ++g_failures;
        const std::vector<int> first = static_cast<LlamaModel<T> *>(model.get())->last_token_ids;
        (void)model->response(model->makeInput("", 0, "again"), nullptr);
        const std::vector<int> second = static_cast<LlamaModel<T> *>(model.get())->last_token_ids;
        check_equal("LlamaModel deterministic tokens", second, first);
    }
    {   // the same flow with a vocabulary file in the reference's binary format (tokenizer.h:138-167): the prompt goes through
        // Encode (byte fallback + merges), every generated id through Decode
        char tmpl[] = "/tmp/llmie_vocab_XXXXXX";
        const int fd = mkstemp(tmpl);
        if (fd < 0) throw std::runtime_error("mkstemp failed");
        FILE *f = fdopen(fd, "wb");
        auto wi = [&](int v) { std::fwrite(&v, 4, 1, f); };
        auto wf = [&](float v) { std::fwrite(&v, 4, 1, f); };
        std::vector<std::pair<std::string, float>> toks;
        static const char *hexd = "0123456789ABCDEF";
        for (int c = 0; c < 256; ++c) toks.push_back({std::string("<0x") + hexd[c >> 4] + hexd[c & 15] + ">", 0.f});
        toks.push_back({"\xE2\x96\x81", -100.f});
        for (char c = 'a'; c <= 'z'; ++c) toks.push_back({std::string(1, c), -200.f});
        for (const char *w : {"he", "ll", "hell", "hello", "wo", "wor", "ld", "world", "\xE2\x96\x81hello", "\xE2\x96\x81world"})
            toks.push_back({w, -static_cast<float>(toks.size() % 17) - 1.f});
        wi(1); wi(0);  // version 1, empty key-value table
        wi(static_cast<int>(toks.size()));
        for (size_t i = 0; i < toks.size(); ++i) {
            wi(static_cast<int>(toks[i].first.size()));
            for (unsigned char c : toks[i].first) wi(c);
            wi(static_cast<int>(i) + 3);
            wf(toks[i].second);
        }
        std::fclose(f);
        srand(42);
        std::unique_ptr<BaseModel> model(llm::createDummyLLMModel<T>(tmpl));
        LlamaModel<T> *lm = static_cast<LlamaModel<T> *>(model.get());
        const std::vector<int> ids = lm->getTokenizer().Encode("hello world");
        const bool enc_ok = lm->getTokenizer().loaded && ids.size() == 2 && lm->getTokenizer().Decode(ids) == " hello world";
        std::string pieces;
        const std::string reply = model->Response(model->MakeInput("", 0, "hello world"), [&](int index, const char *c) {
            if (index >= 0) pieces += c;
        });
        std::string expect;
        for (int id : lm->last_token_ids) expect += lm->getTokenizer().Decode({id});
        const bool ok = enc_ok && reply == pieces && reply == expect && !lm->last_token_ids.empty();
        std::printf(ok ? "LlamaModel chat flow with a vocabulary file passed (%zu prompt ids)\n"
                       : "FAIL LlamaModel chat flow with a vocabulary file (%zu prompt ids)\n", ids.size());
        if (!ok) ++g_failures;
        std::remove(tmpl);
    }
}

int main(int argc, char **) {
    try {
        if (argc > 1) run<half>(true);
        else run<float>(false);
    } catch (const std::exception &e) {
        std::printf("FAIL: exception %s\n", e.what());
        return 2;
    }
    CHECK(hipDeviceSynchronize());
    std::printf(g_failures ? "%d FAILED\n" : "all passed (%d failures)\n", g_failures);
    return g_failures ? 1 : 0;
}

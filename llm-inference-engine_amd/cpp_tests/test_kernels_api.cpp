// Replays the reference's tests/unit_tests drivers through the C++ `launch*` templates
// (same shapes, same input recipes, same tolerances) and checks against the oracle.
//   ./test_kernels_api        fp32      ./test_kernels_api 1      fp16 (the reference's argv convention)
#include "../src/kernels/includes/rmsnorm.cuh"   // all launchers arrive through api/kernels.hpp
#include "test_common.hpp"

template <typename T> static void run(bool fp16) {
    const DataType ty = getTensorType<T>(), ti = getTensorType<int>();
    CublasWrapper gemm(nullptr, nullptr);
    fp16 ? gemm.setFP16GemmConfig() : gemm.setFP32GemmConfig();
    const float rt = fp16 ? 2e-3f : 1e-5f, at = fp16 ? 2e-3f : 1e-5f;

    {  // test_rmsnorm.cu:42-118: T=64, H=4096, eps 1e-6, x=(i*i%3)+1, gamma=(i%3)+1 (fp16: ones)
        const int Tn = 64, H = 4096;
        std::vector<float> x(static_cast<size_t>(Tn) * H), g(H);
        for (size_t i = 0; i < x.size(); ++i) x[i] = fp16 ? 1.f : static_cast<float>((i * i) % 3 + 1);
        for (int i = 0; i < H; ++i) g[i] = fp16 ? 1.f : static_cast<float>(i % 3 + 1);
        DeviceArray<T> dx(cast_vec<T>(x)), dr(x.size()), dg(cast_vec<T>(g));
        TensorWrapper<T> out(Device::GPU, ty, {Tn, H}, dx.d), res(Device::GPU, ty, {Tn, H}, dr.d);
        LayerNormWeight<T> w;
        w.gamma = dg.d;
        launchRMSNorm(&out, &res, &w, 1e-6f);
        std::vector<float> ex = x, er(x.size());
        orc_rmsnorm(ex.data(), er.data(), g.data(), 1e-6f, Tn, H);
        check_close("RMSNorm", to_float(dx.download()), ex, 1e-3f, 1e-3f);
        check_close("RMSNorm residual copy", to_float(dr.download()), er, 0, 0);
    }
    {  // test_add_residual_and_rmsnorm.cu: T=2048, H=128, eps 0.5 ; spec = fp32 kernel :43-121
        const int Tn = 2048, H = 128;
        std::mt19937_64 rng(1);
        std::vector<float> o = storage_round<T>(randn(rng, static_cast<size_t>(Tn) * H, 1.f)), r = storage_round<T>(randn(rng, o.size(), 1.f));
        std::vector<float> b = storage_round<T>(randn(rng, H, 0.1f)), g = storage_round<T>(randn(rng, H, 0.1f));
        for (auto &v : g) v += 1.f;
        g = storage_round<T>(g);
        DeviceArray<T> dout(cast_vec<T>(o)), dres(cast_vec<T>(r)), db(cast_vec<T>(b)), dg(cast_vec<T>(g));
        TensorWrapper<T> out(Device::GPU, ty, {Tn, H}, dout.d), res(Device::GPU, ty, {Tn, H}, dres.d);
        BaseWeight<T> norm;
        norm.bias = db.d;
        launchFusedAddBiasResidualAndRMSNorm(&res, &out, &norm, dg.d, 0.5f);
        orc_fused_add_bias_residual_rmsnorm(r.data(), o.data(), b.data(), g.data(), 0.5f, Tn, H);
        check_close("FusedAddBiasResidualAndRMSNorm", to_float(dout.download()), o, fp16 ? 4e-3f : 1e-5f, fp16 ? 4e-3f : 1e-5f);
        check_close("FusedAddBiasResidualAndRMSNorm residual", to_float(dres.download()), r, fp16 ? 1e-3f : 0, fp16 ? 1e-3f : 0);
    }
    {  // test_add_residual.cu: T=16, H=4096, both (i%2)+1
        const int Tn = 16, H = 4096;
        std::vector<float> a(static_cast<size_t>(Tn) * H);
        for (size_t i = 0; i < a.size(); ++i) a[i] = static_cast<float>(i % 2 + 1);
        DeviceArray<T> dres(cast_vec<T>(a)), dout(cast_vec<T>(a));
        TensorWrapper<T> res(Device::GPU, ty, {Tn, H}, dres.d), out(Device::GPU, ty, {Tn, H}, dout.d);
        launchAddResidual(&res, &out);
        std::vector<float> e = a;
        orc_add_residual(a.data(), e.data(), Tn, H);
        check_close("AddResidual", to_float(dout.download()), e, 0, 0);
    }
    {  // test_linear.cu: srand(233), M=64, K=N=4096, rand()%3, trans_b = true
        srand(233);
        const int M = 64, H = 4096;
        std::vector<float> w(static_cast<size_t>(H) * H), x(static_cast<size_t>(M) * H), e(static_cast<size_t>(M) * H);
        for (auto &v : w) v = static_cast<float>(rand() % 3);
        for (auto &v : x) v = static_cast<float>(rand() % 3);
        DeviceArray<T> dw(cast_vec<T>(w)), dx(cast_vec<T>(x)), dy(e.size());
        TensorWrapper<T> in(Device::GPU, ty, {M, H}, dx.d), out(Device::GPU, ty, {M, H}, dy.d);
        BaseWeight<T> weight;
        weight.shape = {H, H};
        weight.data = dw.d;
        launchLinearGemm(&in, &weight, &out, &gemm, false, true);
        orc_linear(x.data(), w.data(), e.data(), M, H, H, 1);
        check_close("Linear", to_float(dy.download()), storage_round<T>(e), fp16 ? 0 : 0, fp16 ? 0 : 1e-3f);
    }
    {  // test_silu_and_mul.cu: bs=128, I=11008, ones -> 0.7310586
        const int B = 128, I = 11008;
        std::vector<float> in(static_cast<size_t>(B) * 2 * I, 1.f), e(static_cast<size_t>(B) * I);
        DeviceArray<T> din(cast_vec<T>(in)), dout(e.size());
        TensorWrapper<T> tin(Device::GPU, ty, {B, 2, I}, din.d), tout(Device::GPU, ty, {B, I}, dout.d);
        launchSiluAndMul(&tin, &tout);
        orc_silu_and_mul(in.data(), e.data(), B, I);
        check_close("SiluAndMul", to_float(dout.download()), e, 0, fp16 ? 5e-4f : 1e-6f);
    }
    {  // test_build_causal_mask.cu: bs=64, q=128, k=512, rand() lens
        srand(1);
        const int B = 64, Q = 128, Kk = 512;
        std::vector<int> ql(B), kl(B);
        for (auto &v : ql) v = rand() % Q + 1;
        for (auto &v : kl) v = rand() % Kk + 1;
        DeviceArray<int> dq(ql), dk(kl);
        DeviceArray<T> dm(static_cast<size_t>(B) * Q * Kk);
        TensorWrapper<T> mask(Device::GPU, ty, {B, Q, Kk}, dm.d);
        TensorWrapper<int> tq(Device::GPU, ti, {B}, dq.d), tk(Device::GPU, ti, {B}, dk.d);
        launchBuildCausalMasks<T>(&mask, &tq, &tk);
        std::vector<float> e(dm.n);
        orc_build_causal_mask(e.data(), ql.data(), kl.data(), B, Q, Kk);
        check_close("BuildCausalMasks", to_float(dm.download()), e, 0, 0);
    }
    {  // test_transpose_and_remove_padding.cu: [2,2,4,2], in[i]=i, offs 0,0,2,2,2
        const int B = 2, NH = 2, S = 4, HS = 2, Tn = 5;
        std::vector<float> in(B * NH * S * HS);
        for (size_t i = 0; i < in.size(); ++i) in[i] = static_cast<float>(i);
        std::vector<int> off{0, 0, 2, 2, 2};
        DeviceArray<T> din(cast_vec<T>(in)), dout(Tn * NH * HS);
        DeviceArray<int> doff(off);
        TensorWrapper<T> tin(Device::GPU, ty, {B, NH, S, HS}, din.d), tout(Device::GPU, ty, {Tn, NH, HS}, dout.d);
        TensorWrapper<int> toff(Device::GPU, ti, {Tn}, doff.d);
        launchFusedTransposeAndRemovePadding(&tin, &toff, &tout);
        std::vector<float> e(dout.n);
        orc_transpose_remove_padding(in.data(), e.data(), off.data(), Tn, B, S, NH, HS);
        check_close("FusedTransposeAndRemovePadding", to_float(dout.download()), e, 0, 0);
    }
    {  // test_decoder_self_attention.cu: bs=1, nh=kvh=2, hs=4, max_seq=4, step=4 (oracle = kernel math)
        const int B = 1, NH = 2, HS = 4, MS = 4, step = 4;
        std::mt19937_64 rng(3);
        std::vector<float> qkv = storage_round<T>(randn(rng, B * 3 * NH * HS, 1.f));
        std::vector<float> kc = storage_round<T>(randn(rng, B * NH * MS * HS, 1.f)), vc = storage_round<T>(randn(rng, kc.size(), 1.f));
        DeviceArray<T> dqkv(cast_vec<T>(qkv)), dk(cast_vec<T>(kc)), dv(cast_vec<T>(vc)), dout(B * NH * HS);
        DeviceArray<bool> dfin(B);
        int h_step = step, h_layer = 0;
        TensorWrapper<T> tq(Device::GPU, ty, {B, 3 * NH, HS}, dqkv.d), tk(Device::GPU, ty, {1, B, NH, MS, HS}, dk.d);
        TensorWrapper<T> tv(Device::GPU, ty, {1, B, NH, MS, HS}, dv.d), tout(Device::GPU, ty, {B, NH * HS}, dout.d);
        TensorWrapper<int> tstep(Device::CPU, ti, {1}, &h_step), tlayer(Device::CPU, ti, {1}, &h_layer);
        TensorWrapper<bool> tfin(Device::GPU, getTensorType<bool>(), {B}, dfin.d);
        BaseWeight<T> bias;
        LlamaAttentionStaticParams sp{};
        sp.rotary_embedding_dim = 128; sp.rotary_embedding_base = 10000; sp.max_position_embeddings = 2048;
        launchDecoderMaskedMultiHeadAttention<T>(&tq, &bias, &tlayer, &tk, &tv, &tfin, &tstep, &tout, &sp);
        std::vector<float> e(dout.n);
        orc_decoder_mha(qkv.data(), nullptr, kc.data(), vc.data(), e.data(), 0, B, NH, NH, HS, MS, step);
        check_close("DecoderMaskedMultiHeadAttention", to_float(dout.download()), e, rt * 2, at * 2);
        check_close("DecoderMaskedMultiHeadAttention k append", to_float(dk.download()), kc, 0, 0);
    }
    {  // test_topk.cu: probs[i]=i (fp16: i%2048), [2,32000], K=5, 8 blocks per beam
        const int rows = 2, V = 32000, K = 5, bpb = 8;
        std::vector<float> p(static_cast<size_t>(rows) * V);
        for (size_t i = 0; i < p.size(); ++i) p[i] = fp16 ? static_cast<float>(i % 2048) : static_cast<float>(i);
        DeviceArray<T> dp(cast_vec<T>(p)), dtv(rows * bpb * K), dfv(rows * K);
        DeviceArray<int> dti(rows * bpb * K), dfi(rows * K);
        TensorWrapper<T> tp(Device::GPU, ty, {rows, V}, dp.d), ttv(Device::GPU, ty, {1, rows, bpb, K}, dtv.d), tfv(Device::GPU, ty, {rows, K}, dfv.d);
        TensorWrapper<int> tti(Device::GPU, ti, {1, rows, bpb, K}, dti.d), tfi(Device::GPU, ti, {rows, K}, dfi.d);
        launchTopKForBeamSearch(&tp, &tti, &ttv, &tfi, &tfv);
        std::vector<int> eid(rows * K);
        std::vector<float> ev(rows * K);
        orc_topk(p.data(), eid.data(), ev.data(), rows, V, K);
        check_equal("TopKForBeamSearch ids", dfi.download(), eid);
        check_close("TopKForBeamSearch vals", to_float(dfv.download()), ev, 0, 0);
    }
    {  // test_sampling.cu: bs=3, K=3, V=1000, step=6, end=10
        const int B = 3, K = 3;
        std::vector<int> id(B * K), seq(B, 4);
        std::vector<float> val(B * K);
        for (int i = 0; i < B * K; ++i) {
            id[i] = i;
            val[i] = static_cast<float>(K - 1 - (i % K));
        }
        DeviceArray<int> did(id), dseq(seq), dout(B);
        DeviceArray<T> dval(cast_vec<T>(val));
        DeviceArray<bool> dfin(B);
        CHECK(hipMemset(dfin.d, 0, B));
        TensorWrapper<int> tid(Device::GPU, ti, {B, K}, did.d), tseq(Device::GPU, ti, {B}, dseq.d), tout(Device::GPU, ti, {B}, dout.d);
        TensorWrapper<T> tval(Device::GPU, ty, {B, K}, dval.d);
        TensorWrapper<bool> tfin(Device::GPU, getTensorType<bool>(), {B}, dfin.d);
        MapStringToInt params{{"step", 6}, {"vocab_size", 1000}, {"end_id", 10}};
        launchSampling<T>(&tid, &tval, &tseq, &tfin, &tout, &params);
        std::vector<int> eo(B), es = seq;
        std::vector<uint8_t> ef(B, 0);
        orc_sampling(id.data(), val.data(), es.data(), ef.data(), eo.data(), B, K, 6, 10, 1000);
        check_equal("Sampling ids", dout.download(), eo);
        check_equal("Sampling seqlen", dseq.download(), es);
    }
}

int main(int argc, char **) {
    try {
        if (argc > 1) run<half>(true);
        else run<float>(false);
        {  // test_cal_padding_offset.cu (dtype independent): lens = 4-(i*i%3), bs=4... plus the header's doc example
            std::vector<int> lens{4, 3, 5};
            DeviceArray<int> dl(lens), doff(3 * 5), dcum(4);
            CHECK(hipMemset(doff.d, 0xff, sizeof(int) * 15));
            TensorWrapper<int> tl(Device::GPU, DataType::INT32, {3}, dl.d), toff(Device::GPU, DataType::INT32, {3, 5}, doff.d);
            TensorWrapper<int> tcum(Device::GPU, DataType::INT32, {4}, dcum.d);
            launchCalPaddingOffset(&toff, &tcum, &tl);
            check_equal("CalPaddingOffset cum_seqlens", dcum.download(), std::vector<int>{0, 4, 7, 12});
            std::vector<int> got = doff.download();
            got.resize(12);
            check_equal("CalPaddingOffset offsets", got, std::vector<int>{0, 0, 0, 0, 1, 1, 1, 3, 3, 3, 3, 3});
        }
        // error behaviour: a shape mismatch throws like LLM_CHECK (macro.h:74-94)
        bool threw = false;
        try {
            DeviceArray<float> a(4);
            TensorWrapper<float> in(Device::GPU, DataType::FP32, {1, 4}, a.d), out(Device::GPU, DataType::FP32, {1, 4}, a.d);
            BaseWeight<float> w;
            w.shape = {4, 5};
            w.data = a.d;
            CublasWrapper g;
            launchLinearGemm(&in, &w, &out, &g, false, true);
        } catch (const std::runtime_error &) {
            threw = true;
        }
        if (!threw) { std::printf("FAIL: shape mismatch did not throw\n"); ++g_failures; } else std::printf("LLM_CHECK throw passed\n");
    } catch (const std::exception &e) {
        std::printf("FAIL: exception %s\n", e.what());
        return 2;
    }
    CHECK(hipDeviceSynchronize());
    std::printf(g_failures ? "%d FAILED\n" : "all passed (%d failures)\n", g_failures);
    return g_failures ? 1 : 0;
}

// Diagnostic: s_memrealtime stamps of the 128-row split-K kernel (gemm_mid.cuh) at the shapes of a 128-token prefill / batch-128
// decode step.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMID_STAMPS -I llm-inference-engine_amd/csrc tools/micro/mid_probe.hip -o /tmp/mid_probe && /tmp/mid_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "gemm_mid.cuh"
using namespace llmie;

template <int WN, int NS, int WQ> static void run(const char *name, int M, int N, int K, int ks) {
    constexpr int wrow = WQ ? 64 : 128;
    const int lds = NS * (128 * 128 + (WQ ? ((4 * WN + 7) / 8) * 8 * 1024 : 64 * WN * wrow));
    hipFuncSetAttribute(reinterpret_cast<const void *>(mid_splitk_kernel<false, WN, NS, 128, WQ>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int KT = K / 64, per = (KT + ks - 1) / ks, mtiles = (N + 64 * WN - 1) / (64 * WN);
    void *x, *w, *junk;
    float *slab;
    hipMalloc(&x, (size_t)M * K * 2);
    hipMalloc(&w, (size_t)N * K * 2);
    hipMalloc(&slab, (size_t)ks * M * N * 4);
    hipMalloc(&junk, 1 << 29);
    hipMemset(x, 0x11, (size_t)M * K * 2);
    hipMemset(w, 0x22, (size_t)N * K * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f, cold = 0.f;
    for (int rep = 0; rep < 6; ++rep) {
        if (rep < 3) hipMemset(junk, rep, 1 << 29);   // cold caches for the first three
        hipEventRecord(e0);
        mid_splitk_kernel<false, WN, NS, 128, WQ><<<mtiles * ks, 512, lds>>>(x, w, slab, M, N, K, ks, per);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) cold = ms * 1000.f;
        best = std::min(best, ms * 1000.f);
    }
    std::vector<unsigned long long> st(1024 * 8);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(llmie::mid_stamp_buf), st.size() * 8);
    const int nwg = std::min(1024, mtiles * ks);
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < nwg; ++b) t0 = std::min(t0, st[b * 8]);
    const char *names[6] = {"start", "prologue issued", "tile 0 landed", "half way", "loop done", "stores issued"};
    printf("== %s  M=%d N=%d K=%d  WN=%d NS=%d ks=%d  grid %d  event time: cold %.1f us, warm %.1f us  (weights %.1f MB -> %.2f TB/s warm)\n", name, M, N, K, WN, NS,
           ks, mtiles * ks, cold, best, (double)N * K * (WQ ? 1 : 2) / 1e6, (double)N * K * (WQ ? 1 : 2) / best / 1e6);
    for (int i = 0; i < 6; ++i) {
        std::vector<double> v;
        for (int b = 0; b < nwg; ++b) v.push_back((double)(st[b * 8 + i] - t0) * 0.01);
        std::sort(v.begin(), v.end());
        printf("   %-16s median %6.2f  [%6.2f .. %6.2f] us\n", names[i], v[v.size() / 2], v.front(), v.back());
    }
    hipFree(x); hipFree(w); hipFree(slab); hipFree(junk);
}

int main() {
    run<2, 4, 0>("O / fp16", 128, 4096, 4096, 8);
    run<2, 4, 0>("O / fp16 ks=4", 128, 4096, 4096, 4);
    run<2, 4, 0>("down / fp16", 128, 4096, 11008, 8);
    run<3, 3, 0>("QKV / fp16", 128, 12288, 4096, 4);
    run<3, 3, 0>("gate_up / fp16", 128, 22016, 4096, 2);
    run<2, 6, 8>("O / int8", 128, 4096, 4096, 8);
    run<3, 5, 8>("gate_up / int8", 128, 22016, 4096, 2);
    return 0;
}

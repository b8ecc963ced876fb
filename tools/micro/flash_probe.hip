// Diagnostic: the flash prefill kernel alone at the bench's shape, (a) timed, optionally with parts of its work compiled out
// (-DFLASH_SKIP_STAGE / _SOFTMAX / _K / _V: WRONG results, timing only -- what each part costs on the critical path), and
// (b) with shader-clock stamps inside prefill_flash_kernel (prefill.hip, -DFLASH_STAMPS): cycles a wave spends per key tile in
// (0) the barrier, (1) staging (publish tile t + 1, request tile t + 2), (2) K fragment reads + S MFMAs, (3) softmax, (4) V^T reads +
// PV MFMAs, (5) the whole iteration -- summed over the tiles of the first 64 workgroups (the longest: last query tiles), at the
// bench's prefill shape (32 heads, 2048 tokens).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFLASH_STAMPS -I llm-inference-engine_amd/csrc -I include tools/micro/flash_probe.hip llm-inference-engine_amd/csrc/runtime.cpp -o /tmp/flash_probe && /tmp/flash_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "../../llm-inference-engine_amd/csrc/prefill.hip"
using namespace llmie;

int main() {
    const int nh = 32, T = 2048, hs = 128, max_seq = 2048;
    half_t *qkv, *kc, *vc, *out;
    int32_t *cum, *hist;
    float2 *rope;
    hipMalloc(&qkv, (size_t)T * 3 * nh * hs * 2);
    hipMalloc(&kc, (size_t)nh * max_seq * hs * 2);
    hipMalloc(&vc, (size_t)nh * max_seq * hs * 2);
    hipMalloc(&out, (size_t)T * nh * hs * 2);
    hipMalloc(&cum, 8);
    hipMalloc(&hist, 4);
    hipMalloc(&rope, (size_t)max_seq * 64 * 8);
    std::vector<unsigned short> h((size_t)T * 3 * nh * hs);
    unsigned s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = 0x3000 | ((s >> 9) & 0x83ff); }   // small random fp16 values of both signs
    hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(kc, h.data(), (size_t)nh * max_seq * hs * 2, hipMemcpyHostToDevice);
    hipMemcpy(vc, h.data() + 777, (size_t)nh * max_seq * hs * 2, hipMemcpyHostToDevice);
    const int32_t cumh[2] = {0, T}, histh[1] = {0};
    hipMemcpy(cum, cumh, 8, hipMemcpyHostToDevice);
    hipMemcpy(hist, histh, 4, hipMemcpyHostToDevice);
    hipMemset(rope, 0, (size_t)max_seq * 64 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        // rope_done = 1: the flash kernel alone (the caches are filled above)
        prefill_attention_f16(qkv, nullptr, kc, vc, out, cum, hist, rope, 0, 1, T, T, nh, nh, hs, max_seq, hs, nullptr, 0, 1.f, 1.f, nullptr, 0, 0, 1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms * 1000.f);
    }
    printf("flash kernel, 32 heads x 2048 tokens: %.1f us (best of 5)\n", best);
#ifndef FLASH_STAMPS
    return 0;
#else
    std::vector<unsigned long long> st(64 * 8 * 8);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(llmie::flash_stamp_buf), st.size() * 8);
    const char *names[6] = {"barrier", "staging", "K reads + S MFMAs", "softmax", "V reads + PV MFMAs", "iteration"};
    // the first 64 workgroups in launch order = the LAST query tile of every head (32 key tiles); report per-tile cycles
    for (int w = 0; w < 8; ++w) {
        printf("wave %d:", w);
        for (int i = 0; i < 6; ++i) {
            double sum = 0;
            for (int b = 0; b < 32; ++b) sum += (double)st[(b * 8 + w) * 8 + i];
            printf("  %s %.0f", names[i], sum / 32 / 32);
        }
        printf("   (cycles per key tile, mean over 32 workgroups x 32 tiles)\n");
    }
    return 0;
#endif
}

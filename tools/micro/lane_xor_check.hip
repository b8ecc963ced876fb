// Checks device_utils.cuh's LDS-free xor-lane exchanges (DPP / gfx950 row swaps) against __shfl_xor on one wave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I llm-inference-engine_amd/csrc -I include tools/micro/lane_xor_check.hip -o /tmp/lane_xor_check && /tmp/lane_xor_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include "device_utils.cuh"
using namespace llmie;

__global__ void k(const float *in, float *out) {
    const float v = in[threadIdx.x];
    float *o = out + threadIdx.x;
    o[0 * 64] = lane_xor_lt16<1>(v);  o[1 * 64] = __shfl_xor(v, 1, 64);
    o[2 * 64] = lane_xor_lt16<2>(v);  o[3 * 64] = __shfl_xor(v, 2, 64);
    o[4 * 64] = lane_xor_lt16<4>(v);  o[5 * 64] = __shfl_xor(v, 4, 64);
    o[6 * 64] = lane_xor_lt16<8>(v);  o[7 * 64] = __shfl_xor(v, 8, 64);
    o[8 * 64] = lane_xor_sum<16>(v);  o[9 * 64] = v + __shfl_xor(v, 16, 64);
    o[10 * 64] = lane_xor_sum<32>(v); o[11 * 64] = v + __shfl_xor(v, 32, 64);
    o[12 * 64] = lane_xor_max<16>(v); o[13 * 64] = fmaxf(v, __shfl_xor(v, 16, 64));
    o[14 * 64] = lane_xor_max<32>(v); o[15 * 64] = fmaxf(v, __shfl_xor(v, 32, 64));
    float g = v;
    for (int w = 32; w > 0; w >>= 1) g += __shfl_xor(g, w, 64);
    o[16 * 64] = group_sum<64>(v);    o[17 * 64] = g;
    float g8 = v;
    for (int w = 4; w > 0; w >>= 1) g8 += __shfl_xor(g8, w, 64);
    o[18 * 64] = group_sum<8>(v);     o[19 * 64] = g8;
    o[20 * 64] = lane_xor<16>(v);     o[21 * 64] = __shfl_xor(v, 16, 64);
    o[22 * 64] = lane_xor<32>(v);     o[23 * 64] = __shfl_xor(v, 32, 64);
    const int iv = __builtin_bit_cast(int, v) ^ 0x5a5a;
    o[24 * 64] = __builtin_bit_cast(float, lane_xor_o(iv, 16) + lane_xor_o(iv, 4));
    o[25 * 64] = __builtin_bit_cast(float, __shfl_xor(iv, 16, 64) + __shfl_xor(iv, 4, 64));
}

int main() {
    float h[64], *din, *dout, r[26 * 64];
    unsigned s = 7;
    for (int i = 0; i < 64; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)((int)(s >> 8) % 2001 - 1000) / 37.0f; }
    hipMalloc(&din, sizeof(h));
    hipMalloc(&dout, sizeof(r));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(din, dout);
    hipMemcpy(r, dout, sizeof(r), hipMemcpyDeviceToHost);
    const char *names[13] = {"xor 1", "xor 2", "xor 4", "xor 8", "sum 16", "sum 32", "max 16", "max 32", "group_sum<64>", "group_sum<8>", "xor 16", "xor 32", "int xor 16 + 4"};
    int bad = 0;
    for (int t = 0; t < 13; ++t) {
        int diff = 0;
        for (int i = 0; i < 64; ++i) diff += __builtin_memcmp(&r[(2 * t) * 64 + i], &r[(2 * t + 1) * 64 + i], 4) != 0;
        printf("%-14s %s (%d lanes differ)\n", names[t], diff ? "MISMATCH" : "ok", diff);
        if (diff) {
            for (int i = 0; i < 16; ++i) printf("   lane %2d: got %g want %g\n", i, r[(2 * t) * 64 + i], r[(2 * t + 1) * 64 + i]);
        }
        bad += diff != 0;
    }
    return bad ? 1 : 0;
}

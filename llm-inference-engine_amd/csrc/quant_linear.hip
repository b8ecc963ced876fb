// Weight-only quantised linears (the reference only plans them: README.md:36-39, linear.cuh:12 TODO):
//   llmie_linear_w8a16  int8 weights [N,K] + fp16 per-row scale
//   llmie_linear_w4a16  packed int4 [N,K/2] (low nibble = even k, value = nibble-8) + fp16 scale per (row, group)
//   llmie_quantize_w8 / _w4  symmetric round-to-nearest quantisers (device side)
// M <= 8: the K-split streaming GEMV of gemm_kernels.cuh with in-register de-quantisation (bytes per weight
// halve / quarter, so the HBM-bound decode step speeds up accordingly).  8 < M <= 64: skinny MFMA kernel below:
// int8 fragments are expanded to fp16 in registers and fed to v_mfma_f32_16x16x32_f16, scale in the epilogue.
#include "gemm_kernels.cuh"
#include "llmie_internal.h"

#include <cstdlib>

namespace llmie {

// 8 int8 (two words) -> half8 (exact)
__device__ __forceinline__ half8_t dequant8(unsigned int w0, unsigned int w1) {
    const half2_t off = {static_cast<half_t>(1152.f), static_cast<half_t>(1152.f)};
    const unsigned int v0 = w0 ^ 0x80808080u, v1 = w1 ^ 0x80808080u;
    const half2_t a = as_half2(__builtin_amdgcn_perm(0x64646464u, v0, 0x04010400u)) - off;
    const half2_t b = as_half2(__builtin_amdgcn_perm(0x64646464u, v0, 0x04030402u)) - off;
    const half2_t c = as_half2(__builtin_amdgcn_perm(0x64646464u, v1, 0x04010400u)) - off;
    const half2_t d = as_half2(__builtin_amdgcn_perm(0x64646464u, v1, 0x04030402u)) - off;
    return half8_t{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}

// D[n, m] = scale[n] * sum_k Wq[n,k] x[m,k]; one workgroup = one 16-row weight tile, NW waves split K in 64-wide
// steps (one 16-byte load per lane = 16 consecutive k of one row -> two MFMA A fragments).
template <int MT, int NW>
__global__ __launch_bounds__(NW * 64) void skinny_mfma_w8_kernel(const half_t *__restrict__ x,
                                                                 const int8_t *__restrict__ Wq,
                                                                 const half_t *__restrict__ scale, half_t *y, int M, int K,
                                                                 int N, const half_t *__restrict__ bias,
                                                                 const half_t *residual) {
    __shared__ floatx4 red[NW][MT][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int nrow = min(blockIdx.x * 16 + r, N - 1);
    const int8_t *wp = Wq + static_cast<size_t>(nrow) * K + 16 * q;
    const half_t *xp[MT];
    bool xok[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        xok[j] = (16 * j + r) < M;
        xp[j] = x + static_cast<size_t>(xok[j] ? 16 * j + r : 0) * K + 16 * q;
    }
    floatx4 acc[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    constexpr int U = 4;
    const int ksteps = K >> 6;
    for (int s0 = wave * U; s0 < ksteps; s0 += NW * U) {
        uint4_t a[U];
        half8_t b0[U][MT], b1[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {  // unconditional loads (clamped step / row 0 for rows past M), zeroed by select below
            const int s = min(s0 + u, ksteps - 1);
            a[u] = load_nt(reinterpret_cast<const uint4_t *>(wp + 64 * s));
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                b0[u][j] = *reinterpret_cast<const half8_t *>(xp[j] + 64 * s);
                b1[u][j] = *reinterpret_cast<const half8_t *>(xp[j] + 64 * s + 8);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = s0 + u < ksteps;
            const half8_t a0 = dequant8(a[u][0], a[u][1]), a1 = dequant8(a[u][2], a[u][3]);
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, (live && xok[j]) ? b0[u][j] : z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, (live && xok[j]) ? b1[u][j] : z, acc[j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) red[wave][j][lane] = acc[j];
    __syncthreads();
    for (int j = wave; j < MT; j += NW) {
        floatx4 s = red[0][j][lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) s += red[w][j][lane];
        const int m = 16 * j + r;
        const int n0 = blockIdx.x * 16 + 4 * q;
        if (m < M) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n0 + e < N) {
                    float v = s[e] * to_f32(scale[n0 + e]);
                    if (bias) v += to_f32(bias[n0 + e]);
                    if (residual) v += to_f32(residual[static_cast<size_t>(m) * N + n0 + e]);
                    y[static_cast<size_t>(m) * N + n0 + e] = from_f32<half_t>(v);
                }
        }
    }
}

// ---- quantisers: one workgroup per row (int8) / per (row, group) (int4) ----
__global__ __launch_bounds__(256) void quantize_w8_kernel(const half_t *__restrict__ w, int8_t *__restrict__ q,
                                                          half_t *__restrict__ scale, int K) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const half_t *src = w + row * K;
    float amax = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) amax = fmaxf(amax, fabsf(to_f32(src[k])));
    amax = block_max<4>(amax, red);
    half_t sh = from_f32<half_t>(amax / 127.0f);
    if (to_f32(sh) == 0.f) sh = from_f32<half_t>(1.0f);
    if (threadIdx.x == 0) scale[row] = sh;
    const float s = to_f32(sh);
    for (int k = threadIdx.x; k < K; k += 256) {
        float v = rintf(to_f32(src[k]) / s);
        v = fminf(fmaxf(v, -127.f), 127.f);
        q[row * K + k] = static_cast<int8_t>(v);
    }
}

__global__ __launch_bounds__(64) void quantize_w4_kernel(const half_t *__restrict__ w, uint8_t *__restrict__ q,
                                                         half_t *__restrict__ scale, int K, int group) {
    const int groups = K / group;
    const size_t row = blockIdx.x / groups;
    const int g = blockIdx.x % groups;
    const half_t *src = w + row * K + static_cast<size_t>(g) * group;
    float amax = 0.f;
    for (int k = threadIdx.x; k < group; k += 64) amax = fmaxf(amax, fabsf(to_f32(src[k])));
    amax = wave_max(amax);
    half_t sh = from_f32<half_t>(amax / 7.0f);
    if (to_f32(sh) == 0.f) sh = from_f32<half_t>(1.0f);
    if (threadIdx.x == 0) scale[row * groups + g] = sh;
    const float s = to_f32(sh);
    uint8_t *dst = q + (row * K + static_cast<size_t>(g) * group) / 2;
    for (int b = threadIdx.x; b < group / 2; b += 64) {
        float lo = fminf(fmaxf(rintf(to_f32(src[2 * b]) / s), -8.f), 7.f);
        float hi = fminf(fmaxf(rintf(to_f32(src[2 * b + 1]) / s), -8.f), 7.f);
        dst[b] = static_cast<uint8_t>((static_cast<int>(lo) + 8) | ((static_cast<int>(hi) + 8) << 4));
    }
}

// fp16 image of quantised weights: w16[n, k] = fp16(code * scale) -- int8: code = the byte, scale[n]; int4: code = nibble - 8,
// scale[n, k / group].  One thread per 8 weights (16 bytes out); the product of an integer below 2^7 and an fp16 scale is exact
// in fp32 and rounded to fp16 once.
template <int WBITS>
__global__ __launch_bounds__(256) void dequant_f16_kernel(const unsigned char *__restrict__ wq, const half_t *__restrict__ scale,
                                                          half_t *__restrict__ w16, size_t n8, int K, int group) {
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n8) return;
    const size_t e0 = i * 8, row = e0 / K;
    const int k = static_cast<int>(e0 - row * K);
    half8_t o;
    if constexpr (WBITS == 8) {
        const float s = to_f32(scale[row]);
        const uint2 v = *reinterpret_cast<const uint2 *>(wq + e0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int b = static_cast<int>(((j < 4 ? v.x : v.y) >> (8 * (j & 3))) & 0xffu);
            o[j] = from_f32<half_t>(static_cast<float>(static_cast<int8_t>(b)) * s);
        }
    } else {
        const float s = to_f32(scale[row * (K / group) + k / group]);
        const unsigned v = *reinterpret_cast<const unsigned *>(wq + e0 / 2);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = from_f32<half_t>(static_cast<float>(static_cast<int>((v >> (4 * j)) & 0xfu) - 8) * s);
    }
    *reinterpret_cast<half8_t *>(w16 + e0) = o;
}

int dequantize_weights_f16(int wbits, const void *wq, const half_t *scale, half_t *w16, int N, int K, int group, hipStream_t st) {
    if ((wbits != 8 && wbits != 4) || K % 8 != 0 || (wbits == 4 && (group <= 0 || group % 8 != 0 || K % group != 0)) ||
        (reinterpret_cast<uintptr_t>(wq) % 8) != 0 || (reinterpret_cast<uintptr_t>(w16) % 16) != 0) {
        set_error("dequantize_weights: bits=%d K=%d group=%d not supported (K %% 8, group %% 8, aligned pointers)", wbits, K, group);
        return LLMIE_ERR_UNSUPPORTED;
    }
    const size_t n8 = static_cast<size_t>(N) * K / 8;
    const unsigned blocks = static_cast<unsigned>((n8 + 255) / 256);
    if (wbits == 8) dequant_f16_kernel<8><<<blocks, 256, 0, st>>>(static_cast<const unsigned char *>(wq), scale, w16, n8, K, group);
    else dequant_f16_kernel<4><<<blocks, 256, 0, st>>>(static_cast<const unsigned char *>(wq), scale, w16, n8, K, group);
    return launch_status("dequantize_weights");
}

size_t linear_wq_dequant_bytes(int wbits, int M, int K, int N) {
    if (M < kWqPrefillRows || (wbits != 8 && wbits != 4) || K % 8 != 0) return 0;
    return static_cast<size_t>(N) * K * sizeof(half_t);
}

int linear_wq(int wbits, const half_t *x, const void *wq, const half_t *scale, half_t *y, int M, int K, int N, int group,
              int epi, const half_t *bias, const half_t *residual, const half_t *gamma, const half_t *pre_bias, float eps,
              SlabWs ws, hipStream_t st, void *deq, size_t deq_bytes) {
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wq) |
                           reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(pre_bias)) % 16 == 0) &&
                         (static_cast<size_t>(K) * wbits / 8) % 16 == 0;
    // ---- prefill-sized row counts: MFMA-bound, the weights are read M / 256 times from L2 instead of streamed once ----
    // (round 3) int8, M from 192, shapes whose 256-row grid does not fill the chip (N = 4096 of a 7B layer at 193 .. 1023 tokens):
    // split-K passes of 128 rows on the int8 rows against the fp16 image + a partly filled tile grid, by a time model fitted to the
    // sweep (us; O / down of a 7B layer: 22 / 33 per pass against 56 / 123 for the image route whatever the row count):
    //   passes x (N K bytes / 2.8 TB/s + 16)   <   N K x 3 bytes / 4 TB/s  +  K / 64 x 0.55
    bool int8_mid_passes = false;
    if (wbits == 8 && M >= kWqPrefillRows && !gamma && epi == EPI_NONE && ws.p && aligned && K % 256 == 0 && K >= 512 &&
        !g8p_w8_eligible(M, K, N, x, wq, scale, y) && ws.floats >= linear_splitk_ws_floats(8, 128, K, N)) {
        const float nk = static_cast<float>(N) * K;
        const float t_passes = ((M + 127) / 128) * (nk / 2.8e6f + 16.f), t_image = nk * 3.f / 4.0e6f + (K / 64) * 0.55f;
        int8_mid_passes = t_passes * 1.05f < t_image;   // (ties go to the image route)
    }
    if (M >= kWqPrefillRows && !gamma && !int8_mid_passes) {
        // int8: the eight-phase GEMM takes the int8 rows as they are (raw bytes HBM -> LDS by DMA, de-quantised at fragment read,
        // scale in the epilogue)
        if (wbits == 8 && epi == EPI_SWIGLU && !bias && !residual && g8p_w8_swiglu_eligible(M, K, N, x, wq, scale, y)) {
            gemm256_swiglu_launch(false, x, wq, y, M, N, K, nullptr, reinterpret_cast<const float *>(scale), st, 8);
            return launch_status("linear_w8a16(gemm8p SwiGLU)");
        }
        if (wbits == 8 && epi == EPI_NONE && g8p_w8_eligible(M, K, N, x, wq, scale, y) &&
            (reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual)) % 8 == 0) {
            gemm256_launch(false, x, wq, y, M, N, K, bias, residual, nullptr, reinterpret_cast<const float *>(scale), st, 8);
            return launch_status("linear_w8a16(gemm8p)");
        }
        // other shapes, and int4 (group scales along K): one pass writes the fp16 image of the matrix into the caller's scratch,
        // the fp16 GEMM reads it back (mostly from the 256 MiB Infinity Cache): + (wbits / 8 + 2) bytes of traffic per weight
        const size_t need = linear_wq_dequant_bytes(wbits, M, K, N);
        if (deq && need && deq_bytes >= need && reinterpret_cast<uintptr_t>(deq) % 16 == 0 && reinterpret_cast<uintptr_t>(wq) % 8 == 0 &&
            (wbits == 8 || (group % 8 == 0 && K % group == 0))) {
            if (epi == EPI_SWIGLU && !gemm256_swiglu_fills(M, N)) goto no_prefill_form;   // (the fp16 GEMM has no fused SwiGLU there)
            int rc = dequantize_weights_f16(wbits, wq, scale, static_cast<half_t *>(deq), N, K, group, st);
            if (rc) return rc;
            return linear_f16_nk(x, static_cast<const half_t *>(deq), y, M, K, N, epi, bias, residual, SlabWs{nullptr, 0}, st);
        }
    }
no_prefill_form:
    if (aligned && ksplit_eligible(M, K, wbits)) {
        const GemvArgs a{x, wq, y, K, N, bias, residual, gamma, pre_bias, eps, epi, gamma ? 1 : 0, scale, group};
        if (gemv_q_launch(wbits, M, a, st)) return launch_status("linear_wq");
    }
    if (aligned && wbits == 4 && group == 128 && K % 256 == 0 && K >= 512 && M > 8 && !gamma && ws.p)
        return linear_splitk(4, x, wq, scale, y, M, K, N, epi, bias, residual, ws, st);  // MFMA path, group scales in the kernel
    if (aligned && wbits == 4) {
        // other int4 shapes: batches beyond the GEMV's register budget run as row chunks of the largest eligible
        // size (the weights are streamed once per chunk -- correct for any batch, bandwidth-efficient only for small ones)
        int mc = 8;
        while (mc > 0 && !ksplit_eligible(mc, K, 4)) --mc;
        if (mc > 0) {
            const int out_n = epi == EPI_SWIGLU ? N / 2 : N;
            for (int m0 = 0; m0 < M; m0 += mc) {
                const int m = M - m0 < mc ? M - m0 : mc;
                const GemvArgs a{x + static_cast<size_t>(m0) * K, wq, y + static_cast<size_t>(m0) * out_n, K, N, bias,
                                 residual ? residual + static_cast<size_t>(m0) * N : nullptr, gamma, pre_bias, eps, epi, gamma ? 1 : 0,
                                 scale, group};
                if (!gemv_q_launch(4, m, a, st)) {
                    set_error("linear_wq: no int4 GEMV instantiation for M=%d K=%d", m, K);
                    return LLMIE_ERR_UNSUPPORTED;
                }
            }
            return launch_status("linear_wq(int4, row chunks)");
        }
    }
    if (gamma) {
        set_error("linear_wq: fused norm only on the GEMV path (M=%d K=%d bits=%d)", M, K, wbits);
        return LLMIE_ERR_UNSUPPORTED;
    }
    if (wbits == 8 && aligned && K % 256 == 0 && K >= 512 && ws.p)
        return linear_splitk(8, x, wq, scale, y, M, K, N, epi, bias, residual, ws, st);
    if (epi != EPI_NONE) {
        set_error("linear_wq: fused SwiGLU needs the GEMV or split-K path (M=%d K=%d bits=%d)", M, K, wbits);
        return LLMIE_ERR_UNSUPPORTED;
    }
    if (wbits == 8 && aligned && K % 64 == 0 && M <= 64) {
        const int tiles = (N + 15) / 16;
        const int mt = (M + 15) / 16;
        const int8_t *w8 = static_cast<const int8_t *>(wq);
        switch (mt) {
            case 1: skinny_mfma_w8_kernel<1, 8><<<tiles, 512, 0, st>>>(x, w8, scale, y, M, K, N, bias, residual); break;
            case 2: skinny_mfma_w8_kernel<2, 8><<<tiles, 512, 0, st>>>(x, w8, scale, y, M, K, N, bias, residual); break;
            case 3: skinny_mfma_w8_kernel<3, 8><<<tiles, 512, 0, st>>>(x, w8, scale, y, M, K, N, bias, residual); break;
            default: skinny_mfma_w8_kernel<4, 8><<<tiles, 512, 0, st>>>(x, w8, scale, y, M, K, N, bias, residual); break;
        }
        return launch_status("linear_w8a16");
    }
    // shapes none of the quantised kernels take (K not a multiple of their sub-blocks): the fp16 image, where the caller gave room
    if (deq && K % 8 == 0 && deq_bytes >= static_cast<size_t>(N) * K * sizeof(half_t) && reinterpret_cast<uintptr_t>(deq) % 16 == 0 &&
        reinterpret_cast<uintptr_t>(wq) % 8 == 0 && (wbits == 8 || (group % 8 == 0 && K % group == 0)) &&
        (epi != EPI_SWIGLU || M <= 64 || gemm256_swiglu_fills(M, N))) {
        int rc = dequantize_weights_f16(wbits, wq, scale, static_cast<half_t *>(deq), N, K, group, st);
        if (rc) return rc;
        return linear_f16_nk(x, static_cast<const half_t *>(deq), y, M, K, N, epi, bias, residual, SlabWs{nullptr, 0}, st);
    }
    set_error("linear_wq: unsupported shape M=%d K=%d N=%d bits=%d without a split-K workspace (int8: M<=64, K%%64==0; int4: "
              "M<=8 on the GEMV path); size one with llmie_linear_workspace_bytes()", M, K, N, wbits);
    return LLMIE_ERR_UNSUPPORTED;
}

}  // namespace llmie

using namespace llmie;

// [fp16 image | slabs] split of a caller workspace (a workspace too small for the image keeps the round-2 meaning: all slabs)
struct WqWorkspace {
    void *deq;
    size_t deq_bytes;
    SlabWs slabs;
};
static WqWorkspace wq_workspace(int wbits, int M, int K, int N, void *workspace, size_t workspace_bytes) {
    const size_t dq = (linear_wq_dequant_bytes(wbits, M, K, N) + 255) & ~static_cast<size_t>(255);
    if (workspace && dq && workspace_bytes >= dq) {
        char *b = static_cast<char *>(workspace);
        return WqWorkspace{b, dq, SlabWs{reinterpret_cast<float *>(b + dq), (workspace_bytes - dq) / sizeof(float)}};
    }
    return WqWorkspace{nullptr, 0, SlabWs{static_cast<float *>(workspace), workspace_bytes / sizeof(float)}};
}

extern "C" int llmie_linear_w8a16(const void *x, const int8_t *wq, const void *scale, void *y, int M, int K, int N,
                                  const void *bias, const void *residual, void *workspace, size_t workspace_bytes,
                                  llmie_stream stream) {
    LLMIE_REQUIRE(x && wq && scale && y, "linear_w8a16: NULL pointer");
    LLMIE_REQUIRE(M > 0 && K > 0 && N > 0, "linear_w8a16: bad shape");
    LLMIE_REQUIRE(reinterpret_cast<uintptr_t>(workspace) % 16 == 0, "linear_w8a16: workspace must be 16-byte aligned");
    // workspace = [fp16 image of W (prefill-sized M without an in-kernel form) | split-K slabs], as llmie_linear_workspace_bytes sizes it
    const WqWorkspace w = wq_workspace(8, M, K, N, workspace, workspace_bytes);
    return linear_wq(8, (const half_t *)x, wq, (const half_t *)scale, (half_t *)y, M, K, N, 0, EPI_NONE,
                     (const half_t *)bias, (const half_t *)residual, nullptr, nullptr, 0.f, w.slabs, as_stream(stream), w.deq, w.deq_bytes);
}

extern "C" int llmie_linear_w4a16(const void *x, const uint8_t *wq, const void *scale, void *y, int M, int K, int N,
                                  int group, const void *bias, const void *residual, void *workspace, size_t workspace_bytes,
                                  llmie_stream stream) {
    LLMIE_REQUIRE(x && wq && scale && y, "linear_w4a16: NULL pointer");
    LLMIE_REQUIRE(M > 0 && K > 0 && N > 0, "linear_w4a16: bad shape");
    LLMIE_REQUIRE(group > 0 && group % 32 == 0 && K % group == 0, "linear_w4a16: group must be a multiple of 32 dividing K");
    LLMIE_REQUIRE(reinterpret_cast<uintptr_t>(workspace) % 16 == 0, "linear_w4a16: workspace must be 16-byte aligned");
    const WqWorkspace w = wq_workspace(4, M, K, N, workspace, workspace_bytes);
    return linear_wq(4, (const half_t *)x, wq, (const half_t *)scale, (half_t *)y, M, K, N, group, EPI_NONE,
                     (const half_t *)bias, (const half_t *)residual, nullptr, nullptr, 0.f, w.slabs, as_stream(stream), w.deq, w.deq_bytes);
}

extern "C" int llmie_quantize_w8(const void *w, int8_t *wq, void *scale, int N, int K, llmie_stream stream) {
    LLMIE_REQUIRE(w && wq && scale && N > 0 && K > 0, "quantize_w8: bad arguments");
    quantize_w8_kernel<<<N, 256, 0, as_stream(stream)>>>((const half_t *)w, wq, (half_t *)scale, K);
    return launch_status("quantize_w8");
}

extern "C" int llmie_quantize_w4(const void *w, uint8_t *wq, void *scale, int N, int K, int group,
                                 llmie_stream stream) {
    LLMIE_REQUIRE(w && wq && scale && N > 0 && K > 0, "quantize_w4: bad arguments");
    LLMIE_REQUIRE(group > 0 && group % 32 == 0 && K % group == 0, "quantize_w4: group must be a multiple of 32 dividing K");
    quantize_w4_kernel<<<static_cast<unsigned>(static_cast<size_t>(N) * (K / group)), 64, 0, as_stream(stream)>>>(
        (const half_t *)w, wq, (half_t *)scale, K, group);
    return launch_status("quantize_w4");
}

// fp8 (OCP e4m3fn, the gfx950 format -- not MI300's fnuz) linear for the decoder engine:
//   llmie_quantize_fp8      W fp16 [N,K] -> e4m3 [N,K] + fp32 per-row scale (amax/448)
//   llmie_linear_fp8        x fp16 [M,K] is quantised per token to e4m3 on the fly (scale amax/448), then
//                           y[m,n] = wscale[n] * xscale[m] * sum_k Wq[n,k] xq[m,k] on v_mfma_f32_16x16x32_fp8_fp8
//                           (fp32 accumulate): the split-K skinny kernel of gemm_kernels.cuh in its fp8 form (64 weight rows
//                           per workgroup, K sliced over workgroups, e4m3 activation tile shared through swizzled LDS, fp32
//                           slabs); M <= 8 takes the K-split GEMV with the same arithmetic (gemv_ksplit_kernel<.., FP8>).
// A 16-byte weight load per lane = 16 consecutive k of one row = two MFMA A operands with no conversion at all:
// the stream is half the bytes of fp16 and needs no de-quantisation ALU work.
#include "gemm_kernels.cuh"
#include "llmie_internal.h"

#include <cstdlib>

namespace llmie {

typedef long i64_t;

// one workgroup per row: amax -> scale = amax/448 (1 when the row is all zero), q = e4m3(w/scale)
template <typename ST>
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const half_t *__restrict__ w, uint8_t *__restrict__ q,
                                                                float *__restrict__ scale, int K) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const half_t *src = w + row * K;
    float amax = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) amax = fmaxf(amax, fabsf(to_f32(src[k])));
    amax = block_max<4>(amax, red);
    const float s = amax > 0.f ? amax / 448.0f : 1.0f;
    if (threadIdx.x == 0) scale[row] = s;
    unsigned int *dst = reinterpret_cast<unsigned int *>(q + row * K);
    for (int k4 = threadIdx.x; k4 < K / 4; k4 += 256) {
        const int k = 4 * k4;
        dst[k4] = pack4_e4m3(to_f32(src[k]) / s, to_f32(src[k + 1]) / s, to_f32(src[k + 2]) / s, to_f32(src[k + 3]) / s);
    }
}

// Same arithmetic with the row held in registers: 16-byte loads (R chunks of 8 halves per thread), one pass over memory
// (read 2K + write K bytes per row) -- the scalar two-pass form above ran at 1.2 TB/s on [4096, 11008] activations.
template <int R>
__global__ __launch_bounds__(256) void quantize_rows_fp8_vec_kernel(const half_t *__restrict__ w, uint8_t *__restrict__ q,
                                                                    float *__restrict__ scale, int K) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const half8_t *src = reinterpret_cast<const half8_t *>(w + row * K);
    const int nch = K / 8;
    half8_t v[R];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int c = threadIdx.x + 256 * i;
        v[i] = src[c < nch ? c : nch - 1];
        if (c < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(to_f32(v[i][e])));
        }
    }
    amax = block_max<4>(amax, red);
    const float s = amax > 0.f ? amax / 448.0f : 1.0f;
    if (threadIdx.x == 0) scale[row] = s;
    uint2 *dst = reinterpret_cast<uint2 *>(q + row * K);
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < nch)
            dst[c] = uint2{pack4_e4m3(to_f32(v[i][0]) / s, to_f32(v[i][1]) / s, to_f32(v[i][2]) / s, to_f32(v[i][3]) / s),
                           pack4_e4m3(to_f32(v[i][4]) / s, to_f32(v[i][5]) / s, to_f32(v[i][6]) / s, to_f32(v[i][7]) / s)};
    }
}

int quantize_rows_fp8(const half_t *x, uint8_t *xq, float *xscale, int M, int K, hipStream_t st) {
    const int per_thread = (K / 8 + 255) / 256;
    if (K % 8 == 0 && per_thread <= 8 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(xq) % 8 == 0) {
        if (per_thread <= 2) quantize_rows_fp8_vec_kernel<2><<<M, 256, 0, st>>>(x, xq, xscale, K);
        else if (per_thread <= 4) quantize_rows_fp8_vec_kernel<4><<<M, 256, 0, st>>>(x, xq, xscale, K);
        else if (per_thread <= 6) quantize_rows_fp8_vec_kernel<6><<<M, 256, 0, st>>>(x, xq, xscale, K);
        else quantize_rows_fp8_vec_kernel<8><<<M, 256, 0, st>>>(x, xq, xscale, K);
    } else {
        quantize_rows_fp8_kernel<float><<<M, 256, 0, st>>>(x, xq, xscale, K);
    }
    return launch_status("quantize_rows_fp8");
}

int linear_fp8_gemv(const half_t *x, const uint8_t *wq, const float *wscale, half_t *y, int M, int K, int N, int epi,
                    const half_t *bias, const half_t *residual, const half_t *gamma, const half_t *pre_bias, float eps,
                    hipStream_t st) {
    if (!ksplit_eligible(M, K, 8) || (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wq)) % 16 ||
        (gamma && (reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(pre_bias)) % 16)) {
        set_error("linear_fp8: shape M=%d K=%d not on the GEMV path", M, K);
        return LLMIE_ERR_UNSUPPORTED;
    }
    const GemvArgs a{x, wq, y, K, N, bias, residual, gamma, pre_bias, eps, epi, gamma ? 1 : 0, wscale, 0};
    if (!gemv_fp8_launch(M, a, st)) {
        set_error("linear_fp8: no GEMV instantiation for M=%d K=%d", M, K);
        return LLMIE_ERR_UNSUPPORTED;
    }
    return launch_status("linear_fp8(gemv)");
}

}  // namespace llmie

using namespace llmie;

static size_t fp8_align(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

static size_t fp8_act_bytes(int M, int K) {   // quantised activations + per-token scales
    return fp8_align(static_cast<size_t>(M) * K) + fp8_align(static_cast<size_t>(M) * sizeof(float));
}
extern "C" size_t llmie_linear_fp8_workspace_bytes(int M, int K, int N) {
    if (M <= 0 || K <= 0 || N < 0) return 0;
    // quantised activations + per-token scales, then the fp32 slabs of the split-K form (8 < M, below the tiled GEMM's sizes);
    // N = 0: the activation part only (llmie_linear_fp8_swiglu)
    return fp8_act_bytes(M, K) + (N > 0 && M > 8 ? fp8_align(linear_splitk_ws_floats(WF_FP8, M, K, N) * sizeof(float)) : 0);
}

extern "C" int llmie_quantize_fp8(const void *w, uint8_t *wq, float *scale, int N, int K, llmie_stream stream) {
    LLMIE_REQUIRE(w && wq && scale && N > 0 && K > 0 && K % 4 == 0, "quantize_fp8: bad arguments (K %% 4 == 0)");
    return quantize_rows_fp8((const half_t *)w, wq, scale, N, K, as_stream(stream));
}

namespace llmie {
static bool fp8_tiled(const void *w_scale, const void *bias, const void *residual, int M, int K, int N) {
    if (!(M > 8 && K % 128 == 0 && N % 4 == 0 && reinterpret_cast<uintptr_t>(w_scale) % 16 == 0 &&
          (reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual)) % 8 == 0))
        return false;
    if (gemm256_fills(M, N)) return true;
    // (round 3) a grid that does not fill the chip against split-K passes of 128 rows, by the time model fitted to the sweep
    // (tools/dev/s2_run27.sh; us): O / down of a 7B layer 41 / 90 tiled at any row count, 18.9 / 22.3 per pass -- at 768 tokens
    // the six passes took 113 / 134 us
    if (M <= 128 || static_cast<size_t>(N + 512) * K >= (size_t{1} << 32)) return false;
    const int tiles2 = ((M + 255) / 256) * ((N + 127) / 128);
    const float t_tiles = static_cast<float>((tiles2 + 255) / 256) * (K / 128) * 1.15f;
    const float t_passes = ((M + 127) / 128) * (static_cast<float>(N) * K / 8.3e6f + 17.f);
    return t_tiles < t_passes;
}
// llmie_linear_fp8 with its two scratch areas apart: `act_ws` (>= llmie_linear_fp8_workspace_bytes(M, K, 0), 256-byte aligned)
// receives the quantised activations, `slabs` the split-K partial sums (engine: one slab area serves every projection)
int linear_fp8(const half_t *x, const uint8_t *w_fp8, const float *w_scale, half_t *y, int M, int K, int N, const half_t *bias,
               const half_t *residual, void *act_ws, size_t act_ws_bytes, SlabWs slabs, hipStream_t st) {
    const bool tiled = fp8_tiled(w_scale, bias, residual, M, K, N);
    if ((!tiled && (K % 256 != 0 || K < 512)) || reinterpret_cast<uintptr_t>(w_fp8) % 16 || reinterpret_cast<uintptr_t>(act_ws) % 256) {
        set_error("linear_fp8: needs K %% 256 == 0, K >= 512 (K %% 128 == 0 for prefill-sized M x N), 16-byte aligned weights, "
                  "256-byte aligned workspace");
        return LLMIE_ERR_UNSUPPORTED;
    }
    if (act_ws_bytes < fp8_act_bytes(M, K)) {
        set_error("linear_fp8: workspace too small (%zu < %zu bytes)", act_ws_bytes, fp8_act_bytes(M, K));
        return LLMIE_ERR_WORKSPACE;
    }
    if (M <= 8 && ksplit_eligible(M, K, 8) && reinterpret_cast<uintptr_t>(x) % 16 == 0)
        return linear_fp8_gemv(x, w_fp8, w_scale, y, M, K, N, EPI_NONE, bias, residual, nullptr, nullptr, 0.f, st);
    uint8_t *xq = static_cast<uint8_t *>(act_ws);
    float *xscale = reinterpret_cast<float *>(xq + fp8_align(static_cast<size_t>(M) * K));
    int rc = quantize_rows_fp8(x, xq, xscale, M, K, st);
    if (rc) return rc;
    if (tiled) {
        // prefill-sized: MFMA-bound tiled GEMM on v_mfma_scale_f32_16x16x128_f8f6f4
        gemm256_launch(true, xq, w_fp8, y, M, N, K, bias, residual, xscale, w_scale, st);
        return launch_status("linear_fp8(gemm256)");
    }
    for (int m0 = 0; m0 < M; m0 += 128) {
        const int mc = M - m0 < 128 ? M - m0 : 128;
        SplitKSlabs sk;
        rc = linear_splitk_partial(WF_FP8, xq + static_cast<size_t>(m0) * K, w_fp8, mc, K, N, st, &sk, slabs);
        if (rc) return rc;
        rc = splitk_finalize(sk, SlabScale{nullptr, w_scale, xscale + m0}, y + static_cast<size_t>(m0) * N, EPI_NONE, bias,
                             residual ? residual + static_cast<size_t>(m0) * N : nullptr, st);
        if (rc) return rc;
    }
    return launch_status("linear_fp8");
}
}  // namespace llmie

extern "C" int llmie_linear_fp8(const void *x, const uint8_t *w_fp8, const float *w_scale, void *y, int M, int K, int N,
                                const void *bias, const void *residual, void *workspace, size_t workspace_bytes,
                                llmie_stream stream) {
    LLMIE_REQUIRE(x && w_fp8 && w_scale && y && workspace, "linear_fp8: NULL pointer");
    LLMIE_REQUIRE(M > 0 && K > 0 && N > 0, "linear_fp8: bad shape");
    const size_t act = fp8_act_bytes(M, K);
    if (workspace_bytes < act) {
        set_error("linear_fp8: workspace too small (%zu < %zu bytes)", workspace_bytes, llmie_linear_fp8_workspace_bytes(M, K, N));
        return LLMIE_ERR_WORKSPACE;
    }
    const SlabWs slabs{reinterpret_cast<float *>(static_cast<char *>(workspace) + act), (workspace_bytes - act) / sizeof(float)};
    return linear_fp8((const half_t *)x, w_fp8, w_scale, (half_t *)y, M, K, N, (const half_t *)bias, (const half_t *)residual,
                      workspace, act, slabs, as_stream(stream));
}

extern "C" int llmie_linear_fp8_swiglu(const void *x, const uint8_t *w_fp8, const float *w_scale, void *y, int M, int K,
                                       int two_inter, void *workspace, size_t workspace_bytes, llmie_stream stream) {
    LLMIE_REQUIRE(x && w_fp8 && w_scale && y && workspace, "linear_fp8_swiglu: NULL pointer");
    LLMIE_REQUIRE(M > 0 && K > 0 && two_inter > 0 && two_inter % 2 == 0, "linear_fp8_swiglu: bad shape");
    if (!gemm256_swiglu_fills(M, two_inter) || K % 128 != 0 || reinterpret_cast<uintptr_t>(w_fp8) % 16 ||
        reinterpret_cast<uintptr_t>(workspace) % 256 || reinterpret_cast<uintptr_t>(y) % 8)
        LLMIE_UNSUPPORTED("linear_fp8_swiglu: prefill-sized shapes only (>= 192 tiles of 256 tokens x 128 columns, K %% 128 == 0, "
                          "two_inter %% 8 == 0); use llmie_linear_fp8 + llmie_silu_and_mul otherwise");
    if (workspace_bytes < llmie_linear_fp8_workspace_bytes(M, K, 0)) {
        set_error("linear_fp8_swiglu: workspace too small");
        return LLMIE_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    uint8_t *xq = static_cast<uint8_t *>(workspace);
    float *xscale = reinterpret_cast<float *>(xq + fp8_align(static_cast<size_t>(M) * K));
    const int rc = quantize_rows_fp8((const half_t *)x, xq, xscale, M, K, st);
    if (rc) return rc;
    gemm256_swiglu_launch(true, xq, w_fp8, (half_t *)y, M, two_inter, K, xscale, w_scale, st);
    return launch_status("linear_fp8_swiglu");
}


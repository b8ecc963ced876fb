// fp8 (OCP e4m3fn, the gfx950 format -- not MI300's fnuz) linear for the decoder engine:
//   llmie_quantize_fp8      W fp16 [N,K] -> e4m3 [N,K] + fp32 per-row scale (amax/448)
//   llmie_linear_fp8        x fp16 [M,K] is quantised per token to e4m3 on the fly (scale amax/448), then
//                           y[m,n] = wscale[n] * xscale[m] * sum_k Wq[n,k] xq[m,k] on v_mfma_f32_16x16x32_fp8_fp8
//                           (fp32 accumulate) with the split-K skinny structure of gemm_kernels.cuh: 64 weight rows per
//                           workgroup, K sliced over workgroups, activation tile shared through swizzled LDS, fp32 slabs.
// A 16-byte weight load per lane = 16 consecutive k of one row = two MFMA A operands with no conversion at all:
// the stream is half the bytes of fp16 and needs no de-quantisation ALU work.
#include "gemm_kernels.cuh"
#include "llmie_internal.h"

namespace llmie {

typedef long i64_t;

__device__ __forceinline__ unsigned int pack4_e4m3(float a, float b, float c, float d) {
    // saturate to the e4m3fn range first: the conversion would produce NaN past 448
    a = fminf(fmaxf(a, -448.f), 448.f);
    b = fminf(fmaxf(b, -448.f), 448.f);
    c = fminf(fmaxf(c, -448.f), 448.f);
    d = fminf(fmaxf(d, -448.f), 448.f);
    int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return static_cast<unsigned int>(v);
}

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

template <int MT>
__global__ __launch_bounds__(256) void skinny_splitk_fp8_kernel(const uint8_t *__restrict__ xq, const uint8_t *__restrict__ Wq,
                                                                float *__restrict__ slab, int M, int K, int N, int KS,
                                                                int blocks_per_slice) {
    constexpr int BK = 256;          // 4 weight loads per lane per sub-block, 64 k each
    constexpr int ROWS = 16 * MT;
    constexpr int CPR = BK / 16;     // 16-byte chunks per tile row
    constexpr int XCH = (ROWS * CPR + 255) / 256;
    __shared__ __attribute__((aligned(16))) uint8_t xs[2][ROWS * BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int tile = blockIdx.x / KS, ks = blockIdx.x - tile * KS;
    const int n0 = tile * 64 + wave * 16;
    const int nrow = min(n0 + r, N - 1);
    const uint8_t *wp = Wq + static_cast<size_t>(nrow) * K + 16 * q;
    const int nblocks = K / BK;
    const int b_begin = ks * blocks_per_slice, b_end = min(nblocks, b_begin + blocks_per_slice);
    uint4_t a_cur[4], a_nxt[4], xr[XCH];
    auto load_a = [&](int blk, uint4_t(&a)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = load_nt(reinterpret_cast<const uint4_t *>(wp + (static_cast<size_t>(blk) * 4 + u) * 64));
    };
    auto load_x = [&](int blk) {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int id = min(tid + 256 * i, ROWS * CPR - 1), row = id / CPR, ch = id - row * CPR;
            xr[i] = *reinterpret_cast<const uint4_t *>(xq + static_cast<size_t>(min(row, M - 1)) * K + static_cast<size_t>(blk) * BK + ch * 16);
        }
    };
    auto store_x = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int id = tid + 256 * i;
            if (id < ROWS * CPR) {
                const int row = id / CPR, ch = id - row * CPR;
                *reinterpret_cast<uint4_t *>(&xs[buf][row * BK + ((ch ^ (row & 15)) << 4)]) = xr[i];
            }
        }
    };
    floatx4 acc[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    if (b_begin < b_end) {
        load_a(b_begin, a_cur);
        load_x(b_begin);
        store_x(0);
    }
    __syncthreads();
    for (int blk = b_begin, it = 0; blk < b_end; ++blk, ++it) {
        const int buf = it & 1;
        const bool more = blk + 1 < b_end;
        if (more) {
            load_a(blk + 1, a_nxt);
            load_x(blk + 1);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const i64_t a0 = (static_cast<i64_t>(a_cur[u][1]) << 32) | a_cur[u][0];
            const i64_t a1 = (static_cast<i64_t>(a_cur[u][3]) << 32) | a_cur[u][2];
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int row = 16 * j + r, ch = u * 4 + q;  // the lane's 16 consecutive k of this 64-wide step
                const uint4_t bv = *reinterpret_cast<const uint4_t *>(&xs[buf][row * BK + ((ch ^ (row & 15)) << 4)]);
                const i64_t b0 = (static_cast<i64_t>(bv[1]) << 32) | bv[0];
                const i64_t b1 = (static_cast<i64_t>(bv[3]) << 32) | bv[2];
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, b0, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a1, b1, acc[j], 0, 0, 0);
            }
        }
        if (more) {
            store_x(buf ^ 1);
#pragma unroll
            for (int u = 0; u < 4; ++u) a_cur[u] = a_nxt[u];
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = 16 * j + r, n = n0 + 4 * q;
        if (m < M && n < N) {
            float *dst = slab + (static_cast<size_t>(ks) * M + m) * N + n;
            if (n + 3 < N && (N & 3) == 0) {
                *reinterpret_cast<floatx4 *>(dst) = acc[j];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < N) dst[e] = acc[j][e];
            }
        }
    }
}

__global__ __launch_bounds__(256) void fp8_finalize_kernel(const float *__restrict__ slab, half_t *y, int M, int N, int KS,
                                                           const float *__restrict__ wscale, const float *__restrict__ xscale,
                                                           const half_t *__restrict__ bias, const half_t *residual) {
    const size_t total = static_cast<size_t>(M) * N;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * 256) {
        const int m = static_cast<int>(i / N), n = static_cast<int>(i - static_cast<size_t>(m) * N);
        float v = 0.f;
        for (int k = 0; k < KS; ++k) v += slab[static_cast<size_t>(k) * total + i];
        v *= wscale[n] * xscale[m];
        if (bias) v += to_f32(bias[n]);
        if (residual) v += to_f32(residual[i]);
        y[i] = from_f32<half_t>(v);
    }
}

}  // namespace llmie

using namespace llmie;

static size_t fp8_align(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

extern "C" size_t llmie_linear_fp8_workspace_bytes(int M, int K) {
    if (M <= 0 || K <= 0) return 0;
    // quantised activations + per-token scales + split-K slabs (16 slices x 64 tokens x N is not known here: the slab
    // part is sized for N <= 32768)
    return fp8_align(static_cast<size_t>(M) * K) + fp8_align(static_cast<size_t>(M) * sizeof(float)) +
           fp8_align(static_cast<size_t>(16) * 64 * 32768 * sizeof(float));
}

extern "C" int llmie_quantize_fp8(const void *w, uint8_t *wq, float *scale, int N, int K, llmie_stream stream) {
    LLMIE_REQUIRE(w && wq && scale && N > 0 && K > 0 && K % 4 == 0, "quantize_fp8: bad arguments (K %% 4 == 0)");
    quantize_rows_fp8_kernel<float><<<N, 256, 0, as_stream(stream)>>>((const half_t *)w, wq, scale, K);
    return launch_status("quantize_fp8");
}

extern "C" int llmie_linear_fp8(const void *x, const uint8_t *w_fp8, const float *w_scale, void *y, int M, int K, int N,
                                const void *bias, const void *residual, void *workspace, size_t workspace_bytes,
                                llmie_stream stream) {
    LLMIE_REQUIRE(x && w_fp8 && w_scale && y && workspace, "linear_fp8: NULL pointer");
    LLMIE_REQUIRE(M > 0 && K > 0 && N > 0, "linear_fp8: bad shape");
    if (K % 256 != 0 || N > 32768 || reinterpret_cast<uintptr_t>(w_fp8) % 16 || reinterpret_cast<uintptr_t>(workspace) % 256)
        LLMIE_UNSUPPORTED("linear_fp8: needs K %% 256 == 0, N <= 32768, 16-byte aligned weights, 256-byte aligned workspace");
    if (workspace_bytes < llmie_linear_fp8_workspace_bytes(M, K)) {
        set_error("linear_fp8: workspace too small");
        return LLMIE_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    uint8_t *xq = static_cast<uint8_t *>(workspace);
    float *xscale = reinterpret_cast<float *>(xq + fp8_align(static_cast<size_t>(M) * K));
    float *slab = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(xscale) + fp8_align(static_cast<size_t>(M) * sizeof(float)));
    quantize_rows_fp8_kernel<float><<<M, 256, 0, st>>>((const half_t *)x, xq, xscale, K);
    const int tiles = (N + 63) / 64, total_blocks = K / 256;
    int KS = 1;
    while (KS < 16 && tiles * KS < 512 && total_blocks / (KS * 2) >= 2) KS *= 2;
    const int spp = (total_blocks + KS - 1) / KS;
    for (int m0 = 0; m0 < M; m0 += 64) {
        const int mc = M - m0 < 64 ? M - m0 : 64;
        const uint8_t *xs = xq + static_cast<size_t>(m0) * K;
        const dim3 grid(tiles * KS);
        switch ((mc + 15) / 16) {
            case 1: skinny_splitk_fp8_kernel<1><<<grid, 256, 0, st>>>(xs, w_fp8, slab, mc, K, N, KS, spp); break;
            case 2: skinny_splitk_fp8_kernel<2><<<grid, 256, 0, st>>>(xs, w_fp8, slab, mc, K, N, KS, spp); break;
            case 3: skinny_splitk_fp8_kernel<3><<<grid, 256, 0, st>>>(xs, w_fp8, slab, mc, K, N, KS, spp); break;
            default: skinny_splitk_fp8_kernel<4><<<grid, 256, 0, st>>>(xs, w_fp8, slab, mc, K, N, KS, spp); break;
        }
        const size_t total = static_cast<size_t>(mc) * N;
        int fgrid = static_cast<int>((total + 255) / 256);
        if (fgrid > 2048) fgrid = 2048;
        fp8_finalize_kernel<<<fgrid, 256, 0, st>>>(slab, (half_t *)y + static_cast<size_t>(m0) * N, mc, N, KS, w_scale, xscale + m0,
                                                   (const half_t *)bias,
                                                   residual ? (const half_t *)residual + static_cast<size_t>(m0) * N : nullptr);
    }
    return launch_status("linear_fp8");
}

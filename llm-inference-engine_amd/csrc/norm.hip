// RMSNorm family for gfx950: llmie_rmsnorm, llmie_fused_add_bias_residual_rmsnorm,
// llmie_add_residual.  One workgroup per token row; the row stays in registers between
// the sum-of-squares pass and the scale pass (one HBM read, one write per tensor);
// 16-byte accesses; wave64 butterfly + LDS cross-wave reduction.
//
// Semantics follow the reference fp32 kernels (src/kernels/rmsnorm.cu:35-80,
// add_residual_and_rmsnorm.cu:43-121, add_residual.cu:51-76); the reference's fp16
// variants are defective and are NOT reproduced: fp16 uses the same
// math with fp32 accumulation and one rounding per stored element.
#include "device_utils.cuh"
#include "llmie_internal.h"

#include <cstdlib>

namespace llmie {

template <typename T, int BLOCK, int MAXV, bool FUSED>
__global__ __launch_bounds__(BLOCK) void rmsnorm_kernel(
    T *__restrict__ x,            // [T,H] in/out   (FUSED: decoder_out)
    T *__restrict__ resid,        // plain: out copy of x (nullable); FUSED: in/out residual (nullable)
    const T *__restrict__ bias,   // FUSED only, nullable [H]
    const T *__restrict__ gamma,  // [H] (FUSED: nullable -> no scaling, as the reference)
    float eps, int hidden) {
    using V = typename Vec16<T>::type;
    constexpr int N = Vec16<T>::n;
    constexpr int NW = BLOCK / 64;
    __shared__ float red[NW];
    const int nvec = hidden / N;
    const size_t row = static_cast<size_t>(blockIdx.x) * hidden;
    V *xv = reinterpret_cast<V *>(x + row);
    V *rv = resid ? reinterpret_cast<V *>(resid + row) : nullptr;
    const V *bv = reinterpret_cast<const V *>(bias);
    const V *gv = reinterpret_cast<const V *>(gamma);

    V keep[MAXV];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = threadIdx.x + j * BLOCK;
        if (i < nvec) {
            V v = xv[i];
            if constexpr (FUSED) {
                if (rv) {
                    V r = rv[i];
#pragma unroll
                    for (int e = 0; e < N; ++e) v[e] = from_f32<T>(to_f32(v[e]) + to_f32(r[e]));
                    rv[i] = v;  // residual := out + residual (before the bias add, as the reference)
                }
                if (bv) {
                    V b = bv[i];
#pragma unroll
                    for (int e = 0; e < N; ++e) v[e] = from_f32<T>(to_f32(v[e]) + to_f32(b[e]));
                }
            } else {
                if (rv) rv[i] = v;
            }
#pragma unroll
            for (int e = 0; e < N; ++e) ss += to_f32(v[e]) * to_f32(v[e]);
            keep[j] = v;
        }
    }
    ss = block_sum<NW>(ss, red);
    const float inv = rsqrtf(ss / static_cast<float>(hidden) + eps);
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = threadIdx.x + j * BLOCK;
        if (i < nvec) {
            V v = keep[j];
            if (gv) {
                V g = gv[i];
#pragma unroll
                for (int e = 0; e < N; ++e) v[e] = from_f32<T>(to_f32(v[e]) * to_f32(g[e]) * inv);
                xv[i] = v;
            } else if constexpr (FUSED) {
                xv[i] = v;  // reference: no gamma -> only the adds are applied
            }
        }
    }
}

// fp8 engines, prefill: the RMSNorm in front of a projection emits that projection's per-token e4m3 activations directly
// (scale = amax/448 of the fp16-rounded normalised row: bit-identical to rmsnorm_kernel followed by quantize_rows_fp8) -- the
// normalised fp16 row, which nothing else reads, is never written, and the quantisation launch disappears.
template <bool FUSED>
__global__ __launch_bounds__(256) void rmsnorm_quant_kernel(const half_t *__restrict__ x, half_t *__restrict__ resid,
                                                            const half_t *__restrict__ bias, const half_t *__restrict__ gamma,
                                                            float eps, int hidden, uint8_t *__restrict__ xq, float *__restrict__ xscale) {
    constexpr int MAXV = 4;
    __shared__ float red[4];
    const int nvec = hidden / 8;
    const size_t row = static_cast<size_t>(blockIdx.x) * hidden;
    const half8_t *xv = reinterpret_cast<const half8_t *>(x + row);
    half8_t *rv = resid ? reinterpret_cast<half8_t *>(resid + row) : nullptr;
    const half8_t *bv = reinterpret_cast<const half8_t *>(bias);
    const half8_t *gv = reinterpret_cast<const half8_t *>(gamma);
    half8_t keep[MAXV];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = threadIdx.x + j * 256;
        if (i < nvec) {
            half8_t v = xv[i];
            if constexpr (FUSED) {
                if (rv) {
                    const half8_t r = rv[i];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = from_f32<half_t>(to_f32(v[e]) + to_f32(r[e]));
                    rv[i] = v;
                }
                if (bv) {
                    const half8_t b = bv[i];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = from_f32<half_t>(to_f32(v[e]) + to_f32(b[e]));
                }
            } else {
                if (rv) rv[i] = v;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += to_f32(v[e]) * to_f32(v[e]);
            keep[j] = v;
        }
    }
    ss = block_sum<4>(ss, red);
    const float inv = rsqrtf(ss / static_cast<float>(hidden) + eps);
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = threadIdx.x + j * 256;
        if (i < nvec) {
            const half8_t g = gv[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                keep[j][e] = from_f32<half_t>(to_f32(keep[j][e]) * to_f32(g[e]) * inv);
                amax = fmaxf(amax, fabsf(to_f32(keep[j][e])));
            }
        }
    }
    amax = block_max<4>(amax, red);
    const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
    if (threadIdx.x == 0) xscale[blockIdx.x] = sc;
    uint2 *dst = reinterpret_cast<uint2 *>(xq + row);
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = threadIdx.x + j * 256;
        if (i < nvec) {
            const half8_t v = keep[j];
            dst[i] = uint2{pack4_e4m3(to_f32(v[0]) / sc, to_f32(v[1]) / sc, to_f32(v[2]) / sc, to_f32(v[3]) / sc),
                           pack4_e4m3(to_f32(v[4]) / sc, to_f32(v[5]) / sc, to_f32(v[6]) / sc, to_f32(v[7]) / sc)};
        }
    }
}

bool rmsnorm_quant_eligible(int hidden) {
    static const bool off = getenv("LLMIE_NO_NORM_QUANT") != nullptr;
    return !off && hidden % 8 == 0 && hidden / 8 <= 1024;
}
// fused == false: resid = x (copy, nullable); fused == true: x += resid; resid = x; x += bias -- then e4m3(norm(x) * gamma)
int rmsnorm_quant_f16(const half_t *x, half_t *resid, const half_t *bias, const half_t *gamma, float eps, int tokens, int hidden,
                      bool fused, uint8_t *xq, float *xscale, hipStream_t st) {
    if (!gamma || !xq || !xscale || hidden % 8 || hidden / 8 > 1024 ||
        (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(resid) | reinterpret_cast<uintptr_t>(bias) |
         reinterpret_cast<uintptr_t>(gamma)) % 16 || reinterpret_cast<uintptr_t>(xq) % 8) {
        set_error("rmsnorm_quant: unsupported shape / alignment (hidden %d)", hidden);
        return LLMIE_ERR_UNSUPPORTED;
    }
    if (fused) rmsnorm_quant_kernel<true><<<tokens, 256, 0, st>>>(x, resid, bias, gamma, eps, hidden, xq, xscale);
    else rmsnorm_quant_kernel<false><<<tokens, 256, 0, st>>>(x, resid, nullptr, gamma, eps, hidden, xq, xscale);
    return launch_status("rmsnorm_quant");
}

// Any hidden size (unaligned / huge rows): scalar, two passes (second pass hits L2).
template <typename T, bool FUSED>
__global__ __launch_bounds__(256) void rmsnorm_generic_kernel(
    T *__restrict__ x, T *__restrict__ resid, const T *__restrict__ bias,
    const T *__restrict__ gamma, float eps, int hidden) {
    __shared__ float red[4];
    const size_t row = static_cast<size_t>(blockIdx.x) * hidden;
    float ss = 0.f;
    for (int i = threadIdx.x; i < hidden; i += 256) {
        T v = x[row + i];
        if constexpr (FUSED) {
            if (resid) {
                v = from_f32<T>(to_f32(v) + to_f32(resid[row + i]));
                resid[row + i] = v;
            }
            if (bias) v = from_f32<T>(to_f32(v) + to_f32(bias[i]));
            x[row + i] = v;
        } else {
            if (resid) resid[row + i] = v;
        }
        ss += to_f32(v) * to_f32(v);
    }
    ss = block_sum<4>(ss, red);
    const float inv = rsqrtf(ss / static_cast<float>(hidden) + eps);
    if (gamma)
        for (int i = threadIdx.x; i < hidden; i += 256)
            x[row + i] = from_f32<T>(to_f32(x[row + i]) * to_f32(gamma[i]) * inv);
}

// Out-of-place RMSNorm, fp16 (round 3, prefill): y = x * gamma * rsqrt(mean(x^2) + eps) with x left as it is -- the residual stream
// stays un-normalised in its own buffer (the O / down projections add into it in their epilogues), so a norm moves 2 x |x| bytes
// instead of the 3-4 x of rmsnorm.cu's "copy the residual out, normalise in place" form.  Same arithmetic per element as
// rmsnorm_kernel (fp32 sum of squares of the fp16 values, one rounding).
__global__ __launch_bounds__(256) void rmsnorm_oop_kernel(const half_t *__restrict__ x, half_t *__restrict__ y,
                                                          const half_t *__restrict__ gamma, float eps, int hidden) {
    constexpr int MAXV = 4;
    __shared__ float red[4];
    const int nvec = hidden / 8;
    const size_t row = static_cast<size_t>(blockIdx.x) * hidden;
    const half8_t *xv = reinterpret_cast<const half8_t *>(x + row);
    half8_t *yv = reinterpret_cast<half8_t *>(y + row);
    const half8_t *gv = reinterpret_cast<const half8_t *>(gamma);
    half8_t keep[MAXV];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = threadIdx.x + j * 256;
        if (i < nvec) {
            keep[j] = xv[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += to_f32(keep[j][e]) * to_f32(keep[j][e]);
        }
    }
    ss = block_sum<4>(ss, red);
    const float inv = rsqrtf(ss / static_cast<float>(hidden) + eps);
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = threadIdx.x + j * 256;
        if (i < nvec) {
            const half8_t g = gv[i];
            half8_t v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = from_f32<half_t>(to_f32(keep[j][e]) * to_f32(g[e]) * inv);
            yv[i] = v;
        }
    }
}
// One WAVE per row (round 3): hidden <= 4096 -> at most 8 sixteen-byte vectors per lane, all loads of the row in flight at once,
// the sum of squares reduced on DPP -- no LDS, no barrier, four rows per workgroup (a quarter of the workgroups of the
// row-per-workgroup form above, whose two barriers and 2048 eight-KiB workgroups made a 32 MB pass take 9.6 us at 2048 tokens).
template <int NV>
__global__ __launch_bounds__(256) void rmsnorm_oop_wave_kernel(const half_t *__restrict__ x, half_t *__restrict__ y,
                                                               const half_t *__restrict__ gamma, float eps, int hidden, int tokens) {
    const int lane = threadIdx.x & 63, row_i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row_i >= tokens) return;   // (whole waves: no barrier in this kernel)
    const int nvec = hidden / 8;
    const size_t row = static_cast<size_t>(row_i) * hidden;
    const half8_t *xv = reinterpret_cast<const half8_t *>(x + row);
    half8_t *yv = reinterpret_cast<half8_t *>(y + row);
    const half8_t *gv = reinterpret_cast<const half8_t *>(gamma);
    half8_t keep[NV], g[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = min(lane + j * 64, nvec - 1);   // (clamped: a vector past the row end is loaded twice and not counted)
        keep[j] = xv[i];
        g[j] = gv[i];
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (lane + j * 64 < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += to_f32(keep[j][e]) * to_f32(keep[j][e]);
        }
    ss = wave_sum(ss);
    const float inv = rsqrtf(ss / static_cast<float>(hidden) + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (lane + j * 64 < nvec) {
            half8_t v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = from_f32<half_t>(to_f32(keep[j][e]) * to_f32(g[j][e]) * inv);
            yv[lane + j * 64] = v;
        }
}
bool rmsnorm_oop_eligible(int hidden) { return hidden % 8 == 0 && hidden / 8 <= 1024; }
int rmsnorm_oop_f16(const half_t *x, half_t *y, const half_t *gamma, float eps, int tokens, int hidden, hipStream_t st) {
    if (!rmsnorm_oop_eligible(hidden) || (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gamma)) % 16) {
        set_error("rmsnorm (out of place): hidden %d / alignment not supported", hidden);
        return LLMIE_ERR_UNSUPPORTED;
    }
    const int nvec = hidden / 8;
    if (nvec <= 256) rmsnorm_oop_wave_kernel<4><<<(tokens + 3) / 4, 256, 0, st>>>(x, y, gamma, eps, hidden, tokens);
    else if (nvec <= 512) rmsnorm_oop_wave_kernel<8><<<(tokens + 3) / 4, 256, 0, st>>>(x, y, gamma, eps, hidden, tokens);
    else rmsnorm_oop_kernel<<<tokens, 256, 0, st>>>(x, y, gamma, eps, hidden);
    return launch_status("rmsnorm(out of place)");
}

template <typename T, bool FUSED>
static int launch_norm(T *x, T *resid, const T *bias, const T *gamma, float eps, int tokens,
                       int hidden, hipStream_t st) {
    constexpr int N = Vec16<T>::n;
    const bool aligned = (hidden % N == 0) &&
                         ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(resid) |
                           reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(gamma)) % 16 == 0);
    const int nvec = hidden / N;
    dim3 grid(tokens);
    if (aligned && nvec <= 64) {
        rmsnorm_kernel<T, 64, 1, FUSED><<<grid, 64, 0, st>>>(x, resid, bias, gamma, eps, hidden);
    } else if (aligned && nvec <= 1024) {
        rmsnorm_kernel<T, 256, 4, FUSED><<<grid, 256, 0, st>>>(x, resid, bias, gamma, eps, hidden);
    } else if (aligned && nvec <= 4096) {
        rmsnorm_kernel<T, 1024, 4, FUSED><<<grid, 1024, 0, st>>>(x, resid, bias, gamma, eps, hidden);
    } else {
        rmsnorm_generic_kernel<T, FUSED><<<grid, 256, 0, st>>>(x, resid, bias, gamma, eps, hidden);
    }
    return launch_status(FUSED ? "fused_add_bias_residual_rmsnorm" : "rmsnorm");
}

// out += resid, 16-byte grid-stride
template <typename T>
__global__ __launch_bounds__(256) void add_residual_kernel(const T *__restrict__ resid,
                                                           T *__restrict__ out, size_t n) {
    using V = typename Vec16<T>::type;
    constexpr int N = Vec16<T>::n;
    const size_t nvec = n / N;
    const V *rv = reinterpret_cast<const V *>(resid);
    V *ov = reinterpret_cast<V *>(out);
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < nvec; i += static_cast<size_t>(gridDim.x) * 256) {
        V a = ov[i], b = rv[i];
#pragma unroll
        for (int e = 0; e < N; ++e) a[e] = from_f32<T>(to_f32(a[e]) + to_f32(b[e]));
        ov[i] = a;
    }
    // tail (n not a multiple of the vector width)
    for (size_t i = nvec * N + blockIdx.x * 256ull + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256)
        out[i] = from_f32<T>(to_f32(out[i]) + to_f32(resid[i]));
}

template <typename T>
__global__ __launch_bounds__(256) void add_residual_scalar_kernel(const T *__restrict__ resid,
                                                                  T *__restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256)
        out[i] = from_f32<T>(to_f32(out[i]) + to_f32(resid[i]));
}

template <typename T>
static int launch_add_residual(const T *resid, T *out, int tokens, int hidden, hipStream_t st) {
    const size_t n = static_cast<size_t>(tokens) * hidden;
    const bool aligned =
        ((reinterpret_cast<uintptr_t>(resid) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
    const size_t work = (n / Vec16<T>::n + 255) / 256;
    const int grid = static_cast<int>(work < 1 ? 1 : (work > 2048 ? 2048 : work));
    if (aligned)
        add_residual_kernel<T><<<grid, 256, 0, st>>>(resid, out, n);
    else
        add_residual_scalar_kernel<T><<<grid, 256, 0, st>>>(resid, out, n);
    return launch_status("add_residual");
}

}  // namespace llmie

using namespace llmie;

extern "C" int llmie_rmsnorm(void *x, void *resid, const void *gamma, float eps, int num_tokens,
                             int hidden, llmie_dtype dtype, llmie_stream stream) {
    LLMIE_REQUIRE(x && gamma, "rmsnorm: x and gamma must be non-NULL");
    LLMIE_REQUIRE(num_tokens > 0 && hidden > 0, "rmsnorm: bad shape [%d,%d]", num_tokens, hidden);
    if (dtype == LLMIE_F32)
        return launch_norm<float, false>((float *)x, (float *)resid, nullptr, (const float *)gamma, eps,
                                         num_tokens, hidden, as_stream(stream));
    if (dtype == LLMIE_F16)
        return launch_norm<half_t, false>((half_t *)x, (half_t *)resid, nullptr, (const half_t *)gamma,
                                          eps, num_tokens, hidden, as_stream(stream));
    LLMIE_UNSUPPORTED("rmsnorm: dtype %d", (int)dtype);
}

extern "C" int llmie_fused_add_bias_residual_rmsnorm(void *resid, void *out, const void *bias,
                                                     const void *gamma, float eps, int num_tokens,
                                                     int hidden, llmie_dtype dtype,
                                                     llmie_stream stream) {
    LLMIE_REQUIRE(out, "fused_add_bias_residual_rmsnorm: out must be non-NULL");
    LLMIE_REQUIRE(num_tokens > 0 && hidden > 0, "fused_add_bias_residual_rmsnorm: bad shape [%d,%d]",
                  num_tokens, hidden);
    if (dtype == LLMIE_F32)
        return launch_norm<float, true>((float *)out, (float *)resid, (const float *)bias,
                                        (const float *)gamma, eps, num_tokens, hidden, as_stream(stream));
    if (dtype == LLMIE_F16)
        return launch_norm<half_t, true>((half_t *)out, (half_t *)resid, (const half_t *)bias,
                                         (const half_t *)gamma, eps, num_tokens, hidden, as_stream(stream));
    LLMIE_UNSUPPORTED("fused_add_bias_residual_rmsnorm: dtype %d", (int)dtype);
}

extern "C" int llmie_add_residual(const void *resid, void *out, int num_tokens, int hidden,
                                  llmie_dtype dtype, llmie_stream stream) {
    LLMIE_REQUIRE(resid && out, "add_residual: NULL pointer");
    LLMIE_REQUIRE(num_tokens > 0 && hidden > 0, "add_residual: bad shape [%d,%d]", num_tokens, hidden);
    if (dtype == LLMIE_F32)
        return launch_add_residual<float>((const float *)resid, (float *)out, num_tokens, hidden, as_stream(stream));
    if (dtype == LLMIE_F16)
        return launch_add_residual<half_t>((const half_t *)resid, (half_t *)out, num_tokens, hidden, as_stream(stream));
    LLMIE_UNSUPPORTED("add_residual: dtype %d", (int)dtype);
}

// llmie_scale_mask_softmax (scale_and_mask_and_softmax.cu:64-341) and llmie_silu_and_mul
// (silu_and_mul.cu:25-82).
// Softmax: one wave64 per (b,h,q) row for k_len <= 2048 (row kept in registers: 32 values/lane),
// one 256-thread workgroup per row beyond; true row max (the reference's FLT_MIN floor is a
// defect), denominator + 1e-6 kept.
#include "device_utils.cuh"

namespace llmie {

template <typename T, int BLOCK, int PER>
__global__ __launch_bounds__(BLOCK == 64 ? 256 : BLOCK) void softmax_kernel(const T *__restrict__ qk, const T *__restrict__ mask,
                                                        T *__restrict__ out, float scale, int head_num,
                                                        int q_len, int k_len, size_t rows) {
    constexpr int NW = BLOCK / 64;
    __shared__ float red[NW];
    // rows are (b, h, q); BLOCK==64 packs 4 rows per 256-thread launch block via blockDim.y
    const size_t row = static_cast<size_t>(blockIdx.x) * blockDim.y + threadIdx.y;
    if (row >= rows) return;
    const int q = static_cast<int>(row % q_len);
    const size_t bh = row / q_len;
    const size_t b = bh / head_num;
    const T *src = qk + row * k_len;
    const T *m = mask + (b * q_len + q) * k_len;
    T *dst = out + row * k_len;
    float v[PER];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int k = threadIdx.x + j * BLOCK;
        if (k < k_len) {
            v[j] = scale * to_f32(src[k]) + (1.0f - to_f32(m[k])) * (-10000.0f);
            mx = fmaxf(mx, v[j]);
        }
    }
    mx = (NW == 1) ? wave_max(mx) : block_max<NW>(mx, red);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int k = threadIdx.x + j * BLOCK;
        if (k < k_len) {
            v[j] = expf(v[j] - mx);
            sum += v[j];
        }
    }
    sum = (NW == 1) ? wave_sum(sum) : block_sum<NW>(sum, red);
    const float inv = 1.0f / (sum + 1e-6f);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int k = threadIdx.x + j * BLOCK;
        if (k < k_len) dst[k] = from_f32<T>(v[j] * inv);
    }
}

// out[t,i] = silu(in[t,0,i]) * in[t,1,i]
template <typename T>
__global__ __launch_bounds__(256) void silu_mul_kernel(const T *__restrict__ in, T *__restrict__ out,
                                                       int inter, bool vec_ok) {
    const int t = blockIdx.y;
    const T *g = in + static_cast<size_t>(t) * 2 * inter;
    const T *u = g + inter;
    T *o = out + static_cast<size_t>(t) * inter;
    if (vec_ok) {
        using V = typename Vec16<T>::type;
        constexpr int N = Vec16<T>::n;
        const int nvec = inter / N;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < nvec; i += gridDim.x * 256) {
            V a = reinterpret_cast<const V *>(g)[i], b = reinterpret_cast<const V *>(u)[i], r;
#pragma unroll
            for (int e = 0; e < N; ++e) {
                const float x = to_f32(a[e]);
                r[e] = from_f32<T>((x / (1.0f + expf(-x))) * to_f32(b[e]));
            }
            reinterpret_cast<V *>(o)[i] = r;
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < inter; i += gridDim.x * 256) {
            const float x = to_f32(g[i]);
            o[i] = from_f32<T>((x / (1.0f + expf(-x))) * to_f32(u[i]));
        }
    }
}

template <typename T>
static int launch_softmax(const T *qk, const T *mask, T *out, float scale, int batch, int head_num,
                          int q_len, int k_len, hipStream_t st) {
    const size_t rows = static_cast<size_t>(batch) * head_num * q_len;
    if (k_len <= 256) {
        dim3 block(64, 4);
        softmax_kernel<T, 64, 4><<<static_cast<unsigned>((rows + 3) / 4), block, 0, st>>>(qk, mask, out, scale, head_num, q_len, k_len, rows);
    } else if (k_len <= 2048) {
        dim3 block(64, 4);
        softmax_kernel<T, 64, 32><<<static_cast<unsigned>((rows + 3) / 4), block, 0, st>>>(qk, mask, out, scale, head_num, q_len, k_len, rows);
    } else if (k_len <= 8192) {
        dim3 block(256, 1);
        softmax_kernel<T, 256, 32><<<static_cast<unsigned>(rows), block, 0, st>>>(qk, mask, out, scale, head_num, q_len, k_len, rows);
    } else if (k_len <= 32768) {
        dim3 block(1024, 1);
        softmax_kernel<T, 1024, 32><<<static_cast<unsigned>(rows), block, 0, st>>>(qk, mask, out, scale, head_num, q_len, k_len, rows);
    } else {
        set_error("scale_mask_softmax: k_len %d > 32768 not supported", k_len);
        return LLMIE_ERR_UNSUPPORTED;
    }
    return launch_status("scale_mask_softmax");
}

}  // namespace llmie

using namespace llmie;

extern "C" int llmie_scale_mask_softmax(const void *qk, const void *mask, void *out, float scale, int batch,
                                        int head_num, int q_len, int k_len, llmie_dtype dtype,
                                        llmie_stream stream) {
    LLMIE_REQUIRE(qk && mask && out, "scale_mask_softmax: NULL pointer");
    LLMIE_REQUIRE(batch > 0 && head_num > 0 && q_len > 0 && k_len > 0, "scale_mask_softmax: bad shape");
    if (dtype == LLMIE_F32)
        return launch_softmax<float>((const float *)qk, (const float *)mask, (float *)out, scale, batch,
                                     head_num, q_len, k_len, as_stream(stream));
    if (dtype == LLMIE_F16)
        return launch_softmax<half_t>((const half_t *)qk, (const half_t *)mask, (half_t *)out, scale, batch,
                                      head_num, q_len, k_len, as_stream(stream));
    LLMIE_UNSUPPORTED("scale_mask_softmax: dtype %d", (int)dtype);
}

extern "C" int llmie_silu_and_mul(const void *in, void *out, int num_tokens, int inter, llmie_dtype dtype,
                                  llmie_stream stream) {
    LLMIE_REQUIRE(in && out, "silu_and_mul: NULL pointer");
    LLMIE_REQUIRE(num_tokens > 0 && inter > 0, "silu_and_mul: bad shape");
    LLMIE_REQUIRE(num_tokens <= 65535, "silu_and_mul: num_tokens > 65535");
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
    int gx = (inter / 8 + 255) / 256;
    if (gx < 1) gx = 1;
    dim3 grid(gx, num_tokens);
    if (dtype == LLMIE_F32)
        silu_mul_kernel<float><<<grid, 256, 0, as_stream(stream)>>>((const float *)in, (float *)out, inter,
                                                                   aligned && inter % 4 == 0);
    else if (dtype == LLMIE_F16)
        silu_mul_kernel<half_t><<<grid, 256, 0, as_stream(stream)>>>((const half_t *)in, (half_t *)out, inter,
                                                                    aligned && inter % 8 == 0);
    else
        LLMIE_UNSUPPORTED("silu_and_mul: dtype %d", (int)dtype);
    return launch_status("silu_and_mul");
}

// RoPE kernels: llmie_rope_decode (rope.cu:4-98) and llmie_qkv_bias_transpose_rope
// (qkv_bias_and_rope.cu:5-138).  rotate-half pairing (d, d+hs/2); angle = pos / base^(2d/rot_dim)
// (rope_utils.cuh:6-19) evaluated in fp32 with the precise powf/sinf/cosf (positions reach
// thousands of radians: the fast approximations are not good enough).  The per-token cos/sin
// row is computed once per workgroup into LDS and shared by all heads.
#include "device_utils.cuh"

namespace llmie {

__device__ __forceinline__ void rope_cis(int d, int rotary_dim, float base, float pos, float &c, float &s) {
    const float ang = pos / powf(base, static_cast<float>(2 * d) / static_cast<float>(rotary_dim));
    c = cosf(ang);
    s = sinf(ang);
}

// Decode: in place on qkv[bs, nh+2kvh, hs]; q heads and k heads rotated once each.
template <typename T>
__global__ __launch_bounds__(256) void rope_decode_kernel(T *__restrict__ qkv, int head_num, int kv_head_num,
                                                          int head_size, int step,
                                                          const int32_t *__restrict__ step_dev,
                                                          int rotary_dim, float base) {
    extern __shared__ float cs[];  // [2 * half]
    const int half = head_size >> 1;
    const int nrot = min(half, rotary_dim >> 1);
    const int pos_i = (step_dev ? *step_dev : step) - 1;
    if (pos_i < 0) return;   // a corrupt device-resident step: leave the rows alone (whole workgroup)
    const float pos = static_cast<float>(pos_i);
    for (int d = threadIdx.x; d < nrot; d += 256) rope_cis(d, rotary_dim, base, pos, cs[2 * d], cs[2 * d + 1]);
    __syncthreads();
    const int b = blockIdx.x;
    const int rot_heads = head_num + kv_head_num;
    T *row = qkv + static_cast<size_t>(b) * (head_num + 2 * kv_head_num) * head_size;
    for (int i = threadIdx.x; i < rot_heads * nrot; i += 256) {
        const int h = i / nrot, d = i - h * nrot;
        T *x = row + static_cast<size_t>(h) * head_size;
        const float x0 = to_f32(x[d]), x1 = to_f32(x[d + half]);
        const float c = cs[2 * d], s = cs[2 * d + 1];
        x[d] = from_f32<T>(x0 * c - x1 * s);
        x[d + half] = from_f32<T>(x1 * c + x0 * s);
    }
}

// Prefill: one workgroup per packed token.
template <typename T>
__global__ __launch_bounds__(256) void qkv_transpose_rope_kernel(
    T *__restrict__ q, T *__restrict__ k, T *__restrict__ v, const T *__restrict__ qkv,
    const T *__restrict__ bias, const int32_t *__restrict__ padding_offset,
    const int32_t *__restrict__ history_len, int batch, int seq_len, int head_num, int kv_head_num,
    int head_size, int rotary_dim, float base) {
    extern __shared__ float cs[];
    const int t = blockIdx.x;
    const int dst_tok = t + padding_offset[t];
    const int b = dst_tok / seq_len, s_ = dst_tok % seq_len;
    if (b >= batch) return;  // corrupt padding offset: never write outside the padded buffers
    const int half = head_size >> 1;
    const int nrot = min(half, rotary_dim >> 1);
    const float pos = static_cast<float>(history_len[b] + s_);
    for (int d = threadIdx.x; d < nrot; d += 256) rope_cis(d, rotary_dim, base, pos, cs[2 * d], cs[2 * d + 1]);
    __syncthreads();
    const int heads = head_num + 2 * kv_head_num;
    const T *row = qkv + static_cast<size_t>(t) * heads * head_size;
    for (int i = threadIdx.x; i < heads * half; i += 256) {
        const int h = i / half, d = i - h * half;
        const T *src = row + static_cast<size_t>(h) * head_size;
        float x0 = to_f32(src[d]), x1 = to_f32(src[d + half]);
        if (bias) {
            x0 += to_f32(bias[static_cast<size_t>(h) * head_size + d]);
            x1 += to_f32(bias[static_cast<size_t>(h) * head_size + d + half]);
        }
        T *dst;
        bool rotate = true;
        if (h < head_num) {
            dst = q + ((static_cast<size_t>(b) * head_num + h) * seq_len + s_) * head_size;
        } else if (h < head_num + kv_head_num) {
            dst = k + ((static_cast<size_t>(b) * kv_head_num + (h - head_num)) * seq_len + s_) * head_size;
        } else {
            dst = v + ((static_cast<size_t>(b) * kv_head_num + (h - head_num - kv_head_num)) * seq_len + s_) * head_size;
            rotate = false;
        }
        if (rotate && d < nrot) {
            const float c = cs[2 * d], sn = cs[2 * d + 1];
            dst[d] = from_f32<T>(x0 * c - x1 * sn);
            dst[d + half] = from_f32<T>(x1 * c + x0 * sn);
        } else {
            dst[d] = from_f32<T>(x0);
            dst[d + half] = from_f32<T>(x1);
        }
    }
}

}  // namespace llmie

using namespace llmie;

extern "C" int llmie_rope_decode(void *qkv, int batch, int head_num, int kv_head_num, int head_size,
                                 int step, const int32_t *step_dev, int rotary_dim, float rotary_base,
                                 llmie_dtype dtype, llmie_stream stream) {
    LLMIE_REQUIRE(qkv, "rope_decode: NULL qkv");
    LLMIE_REQUIRE(batch > 0 && head_num > 0 && kv_head_num > 0 && head_size > 0 && head_size % 2 == 0,
                  "rope_decode: bad shape");
    LLMIE_REQUIRE(rotary_dim > 0 && rotary_dim % 2 == 0, "rope_decode: rotary_dim must be even and > 0");
    LLMIE_REQUIRE(step_dev || step >= 1, "rope_decode: step must be >= 1");
    const size_t lds = sizeof(float) * head_size;
    if (dtype == LLMIE_F32)
        rope_decode_kernel<float><<<batch, 256, lds, as_stream(stream)>>>((float *)qkv, head_num, kv_head_num,
                                                                         head_size, step, step_dev, rotary_dim, rotary_base);
    else if (dtype == LLMIE_F16)
        rope_decode_kernel<half_t><<<batch, 256, lds, as_stream(stream)>>>((half_t *)qkv, head_num, kv_head_num,
                                                                          head_size, step, step_dev, rotary_dim, rotary_base);
    else
        LLMIE_UNSUPPORTED("rope_decode: dtype %d", (int)dtype);
    return launch_status("rope_decode");
}

extern "C" int llmie_qkv_bias_transpose_rope(void *q, void *k, void *v, const void *qkv, const void *bias,
                                             const int32_t *padding_offset, const int32_t *history_len,
                                             int batch, int seq_len, int num_tokens, int head_num,
                                             int kv_head_num, int head_size, int rotary_dim,
                                             float rotary_base, llmie_dtype dtype, llmie_stream stream) {
    LLMIE_REQUIRE(q && k && v && qkv && padding_offset && history_len, "qkv_bias_transpose_rope: NULL pointer");
    LLMIE_REQUIRE(batch > 0 && seq_len > 0 && num_tokens > 0 && head_num > 0 && kv_head_num > 0 &&
                      head_size > 0 && head_size % 2 == 0, "qkv_bias_transpose_rope: bad shape");
    LLMIE_REQUIRE(rotary_dim > 0 && rotary_dim % 2 == 0, "qkv_bias_transpose_rope: rotary_dim must be even and > 0");
    LLMIE_REQUIRE(num_tokens <= batch * seq_len, "qkv_bias_transpose_rope: num_tokens > batch*seq_len");
    const size_t lds = sizeof(float) * head_size;
    if (dtype == LLMIE_F32)
        qkv_transpose_rope_kernel<float><<<num_tokens, 256, lds, as_stream(stream)>>>(
            (float *)q, (float *)k, (float *)v, (const float *)qkv, (const float *)bias, padding_offset,
            history_len, batch, seq_len, head_num, kv_head_num, head_size, rotary_dim, rotary_base);
    else if (dtype == LLMIE_F16)
        qkv_transpose_rope_kernel<half_t><<<num_tokens, 256, lds, as_stream(stream)>>>(
            (half_t *)q, (half_t *)k, (half_t *)v, (const half_t *)qkv, (const half_t *)bias, padding_offset,
            history_len, batch, seq_len, head_num, kv_head_num, head_size, rotary_dim, rotary_base);
    else
        LLMIE_UNSUPPORTED("qkv_bias_transpose_rope: dtype %d", (int)dtype);
    return launch_status("qkv_bias_transpose_rope");
}

// Prefill (context) attention for the decoder engine, fp16, head_size 128:
//   prefill_rope_append_kernel   fused form of launchFusedQKVAddBiasAndTransposeAndRope + launchConcatKVCache
//                                (qkv_bias_and_rope.cu:5-138, concat_past_kv.cu:10-89): rotates q in place in the packed
//                                QKV buffer, rotates k and writes k,v straight into the cache slot history+pos -- no padded
//                                q/k/v buffers, no padding offsets.
//   prefill_flash_kernel         fused form of launchRepeatKVCache + batched QK^T + launchFusedScaleMaskAndSoftmax +
//                                batched PV + launchFusedTransposeAndRemovePadding (context_attention.cpp:205-292):
//                                flash attention with online softmax; the causal mask is computed from the lengths
//                                (k <= q_pos + history), the [bs,nh,q,k] score matrix is never materialised, GQA heads read
//                                their kv head in place.  MFMA v_mfma_f32_16x16x32_f16 for both products:
//                                   S^T = K . Q^T   (A = K tile from LDS, B = Q rows held in registers)
//                                   O^T += V^T . P^T (A = V^T gathered from LDS, B = P straight from the S accumulators:
//                                                     the S^T layout is exactly the B-operand layout, no data movement)
//                                so every lane owns one query row's statistics (col = lane & 15) throughout.
#include "llmie_internal.h"

namespace llmie {


// token -> (batch, position in its sequence) from the exclusive prefix cum[batch+1]
__device__ __forceinline__ void locate_token(const int32_t *__restrict__ cum, int batch, int t, int &b, int &pos) {
    int lo = 0;
    for (int i = 0; i < batch; ++i)
        if (t >= cum[i]) lo = i;
    b = lo;
    pos = t - cum[lo];
}

// KV8: the caches are e4m3 bytes, stored = e4m3(x / scale) (same [L, bs, kvh, max_seq, hs] indexing in elements)
__device__ __forceinline__ uint8_t e4m3_of(float x) {
    x = fminf(fmaxf(x, -448.f), 448.f);
    return static_cast<uint8_t>(__builtin_amdgcn_cvt_pk_fp8_f32(x, 0.f, 0, false) & 0xFF);
}

template <int HS, bool KV8>
__global__ __launch_bounds__(256) void prefill_rope_append_kernel(half_t *__restrict__ qkv, const half_t *__restrict__ bias,
                                                                  void *__restrict__ k_cache, void *__restrict__ v_cache,
                                                                  const int32_t *__restrict__ cum, const int32_t *__restrict__ hist,
                                                                  const float2 *__restrict__ rope, int batch, int head_num,
                                                                  int kv_head_num, int max_seq_len, int rotary_dim, size_t layer_off,
                                                                  float k_inv_scale, float v_inv_scale) {
    const int t = blockIdx.x;
    int b, pos;
    locate_token(cum, batch, t, b, pos);
    const int tpos = hist[b] + pos;
    if (tpos < 0 || tpos >= max_seq_len) return;  // never write outside the slab
    constexpr int HALF = HS / 2;
    const int heads = head_num + 2 * kv_head_num;
    half_t *row = qkv + static_cast<size_t>(t) * heads * HS;
    const float2 *cs = rope + static_cast<size_t>(tpos) * HALF;
    for (int i = threadIdx.x; i < heads * HALF; i += 256) {
        const int h = i / HALF, d = i - h * HALF;
        half_t *src = row + static_cast<size_t>(h) * HS;
        float x0 = to_f32(src[d]), x1 = to_f32(src[d + HALF]);
        if (bias) {
            x0 += to_f32(bias[static_cast<size_t>(h) * HS + d]);
            x1 += to_f32(bias[static_cast<size_t>(h) * HS + d + HALF]);
        }
        float o0 = x0, o1 = x1;
        if (h < head_num + kv_head_num && d < (rotary_dim >> 1)) {
            const float2 v = cs[d];
            o0 = x0 * v.x - x1 * v.y;
            o1 = x1 * v.x + x0 * v.y;
        }
        if (h < head_num) {
            src[d] = from_f32<half_t>(o0);
            src[d + HALF] = from_f32<half_t>(o1);
        } else {
            const bool is_k = h < head_num + kv_head_num;
            const int g = is_k ? h - head_num : h - head_num - kv_head_num;
            const size_t off = layer_off + ((static_cast<size_t>(b) * kv_head_num + g) * max_seq_len + tpos) * HS;
            if constexpr (KV8) {
                // quantise the fp16-rounded value (what the fp16 cache would hold), as the decode kernel does
                uint8_t *dst = static_cast<uint8_t *>(is_k ? k_cache : v_cache) + off;
                const float inv = is_k ? k_inv_scale : v_inv_scale;
                dst[d] = e4m3_of(to_f32(from_f32<half_t>(o0)) * inv);
                dst[d + HALF] = e4m3_of(to_f32(from_f32<half_t>(o1)) * inv);
            } else {
                half_t *dst = static_cast<half_t *>(is_k ? k_cache : v_cache) + off;
                dst[d] = from_f32<half_t>(o0);
                dst[d + HALF] = from_f32<half_t>(o1);
            }
        }
    }
}

// grid: (q tiles of 64 rows over max_q_len, head_num, batch); block 256 = 4 waves x 16 query rows
template <int HS, bool KV8>
__global__ __launch_bounds__(256) void prefill_flash_kernel(const half_t *__restrict__ qkv, const void *__restrict__ k_cache,
                                                            const void *__restrict__ v_cache, half_t *__restrict__ out,
                                                            const int32_t *__restrict__ cum, const int32_t *__restrict__ hist,
                                                            int head_num, int kv_head_num, int max_seq_len, size_t layer_off,
                                                            float k_scale, float v_scale) {
    static_assert(HS == 128, "tuned for head_size 128");
    constexpr int BQ = 64, BT = 64, VSTRIDE = HS + 4;  // V rows padded by 8 bytes (bank spread for the column gathers)
    __shared__ __attribute__((aligned(16))) half_t Ks[BT * HS];
    __shared__ __attribute__((aligned(16))) half_t Vs[BT * VSTRIDE];
    const int b = blockIdx.z, h = blockIdx.y;
    const int len = cum[b + 1] - cum[b], history = hist[b];
    const int q0 = blockIdx.x * BQ;
    if (q0 >= len) return;
    const int ctx = history + len;
    const int g = h / (head_num / kv_head_num);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int heads = head_num + 2 * kv_head_num;
    const float scale = rsqrtf(static_cast<float>(HS));

    // this lane's query row (clamped for the tail tile; its results are not stored)
    const int qrow = q0 + wave * 16 + r;
    const int qrow_c = min(qrow, len - 1);
    const half_t *qptr = qkv + (static_cast<size_t>(cum[b] + qrow_c) * heads + h) * HS;
    half8_t qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const half8_t *>(qptr + 32 * s + 8 * q);
    const int qpos = history + qrow_c;  // keys t <= qpos are visible

    const size_t head_off = layer_off + (static_cast<size_t>(b) * kv_head_num + g) * max_seq_len * HS;
    const half_t *kc = static_cast<const half_t *>(k_cache) + head_off, *vc = static_cast<const half_t *>(v_cache) + head_off;
    const uint8_t *kc8 = static_cast<const uint8_t *>(k_cache) + head_off, *vc8 = static_cast<const uint8_t *>(v_cache) + head_off;
    (void)kc; (void)vc; (void)kc8; (void)vc8;

    floatx4 o[8];  // O^T tiles: rows d = 16*dt + 4q + e, col = this lane's query row
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) o[dt] = floatx4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    const int t_hi = min(ctx, history + min(q0 + BQ, len));  // keys needed by any row of this q tile
    for (int t0 = 0; t0 < t_hi; t0 += BT) {
        __syncthreads();  // previous tile consumed
        // stage K (swizzled 16-byte chunks) and V (row-major, padded) tiles: 64 rows x 16 chunks each, 4 per thread
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + 256 * i, row = id >> 4, ch = id & 15;
            const int t = min(t0 + row, ctx - 1);
            half8_t kvv, vvv;
            if constexpr (KV8) {
                // 8 e4m3 bytes -> 8 halves (x scale): the tiles in LDS are fp16 either way
                const uint2 kb = *reinterpret_cast<const uint2 *>(kc8 + static_cast<size_t>(t) * HS + ch * 8);
                const uint2 vb = *reinterpret_cast<const uint2 *>(vc8 + static_cast<size_t>(t) * HS + ch * 8);
                const half2_t k0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(kb.x), k_scale, false);
                const half2_t k1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(kb.x), k_scale, true);
                const half2_t k2 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(kb.y), k_scale, false);
                const half2_t k3 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(kb.y), k_scale, true);
                const half2_t v0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(vb.x), v_scale, false);
                const half2_t v1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(vb.x), v_scale, true);
                const half2_t v2 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(vb.y), v_scale, false);
                const half2_t v3 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(vb.y), v_scale, true);
                kvv = half8_t{k0[0], k0[1], k1[0], k1[1], k2[0], k2[1], k3[0], k3[1]};
                vvv = half8_t{v0[0], v0[1], v1[0], v1[1], v2[0], v2[1], v3[0], v3[1]};
            } else {
                kvv = *reinterpret_cast<const half8_t *>(kc + static_cast<size_t>(t) * HS + ch * 8);
                vvv = *reinterpret_cast<const half8_t *>(vc + static_cast<size_t>(t) * HS + ch * 8);
            }
            *reinterpret_cast<half8_t *>(Ks + row * HS + ((ch ^ (row & 15)) << 3)) = kvv;
            *reinterpret_cast<half4_t *>(Vs + row * VSTRIDE + ch * 8) = half4_t{vvv[0], vvv[1], vvv[2], vvv[3]};
            *reinterpret_cast<half4_t *>(Vs + row * VSTRIDE + ch * 8 + 4) = half4_t{vvv[4], vvv[5], vvv[6], vvv[7]};
        }
        __syncthreads();
        // ---- S^T = K . Q^T : 4 key tiles of 16, 4 k-steps over the head dim ----
        floatx4 sacc[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            sacc[tt] = floatx4{0.f, 0.f, 0.f, 0.f};
            const int row = tt * 16 + r;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const half8_t kf = *reinterpret_cast<const half8_t *>(Ks + row * HS + (((s * 4 + q) ^ (row & 15)) << 3));
                sacc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[s], sacc[tt], 0, 0, 0);
            }
        }
        // lane holds S[qrow][t = t0 + 16 tt + 4 q + e]; mask + scale
        float mloc = -INFINITY;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = t0 + tt * 16 + 4 * q + e;
                const float v = (t <= qpos && t < ctx) ? sacc[tt][e] * scale : -INFINITY;
                sacc[tt][e] = v;
                mloc = fmaxf(mloc, v);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;  // fully masked so far: keep everything at zero
        const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_use);
        float lsum = 0.f;
        half8_t pf[2];  // P^T fragments = MFMA B operand of the two 32-key steps
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p = (sacc[tt][e] == -INFINITY) ? 0.f : __expf(sacc[tt][e] - m_use);
                lsum += p;
                pf[tt >> 1][(tt & 1) * 4 + e] = from_f32<half_t>(p);
            }
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) o[dt] *= alpha;
        // ---- O^T += V^T . P^T : A fragment = V[t(q,j)][d = 16 dt + r] gathered down a column ----
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                half8_t vf;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int trow = 32 * c + (j >> 2) * 16 + 4 * q + (j & 3);
                    vf[j] = Vs[trow * VSTRIDE + dt * 16 + r];
                }
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[c], o[dt], 0, 0, 0);
            }
        }
    }
    if (qrow < len) {
        const float inv = 1.0f / (l_run + 1e-6f);  // the reference's denominator epsilon (scale_and_mask_and_softmax.cu:118)
        half_t *optr = out + (static_cast<size_t>(cum[b] + qrow) * head_num + h) * HS;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            const half4_t v = {from_f32<half_t>(o[dt][0] * inv), from_f32<half_t>(o[dt][1] * inv),
                               from_f32<half_t>(o[dt][2] * inv), from_f32<half_t>(o[dt][3] * inv)};
            *reinterpret_cast<half4_t *>(optr + dt * 16 + 4 * q) = v;
        }
    }
}

int prefill_attention_f16(half_t *qkv, const half_t *qkv_bias, void *k_cache, void *v_cache, half_t *out,
                          const int32_t *cum_seqlens, const int32_t *history_len, const float2 *rope, int layer, int batch,
                          int num_tokens, int max_q_len, int head_num, int kv_head_num, int head_size, int max_seq_len,
                          int rotary_dim, hipStream_t st, int kv_fp8, float k_scale, float v_scale) {
    if (head_size != 128) {
        set_error("prefill attention: head_size %d not supported by the flash kernel (128 only)", head_size);
        return LLMIE_ERR_UNSUPPORTED;
    }
    const size_t layer_off = static_cast<size_t>(layer) * batch * kv_head_num * max_seq_len * head_size;
    dim3 grid((max_q_len + 63) / 64, head_num, batch);
    if (kv_fp8) {
        prefill_rope_append_kernel<128, true><<<num_tokens, 256, 0, st>>>(qkv, qkv_bias, k_cache, v_cache, cum_seqlens, history_len,
                                                                          rope, batch, head_num, kv_head_num, max_seq_len,
                                                                          rotary_dim, layer_off, 1.0f / k_scale, 1.0f / v_scale);
        prefill_flash_kernel<128, true><<<grid, 256, 0, st>>>(qkv, k_cache, v_cache, out, cum_seqlens, history_len, head_num,
                                                              kv_head_num, max_seq_len, layer_off, k_scale, v_scale);
    } else {
        prefill_rope_append_kernel<128, false><<<num_tokens, 256, 0, st>>>(qkv, qkv_bias, k_cache, v_cache, cum_seqlens, history_len,
                                                                           rope, batch, head_num, kv_head_num, max_seq_len,
                                                                           rotary_dim, layer_off, 1.f, 1.f);
        prefill_flash_kernel<128, false><<<grid, 256, 0, st>>>(qkv, k_cache, v_cache, out, cum_seqlens, history_len, head_num,
                                                               kv_head_num, max_seq_len, layer_off, 1.f, 1.f);
    }
    return launch_status("prefill_attention");
}

}  // namespace llmie

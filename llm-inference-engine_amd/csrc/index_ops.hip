// Index / copy kernels (bit-exact integer & byte work, HBM-bound):
//   llmie_input_embedding, llmie_cal_padding_offset, llmie_build_causal_mask,
//   llmie_transpose_remove_padding, llmie_concat_kv, llmie_repeat_kv.
// All copies move 16 bytes per lane where the head/hidden size allows, offsets in 64-bit.
#include "device_utils.cuh"

namespace llmie {

// -------- embedding gather: out[t,:] = table[ids[t],:]  (input_embedding.cu:4-22) --------
template <typename T>
__global__ __launch_bounds__(256) void embedding_kernel(const int32_t *__restrict__ ids,
                                                        const T *__restrict__ table,
                                                        T *__restrict__ out, int hidden, int vocab,
                                                        bool vec_ok) {
    const int t = blockIdx.x;
    const int id = ids[t];
    if (id < 0 || id >= vocab) return;
    const T *src = table + static_cast<size_t>(id) * hidden;
    T *dst = out + static_cast<size_t>(t) * hidden;
    if (vec_ok) {
        using V = typename Vec16<T>::type;
        const int nvec = hidden / Vec16<T>::n;
        const V *s = reinterpret_cast<const V *>(src);
        V *d = reinterpret_cast<V *>(dst);
        for (int i = threadIdx.x; i < nvec; i += 256) d[i] = s[i];
    } else {
        for (int i = threadIdx.x; i < hidden; i += 256) dst[i] = src[i];
    }
}

// -------- padding offset (cal_padding_offset.cu:17-43): one workgroup, parallel over the
// batch instead of the reference's <<<1,1>>> serial loop --------
__global__ __launch_bounds__(256) void padding_offset_kernel(int32_t *__restrict__ padding_offset,
                                                             int32_t *__restrict__ cum_seqlens,
                                                             const int32_t *__restrict__ lens,
                                                             int batch, int max_q_len) {
    extern __shared__ int32_t cum[];  // [batch+1] exclusive prefix of lens
    // serial prefix by one lane per 256-chunk is enough: batch is at most a few thousand
    if (threadIdx.x == 0) {
        int total = 0;
        for (int b = 0; b < batch; ++b) {
            cum[b] = total;
            total += lens[b];
        }
        cum[batch] = total;
    }
    __syncthreads();
    for (int b = threadIdx.x; b <= batch; b += 256) cum_seqlens[b] = cum[b];
    for (int b = 0; b < batch; ++b) {
        const int start = cum[b], len = cum[b + 1] - cum[b];
        const int pad = b * max_q_len - start;  // = sum_{j<b}(max_q_len - lens[j])
        // never write past [batch, max_q_len] even if a length is corrupt (> max_q_len)
        for (int j = threadIdx.x; j < len && start + j < batch * max_q_len; j += 256)
            padding_offset[start + j] = pad;
    }
}

// -------- causal mask (build_causal_mask.cu:4-23) --------
template <typename T>
__global__ __launch_bounds__(256) void causal_mask_kernel(T *__restrict__ mask,
                                                          const int32_t *__restrict__ q_lens,
                                                          const int32_t *__restrict__ k_lens,
                                                          int max_q_len, int max_k_len) {
    const int b = blockIdx.y;
    const int ql = q_lens[b], kl = k_lens[b];
    const size_t per = static_cast<size_t>(max_q_len) * max_k_len;
    T *m = mask + static_cast<size_t>(b) * per;
    for (size_t o = blockIdx.x * 256ull + threadIdx.x; o < per; o += static_cast<size_t>(gridDim.x) * 256) {
        const int q = static_cast<int>(o / max_k_len);
        const int k = static_cast<int>(o - static_cast<size_t>(q) * max_k_len);
        const bool one = (q < ql) && (k < kl) && (k <= q + (kl - ql));
        m[o] = from_f32<T>(one ? 1.f : 0.f);
    }
}

// -------- [bs,nh,S,hs] -> [T,nh,hs] without pads (transpose_and_remove_padding.cu:15-43) ----
template <typename T>
__global__ __launch_bounds__(256) void transpose_remove_padding_kernel(
    const T *__restrict__ src, T *__restrict__ dst, const int32_t *__restrict__ padding_offset,
    int seq_len, int head_num, int head_size, bool vec_ok) {
    const int t = blockIdx.x;
    const int pt = t + padding_offset[t];
    const int b = pt / seq_len, s = pt % seq_len;
    const T *sbase = src + (static_cast<size_t>(b) * head_num * seq_len + s) * head_size;
    T *dbase = dst + static_cast<size_t>(t) * head_num * head_size;
    if (vec_ok) {
        using V = typename Vec16<T>::type;
        const int vph = head_size / Vec16<T>::n;
        for (int i = threadIdx.x; i < head_num * vph; i += 256) {
            const int h = i / vph, v = i - h * vph;
            reinterpret_cast<V *>(dbase + static_cast<size_t>(h) * head_size)[v] =
                reinterpret_cast<const V *>(sbase + static_cast<size_t>(h) * seq_len * head_size)[v];
        }
    } else {
        for (int i = threadIdx.x; i < head_num * head_size; i += 256) {
            const int h = i / head_size, d = i - h * head_size;
            dbase[i] = sbase[static_cast<size_t>(h) * seq_len * head_size + d];
        }
    }
}

// -------- KV append (concat_past_kv.cu:10-42) and GQA broadcast (repeat_kv.cu:13-49) --------
// One thread per 16-byte (or scalar) element of the [b, h, t, :] space.
template <typename T, bool REPEAT>
__global__ __launch_bounds__(256) void kv_copy_kernel(
    const T *__restrict__ src, T *__restrict__ dst, const int32_t *__restrict__ len_a,
    const int32_t *__restrict__ history, size_t layer_off, int heads, int rep, int t_dim,
    int max_seq_len, int head_size, int epr /* elements (vec or scalar) per row */, bool vec_ok,
    size_t total) {
    using V = typename Vec16<T>::type;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * 256) {
        const int e = static_cast<int>(i % epr);
        size_t r = i / epr;
        const int t = static_cast<int>(r % t_dim);
        r /= t_dim;
        const int h = static_cast<int>(r % heads);
        const int b = static_cast<int>(r / heads);
        if (t >= len_a[b]) continue;
        size_t so, dofs;
        if constexpr (REPEAT) {
            // src = cache[layer,b,h/rep,t,:]  dst = [b,h,t,:] with t_dim = max_k_len
            const int kvh = heads / rep;
            so = layer_off + ((static_cast<size_t>(b) * kvh + h / rep) * max_seq_len + t) * head_size;
            dofs = ((static_cast<size_t>(b) * heads + h) * t_dim + t) * head_size;
        } else {
            // src = [b,h,t,:] with t_dim = max_q_len  dst = cache[layer,b,h,history[b]+t,:]
            if (history[b] < 0 || history[b] + t >= max_seq_len) continue;  // never write past the slab
            so = ((static_cast<size_t>(b) * heads + h) * t_dim + t) * head_size;
            dofs = layer_off + ((static_cast<size_t>(b) * heads + h) * max_seq_len + history[b] + t) * head_size;
        }
        if (vec_ok)
            reinterpret_cast<V *>(dst + dofs)[e] = reinterpret_cast<const V *>(src + so)[e];
        else
            dst[dofs + e] = src[so + e];
    }
}

static inline int grid_for(size_t work_items) {
    size_t g = (work_items + 255) / 256;
    return static_cast<int>(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

template <typename T> static bool vec_aligned(int inner, const void *a, const void *b) {
    return inner % Vec16<T>::n == 0 &&
           ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) % 16 == 0);
}

}  // namespace llmie

using namespace llmie;

extern "C" int llmie_input_embedding(const int32_t *ids, const void *table, void *out, int num_tokens,
                                     int hidden, int vocab, llmie_dtype dtype, llmie_stream stream) {
    LLMIE_REQUIRE(ids && table && out, "input_embedding: NULL pointer");
    LLMIE_REQUIRE(num_tokens > 0 && hidden > 0 && vocab > 0, "input_embedding: bad shape");
    hipStream_t st = as_stream(stream);
    if (dtype == LLMIE_F32)
        embedding_kernel<float><<<num_tokens, 256, 0, st>>>(ids, (const float *)table, (float *)out, hidden,
                                                            vocab, vec_aligned<float>(hidden, table, out));
    else if (dtype == LLMIE_F16)
        embedding_kernel<half_t><<<num_tokens, 256, 0, st>>>(ids, (const half_t *)table, (half_t *)out, hidden,
                                                             vocab, vec_aligned<half_t>(hidden, table, out));
    else
        LLMIE_UNSUPPORTED("input_embedding: dtype %d", (int)dtype);
    return launch_status("input_embedding");
}

extern "C" int llmie_cal_padding_offset(int32_t *padding_offset, int32_t *cum_seqlens,
                                        const int32_t *input_lengths, int batch, int max_q_len,
                                        llmie_stream stream) {
    LLMIE_REQUIRE(padding_offset && cum_seqlens && input_lengths, "cal_padding_offset: NULL pointer");
    LLMIE_REQUIRE(batch > 0 && max_q_len > 0, "cal_padding_offset: bad shape");
    LLMIE_REQUIRE(batch <= 16000, "cal_padding_offset: batch %d exceeds the LDS prefix buffer", batch);
    padding_offset_kernel<<<1, 256, sizeof(int32_t) * (batch + 1), as_stream(stream)>>>(
        padding_offset, cum_seqlens, input_lengths, batch, max_q_len);
    return launch_status("cal_padding_offset");
}

extern "C" int llmie_build_causal_mask(void *mask, const int32_t *q_lens, const int32_t *k_lens,
                                       int batch, int max_q_len, int max_k_len, llmie_dtype dtype,
                                       llmie_stream stream) {
    LLMIE_REQUIRE(mask && q_lens && k_lens, "build_causal_mask: NULL pointer");
    LLMIE_REQUIRE(batch > 0 && max_q_len > 0 && max_k_len > 0, "build_causal_mask: bad shape");
    const size_t per = static_cast<size_t>(max_q_len) * max_k_len;
    int gx = grid_for(per);
    if (gx > 1024) gx = 1024;
    dim3 grid(gx, batch);
    if (dtype == LLMIE_F32)
        causal_mask_kernel<float><<<grid, 256, 0, as_stream(stream)>>>((float *)mask, q_lens, k_lens, max_q_len, max_k_len);
    else if (dtype == LLMIE_F16)
        causal_mask_kernel<half_t><<<grid, 256, 0, as_stream(stream)>>>((half_t *)mask, q_lens, k_lens, max_q_len, max_k_len);
    else
        LLMIE_UNSUPPORTED("build_causal_mask: dtype %d", (int)dtype);
    return launch_status("build_causal_mask");
}

extern "C" int llmie_transpose_remove_padding(const void *src, void *dst, const int32_t *padding_offset,
                                              int num_tokens, int batch, int seq_len, int head_num,
                                              int head_size, llmie_dtype dtype, llmie_stream stream) {
    LLMIE_REQUIRE(src && dst && padding_offset, "transpose_remove_padding: NULL pointer");
    LLMIE_REQUIRE(num_tokens > 0 && batch > 0 && seq_len > 0 && head_num > 0 && head_size > 0,
                  "transpose_remove_padding: bad shape");
    LLMIE_REQUIRE(num_tokens <= batch * seq_len, "transpose_remove_padding: num_tokens > batch*seq_len");
    hipStream_t st = as_stream(stream);
    if (dtype == LLMIE_F32)
        transpose_remove_padding_kernel<float><<<num_tokens, 256, 0, st>>>(
            (const float *)src, (float *)dst, padding_offset, seq_len, head_num, head_size,
            vec_aligned<float>(head_size, src, dst));
    else if (dtype == LLMIE_F16)
        transpose_remove_padding_kernel<half_t><<<num_tokens, 256, 0, st>>>(
            (const half_t *)src, (half_t *)dst, padding_offset, seq_len, head_num, head_size,
            vec_aligned<half_t>(head_size, src, dst));
    else
        LLMIE_UNSUPPORTED("transpose_remove_padding: dtype %d", (int)dtype);
    return launch_status("transpose_remove_padding");
}

template <typename T, bool REPEAT>
static int launch_kv_copy(const void *src, void *dst, const int32_t *len_a, const int32_t *history,
                          int layer, int batch, int heads, int kv_heads, int t_dim, int max_seq_len,
                          int head_size, hipStream_t st) {
    const bool v = vec_aligned<T>(head_size, src, dst);
    const int epr = v ? head_size / Vec16<T>::n : head_size;
    const size_t total = static_cast<size_t>(batch) * heads * t_dim * epr;
    const size_t layer_off = static_cast<size_t>(layer) * batch * kv_heads * max_seq_len * head_size;
    kv_copy_kernel<T, REPEAT><<<grid_for(total), 256, 0, st>>>(
        (const T *)src, (T *)dst, len_a, history, layer_off, heads, heads / kv_heads, t_dim, max_seq_len,
        head_size, epr, v, total);
    return launch_status(REPEAT ? "repeat_kv" : "concat_kv");
}

extern "C" int llmie_concat_kv(const void *src, void *cache, const int32_t *cur_len,
                               const int32_t *history_len, int layer, int batch, int kv_head_num,
                               int max_q_len, int max_seq_len, int head_size, llmie_dtype dtype,
                               llmie_stream stream) {
    LLMIE_REQUIRE(src && cache && cur_len && history_len, "concat_kv: NULL pointer");
    LLMIE_REQUIRE(layer >= 0 && batch > 0 && kv_head_num > 0 && max_q_len > 0 && max_seq_len > 0 && head_size > 0,
                  "concat_kv: bad shape");
    if (dtype == LLMIE_F32)
        return launch_kv_copy<float, false>(src, cache, cur_len, history_len, layer, batch, kv_head_num,
                                            kv_head_num, max_q_len, max_seq_len, head_size, as_stream(stream));
    if (dtype == LLMIE_F16)
        return launch_kv_copy<half_t, false>(src, cache, cur_len, history_len, layer, batch, kv_head_num,
                                             kv_head_num, max_q_len, max_seq_len, head_size, as_stream(stream));
    LLMIE_UNSUPPORTED("concat_kv: dtype %d", (int)dtype);
}

extern "C" int llmie_repeat_kv(const void *cache, void *dst, const int32_t *ctx_len, int layer, int batch,
                               int head_num, int kv_head_num, int max_k_len, int max_seq_len,
                               int head_size, llmie_dtype dtype, llmie_stream stream) {
    LLMIE_REQUIRE(cache && dst && ctx_len, "repeat_kv: NULL pointer");
    LLMIE_REQUIRE(layer >= 0 && batch > 0 && head_num > 0 && kv_head_num > 0 && max_k_len > 0 &&
                      max_seq_len > 0 && head_size > 0, "repeat_kv: bad shape");
    LLMIE_REQUIRE(head_num % kv_head_num == 0, "repeat_kv: kv_head_num must divide head_num");
    LLMIE_REQUIRE(max_k_len <= max_seq_len, "repeat_kv: max_k_len > max_seq_len");
    if (dtype == LLMIE_F32)
        return launch_kv_copy<float, true>(cache, dst, ctx_len, nullptr, layer, batch, head_num, kv_head_num,
                                           max_k_len, max_seq_len, head_size, as_stream(stream));
    if (dtype == LLMIE_F16)
        return launch_kv_copy<half_t, true>(cache, dst, ctx_len, nullptr, layer, batch, head_num, kv_head_num,
                                            max_k_len, max_seq_len, head_size, as_stream(stream));
    LLMIE_UNSUPPORTED("repeat_kv: dtype %d", (int)dtype);
}

// dense [L, batch, kvh, max_seq, hs] <-> page pools [L, num_pages, kvh, 128, hs] through the block table; one workgroup per
// (token, kv head, batch), 16-byte accesses when the row allows, plain bytes otherwise; bit-exact copy
namespace llmie {
__global__ __launch_bounds__(64) void kv_pages_copy_kernel(unsigned char *dense, unsigned char *pool, const int32_t *__restrict__ table,
                                                           const int32_t *__restrict__ ctx_len, int to_pages, int batch, int kvh,
                                                           int max_seq, int row_bytes, int max_pages, int num_pages) {
    const int t = blockIdx.x, g = blockIdx.y, lb = blockIdx.z;  // lb = layer * batch + b
    const int b = lb % batch, layer = lb / batch;
    if (t >= ctx_len[b]) return;
    const int page = table[static_cast<size_t>(b) * max_pages + t / 128];
    unsigned char *d = dense + ((static_cast<size_t>(lb) * kvh + g) * max_seq + t) * row_bytes;
    unsigned char *p = pool + (((static_cast<size_t>(layer) * num_pages + page) * kvh + g) * 128 + t % 128) * row_bytes;
    unsigned char *dst = to_pages ? p : d;
    const unsigned char *src = to_pages ? d : p;
    if (row_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) % 16 == 0) {
        for (int i = threadIdx.x; i < row_bytes / 16; i += 64) reinterpret_cast<uint4_t *>(dst)[i] = reinterpret_cast<const uint4_t *>(src)[i];
    } else {
        for (int i = threadIdx.x; i < row_bytes; i += 64) dst[i] = src[i];
    }
}
}  // namespace llmie

extern "C" int llmie_kv_pages_copy(void *dense, void *pool, const int32_t *block_table, const int32_t *ctx_len, int to_pages,
                                   int layers, int batch, int kv_head_num, int max_seq_len, int head_size, int max_pages,
                                   int num_pages, int elem_bytes, llmie_stream stream) {
    LLMIE_REQUIRE(dense && pool && block_table && ctx_len, "kv_pages_copy: NULL pointer");
    LLMIE_REQUIRE(layers > 0 && batch > 0 && kv_head_num > 0 && max_seq_len > 0 && head_size > 0 && max_pages > 0 && num_pages > 0 &&
                      elem_bytes > 0, "kv_pages_copy: bad shape");
    LLMIE_REQUIRE(static_cast<long long>(max_pages) * 128 >= max_seq_len, "kv_pages_copy: max_pages * 128 < max_seq_len");
    LLMIE_REQUIRE(kv_head_num <= 65535 && static_cast<long long>(layers) * batch <= 65535, "kv_pages_copy: grid too large");
    dim3 grid(max_seq_len, kv_head_num, layers * batch);
    llmie::kv_pages_copy_kernel<<<grid, 64, 0, llmie::as_stream(stream)>>>(static_cast<unsigned char *>(dense), static_cast<unsigned char *>(pool),
                                                                        block_table, ctx_len, to_pages, batch, kv_head_num, max_seq_len,
                                                                        head_size * elem_bytes, max_pages, num_pages);
    return llmie::launch_status("kv_pages_copy");
}


// llmie_topk (topk.cu:24-140) and llmie_sampling (sampling.cu:14-102) for gfx950.
//
// top-k: per row the K largest logits, descending, ties -> lower id (a strict total order, so the
// result is bit-exact against the oracle).  Round 1: rows*blocks_per_row workgroups each scan a
// contiguous slice with 16-byte loads, every thread keeps a sorted K-list in registers, then K
// rounds of workgroup arg-max (wave64 shuffles + LDS) extract the slice's list.  Round 2 merges
// the blocks_per_row*K candidates of a row the same way.  The reference's defects (uninitialised
// round-2 heap, missing row offset, -1e-20 sentinel, hard-coded K=5: SURVEY 9-K8) are not kept.
//
// sampling: one lane per sequence; Philox4x32-10 keyed by (step, "LLMI"), counter = batch index.
#include "device_utils.cuh"

#include <climits>

namespace llmie {

__device__ __forceinline__ bool better(float av, int ai, float bv, int bi) {
    return av > bv || (av == bv && ai < bi);
}

template <int KMAX> struct TopList {
    float v[KMAX];
    int id[KMAX];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
            v[i] = -INFINITY;
            id[i] = INT_MAX;
        }
    }
    // K <= KMAX live entries; insert keeps (value desc, id asc)
    __device__ __forceinline__ void insert(float x, int xi, int K) {
        if (!better(x, xi, v[KMAX - 1], id[KMAX - 1])) return;
        (void)K;
        v[KMAX - 1] = x;
        id[KMAX - 1] = xi;
#pragma unroll
        for (int j = KMAX - 1; j > 0; --j) {
            if (better(v[j], id[j], v[j - 1], id[j - 1])) {
                const float tv = v[j]; v[j] = v[j - 1]; v[j - 1] = tv;
                const int ti = id[j]; id[j] = id[j - 1]; id[j - 1] = ti;
            }
        }
    }
    __device__ __forceinline__ void pop() {
#pragma unroll
        for (int j = 0; j < KMAX - 1; ++j) {
            v[j] = v[j + 1];
            id[j] = id[j + 1];
        }
        v[KMAX - 1] = -INFINITY;
        id[KMAX - 1] = INT_MAX;
    }
};

// K rounds of block arg-max over the heads of the per-thread lists; thread 0 writes the result.
template <typename T, int KMAX, int NW>
__device__ __forceinline__ void block_select(TopList<KMAX> &tl, int K, int32_t *out_ids, T *out_vals,
                                             float *s_v, int *s_i) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < K; ++k) {
        float bv = tl.v[0];
        int bi = tl.id[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = lane_xor_o(bv, o);   // (DPP / row swaps: no LDS round trip; device_utils.cuh)
            const int oi = lane_xor_o(bi, o);
            if (better(ov, oi, bv, bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if (NW > 1) {
            __syncthreads();
            if (lane == 0) {
                s_v[wave] = bv;
                s_i[wave] = bi;
            }
            __syncthreads();
            bv = s_v[0];
            bi = s_i[0];
#pragma unroll
            for (int w = 1; w < NW; ++w)
                if (better(s_v[w], s_i[w], bv, bi)) {
                    bv = s_v[w];
                    bi = s_i[w];
                }
        }
        if (bi != INT_MAX && tl.id[0] == bi) tl.pop();  // ids are unique: exactly one owner
        if (threadIdx.x == 0) {
            out_ids[k] = (bi == INT_MAX) ? -1 : bi;
            out_vals[k] = from_f32<T>(bv);
        }
    }
}

template <typename T, int KMAX>
__global__ __launch_bounds__(256) void topk_round1_kernel(const T *__restrict__ probs, int32_t *__restrict__ ids,
                                                          T *__restrict__ vals, int vocab, int K, int bpr,
                                                          bool vec_ok) {
    __shared__ float s_v[4];
    __shared__ int s_i[4];
    const int row = blockIdx.x / bpr, blk = blockIdx.x % bpr;
    const T *p = probs + static_cast<size_t>(row) * vocab;
    TopList<KMAX> tl;
    tl.init();
    if (vec_ok) {
        using V = typename Vec16<T>::type;
        constexpr int N = Vec16<T>::n;
        const int nvec = vocab / N;
        const int per = (nvec + bpr - 1) / bpr;
        const int v0 = blk * per, v1 = min(nvec, v0 + per);
        for (int i = v0 + threadIdx.x; i < v1; i += 256) {
            const V x = reinterpret_cast<const V *>(p)[i];
#pragma unroll
            for (int e = 0; e < N; ++e) tl.insert(to_f32(x[e]), i * N + e, K);
        }
    } else {
        const int per = (vocab + bpr - 1) / bpr;
        const int e0 = blk * per, e1 = min(vocab, e0 + per);
        for (int i = e0 + threadIdx.x; i < e1; i += 256) tl.insert(to_f32(p[i]), i, K);
    }
    block_select<T, KMAX, 4>(tl, K, ids + static_cast<size_t>(blockIdx.x) * K, vals + static_cast<size_t>(blockIdx.x) * K, s_v, s_i);
}

template <typename T, int KMAX>
__global__ __launch_bounds__(64) void topk_round2_kernel(const int32_t *__restrict__ tmp_ids, const T *__restrict__ tmp_vals,
                                                         int32_t *__restrict__ ids, T *__restrict__ vals, int K, int bpr) {
    const int row = blockIdx.x;
    const int n = bpr * K;
    TopList<KMAX> tl;
    tl.init();
    for (int i = threadIdx.x; i < n; i += 64) {
        const int id = tmp_ids[static_cast<size_t>(row) * n + i];
        if (id >= 0) tl.insert(to_f32(tmp_vals[static_cast<size_t>(row) * n + i]), id, K);
    }
    block_select<T, KMAX, 1>(tl, K, ids + static_cast<size_t>(row) * K, vals + static_cast<size_t>(row) * K, nullptr, nullptr);
}

template <typename T>
static int topk_impl(const T *probs, int32_t *tmp_ids, T *tmp_vals, int32_t *ids, T *vals, int rows, int vocab,
                     int K, int bpr, hipStream_t st) {
    const bool vec_ok = vocab % Vec16<T>::n == 0 && reinterpret_cast<uintptr_t>(probs) % 16 == 0;
    int32_t *r1_ids = (bpr == 1) ? ids : tmp_ids;
    T *r1_vals = (bpr == 1) ? vals : tmp_vals;
    if (K <= 8)
        topk_round1_kernel<T, 8><<<rows * bpr, 256, 0, st>>>(probs, r1_ids, r1_vals, vocab, K, bpr, vec_ok);
    else
        topk_round1_kernel<T, 32><<<rows * bpr, 256, 0, st>>>(probs, r1_ids, r1_vals, vocab, K, bpr, vec_ok);
    if (bpr > 1) {
        if (K <= 8)
            topk_round2_kernel<T, 8><<<rows, 64, 0, st>>>(tmp_ids, tmp_vals, ids, vals, K, bpr);
        else
            topk_round2_kernel<T, 32><<<rows, 64, 0, st>>>(tmp_ids, tmp_vals, ids, vals, K, bpr);
    }
    return launch_status("topk");
}

// ---------------- sampling ----------------
__device__ __forceinline__ float uniform_philox(uint32_t seed, uint32_t stream) {
    uint32_t c0 = stream, c1 = 0, c2 = 0, c3 = 0;
    uint32_t k0 = seed, k1 = 0x4c4c4d49u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return static_cast<float>((c0 >> 8) + 1u) * (1.0f / 16777216.0f);
}

template <typename T>
__global__ __launch_bounds__(64) void sampling_kernel(const int32_t *__restrict__ topk_id, const T *__restrict__ topk_val,
                                                      int32_t *__restrict__ seq_len, uint8_t *__restrict__ finished,
                                                      int32_t *__restrict__ out_id, int batch, int K, int step_arg,
                                                      const int32_t *__restrict__ step_dev, int end_id, int vocab) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    const int step = step_dev ? *step_dev : step_arg;
    const int32_t *id = topk_id + static_cast<size_t>(b) * K;
    const T *val = topk_val + static_cast<size_t>(b) * K;
    // candidates with id < 0 are list slots the top-k never filled (NaN logits are never "better" than anything, sampling.cu has
    // no such case because cub's sort keeps NaNs): they are skipped; a row with no valid candidate at all ends its sequence
    // (end_id, finished) instead of emitting token -1
    int first = -1;
    for (int i = 0; i < K && first < 0; ++i)
        if (id[i] >= 0) first = i;
    int chosen = end_id;
    if (first >= 0) {
        const float v0 = to_f32(val[first]);
        float sum = 0.f;
        for (int i = first; i < K; ++i)
            if (id[i] >= 0) sum += expf(to_f32(val[i]) - v0);
        float thr = uniform_philox(static_cast<uint32_t>(step), static_cast<uint32_t>(b)) * sum;
        chosen = id[first] % vocab;
        for (int i = first; i < K; ++i) {
            if (id[i] < 0) continue;
            thr -= expf(to_f32(val[i]) - v0);
            if (thr < 0.f) {
                chosen = id[i] % vocab;
                break;
            }
        }
    }
    out_id[b] = chosen;
    if (!finished[b]) ++seq_len[b];
    finished[b] = static_cast<uint8_t>(chosen == end_id);
}

__global__ void advance_step_kernel(int32_t *step) { *step += 1; }

// ---------------- fused tail of a decode step (round 3) ----------------
// top-k round 2 + sampling + the next step's input embedding + the step counter in ONE launch (llama.cpp:293-318 then :219 of the
// next token: four launches of ~4.5 us each in the batch-1 step).  One wave per sequence: the round-2 merge of the row's
// blocks_per_row * K candidates exactly as topk_round2_kernel does it (the butterfly leaves the winner of every round in all lanes,
// so the K results stay in registers, rounded to T as the unfused kernel stores them), then lane 0 samples with sampling_kernel's
// arithmetic, the wave copies the chosen token's embedding row into next_hidden[row] (input_embedding's rule: ids outside the table
// are skipped), and the LAST wave to finish (a ticket every row takes after it has read the step) advances *step_dev.
template <typename T, int KMAX>
__global__ __launch_bounds__(64) void decode_tail_kernel(const int32_t *__restrict__ tmp_ids, const T *__restrict__ tmp_vals,
                                                         int32_t *__restrict__ ids, T *__restrict__ vals, int K, int bpr,
                                                         int32_t *__restrict__ seq_len, uint8_t *__restrict__ finished,
                                                         int32_t *__restrict__ out_id, int rows, int step_arg, int32_t *step_dev,
                                                         int end_id, int vocab, const T *__restrict__ embed, T *__restrict__ next_hidden,
                                                         int hidden, int advance, unsigned *ticket) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const int step = step_dev ? *step_dev : step_arg;
    const int n = bpr * K;
    TopList<KMAX> tl;
    tl.init();
    for (int i = lane; i < n; i += 64) {
        const int id = tmp_ids[static_cast<size_t>(row) * n + i];
        if (id >= 0) tl.insert(to_f32(tmp_vals[static_cast<size_t>(row) * n + i]), id, K);
    }
    int sid[KMAX];
    float sval[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        sid[k] = -1;
        sval[k] = 0.f;
        if (k < K) {
            float bv = tl.v[0];
            int bi = tl.id[0];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = lane_xor_o(bv, o);   // (DPP / row swaps: no LDS round trip; device_utils.cuh)
                const int oi = lane_xor_o(bi, o);
                if (better(ov, oi, bv, bi)) {
                    bv = ov;
                    bi = oi;
                }
            }
            if (bi != INT_MAX && tl.id[0] == bi) tl.pop();
            const T stored = from_f32<T>(bv);
            sid[k] = (bi == INT_MAX) ? -1 : bi;
            sval[k] = to_f32(stored);
            if (lane == 0) {
                ids[static_cast<size_t>(row) * K + k] = sid[k];
                vals[static_cast<size_t>(row) * K + k] = stored;
            }
        }
    }
    // sampling_kernel's arithmetic on the K candidates (identical in every lane; lane 0 stores)
    int first = -1;
#pragma unroll
    for (int i = 0; i < KMAX; ++i)
        if (i < K && first < 0 && sid[i] >= 0) first = i;
    int chosen = end_id;
    if (first >= 0) {
        float v0 = 0.f;
#pragma unroll
        for (int i = 0; i < KMAX; ++i)
            if (i == first) v0 = sval[i];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < KMAX; ++i)
            if (i < K && i >= first && sid[i] >= 0) sum += expf(sval[i] - v0);
        float thr = uniform_philox(static_cast<uint32_t>(step), static_cast<uint32_t>(row)) * sum;
        bool done = false;
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
            if (i == first) chosen = sid[i] % vocab;
        }
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
            if (i < K && i >= first && sid[i] >= 0 && !done) {
                thr -= expf(sval[i] - v0);
                if (thr < 0.f) {
                    chosen = sid[i] % vocab;
                    done = true;
                }
            }
        }
    }
    if (lane == 0) {
        out_id[row] = chosen;
        if (!finished[row]) ++seq_len[row];
        finished[row] = static_cast<uint8_t>(chosen == end_id);
    }
    if (next_hidden && chosen >= 0 && chosen < vocab) {
        constexpr int N = 16 / sizeof(T);
        const T *src = embed + static_cast<size_t>(chosen) * hidden;
        T *dst = next_hidden + static_cast<size_t>(row) * hidden;
        if (hidden % N == 0 && (reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) % 16 == 0) {
            for (int i = lane; i < hidden / N; i += 64) reinterpret_cast<uint4_t *>(dst)[i] = reinterpret_cast<const uint4_t *>(src)[i];
        } else {
            for (int i = lane; i < hidden; i += 64) dst[i] = src[i];
        }
    }
    if (advance && step_dev && lane == 0) {
        // every row has read the step before it takes its ticket: the last one moves the counter and re-zeroes the ticket word
        if (atomicAdd(ticket, 1u) == static_cast<unsigned>(rows) - 1u) {
            *step_dev = step + 1;
            atomicExch(ticket, 0u);
        }
    }
}

int topk_round1_only(const void *probs, int32_t *tmp_ids, void *tmp_vals, int rows, int vocab, int K, int bpr, llmie_dtype dtype, hipStream_t st) {
    if (dtype == LLMIE_F16) {
        const half_t *p = static_cast<const half_t *>(probs);
        const bool vec_ok = vocab % Vec16<half_t>::n == 0 && reinterpret_cast<uintptr_t>(p) % 16 == 0;
        if (K <= 8) topk_round1_kernel<half_t, 8><<<rows * bpr, 256, 0, st>>>(p, tmp_ids, static_cast<half_t *>(tmp_vals), vocab, K, bpr, vec_ok);
        else topk_round1_kernel<half_t, 32><<<rows * bpr, 256, 0, st>>>(p, tmp_ids, static_cast<half_t *>(tmp_vals), vocab, K, bpr, vec_ok);
    } else {
        const float *p = static_cast<const float *>(probs);
        const bool vec_ok = vocab % Vec16<float>::n == 0 && reinterpret_cast<uintptr_t>(p) % 16 == 0;
        if (K <= 8) topk_round1_kernel<float, 8><<<rows * bpr, 256, 0, st>>>(p, tmp_ids, static_cast<float *>(tmp_vals), vocab, K, bpr, vec_ok);
        else topk_round1_kernel<float, 32><<<rows * bpr, 256, 0, st>>>(p, tmp_ids, static_cast<float *>(tmp_vals), vocab, K, bpr, vec_ok);
    }
    return launch_status("topk(round 1)");
}

int decode_tail(const int32_t *tmp_ids, const void *tmp_vals, int32_t *ids, void *vals, int K, int bpr, int32_t *seq_len, uint8_t *finished,
                int32_t *out_id, int rows, int step, int32_t *step_dev, int end_id, int vocab, const void *embed, void *next_hidden, int hidden,
                int advance, unsigned *ticket, llmie_dtype dtype, hipStream_t st) {
#define LLMIE_TAIL(T_, KM_)                                                                                                         \
    decode_tail_kernel<T_, KM_><<<rows, 64, 0, st>>>(tmp_ids, static_cast<const T_ *>(tmp_vals), ids, static_cast<T_ *>(vals), K, bpr, seq_len, \
                                                     finished, out_id, rows, step, step_dev, end_id, vocab, static_cast<const T_ *>(embed),   \
                                                     static_cast<T_ *>(next_hidden), hidden, advance, ticket)
    if (dtype == LLMIE_F16) {
        if (K <= 8) LLMIE_TAIL(half_t, 8);
        else LLMIE_TAIL(half_t, 32);
    } else {
        if (K <= 8) LLMIE_TAIL(float, 8);
        else LLMIE_TAIL(float, 32);
    }
#undef LLMIE_TAIL
    return launch_status("decode_tail");
}


}  // namespace llmie

using namespace llmie;

extern "C" int llmie_topk(const void *probs, int32_t *tmp_ids, void *tmp_vals, int32_t *ids, void *vals, int rows,
                          int vocab, int K, int blocks_per_row, llmie_dtype dtype, llmie_stream stream) {
    LLMIE_REQUIRE(probs && ids && vals, "topk: NULL pointer");
    LLMIE_REQUIRE(rows > 0 && vocab > 0, "topk: bad shape");
    LLMIE_REQUIRE(K >= 1 && K <= 32 && K <= vocab, "topk: K=%d outside [1, min(32, vocab)]", K);
    LLMIE_REQUIRE(blocks_per_row >= 1 && blocks_per_row <= 64, "topk: blocks_per_row=%d outside [1,64]", blocks_per_row);
    LLMIE_REQUIRE(blocks_per_row == 1 || (tmp_ids && tmp_vals), "topk: tmp buffers required when blocks_per_row > 1");
    if (dtype == LLMIE_F32)
        return topk_impl<float>((const float *)probs, tmp_ids, (float *)tmp_vals, ids, (float *)vals, rows, vocab, K,
                                blocks_per_row, as_stream(stream));
    if (dtype == LLMIE_F16)
        return topk_impl<half_t>((const half_t *)probs, tmp_ids, (half_t *)tmp_vals, ids, (half_t *)vals, rows, vocab, K,
                                 blocks_per_row, as_stream(stream));
    LLMIE_UNSUPPORTED("topk: dtype %d", (int)dtype);
}

extern "C" int llmie_sampling(const int32_t *topk_id, const void *topk_val, int32_t *seq_len, uint8_t *finished,
                              int32_t *out_id, int batch, int K, int step, const int32_t *step_dev, int end_id,
                              int vocab, llmie_dtype dtype, llmie_stream stream) {
    LLMIE_REQUIRE(topk_id && topk_val && seq_len && finished && out_id, "sampling: NULL pointer");
    LLMIE_REQUIRE(batch > 0 && K > 0 && vocab > 0, "sampling: bad shape");
    const int grid = (batch + 63) / 64;
    if (dtype == LLMIE_F32)
        sampling_kernel<float><<<grid, 64, 0, as_stream(stream)>>>(topk_id, (const float *)topk_val, seq_len, finished,
                                                                  out_id, batch, K, step, step_dev, end_id, vocab);
    else if (dtype == LLMIE_F16)
        sampling_kernel<half_t><<<grid, 64, 0, as_stream(stream)>>>(topk_id, (const half_t *)topk_val, seq_len, finished,
                                                                   out_id, batch, K, step, step_dev, end_id, vocab);
    else
        LLMIE_UNSUPPORTED("sampling: dtype %d", (int)dtype);
    return launch_status("sampling");
}

extern "C" int llmie_advance_step(int32_t *step_dev, llmie_stream stream) {
    LLMIE_REQUIRE(step_dev, "advance_step: NULL pointer");
    advance_step_kernel<<<1, 1, 0, as_stream(stream)>>>(step_dev);
    return launch_status("advance_step");
}

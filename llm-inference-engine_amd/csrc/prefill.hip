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

#include <cstdlib>

namespace llmie {


#ifdef FLASH_STAMPS   // diagnostic build only (tools/micro/flash_probe.hip): where a wave's cycles go inside one key tile; never defined in the product build
__device__ unsigned long long flash_stamp_buf[64 * 8 * 8];   // [workgroup < 64][wave][segment]: shader-clock cycles summed over the tiles
#define FLASH_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define FLASH_ACC(i, a, b) stamp_acc[i] += (b) - (a)
#else
#define FLASH_T(var) do { } while (0)
#define FLASH_ACC(i, a, b) do { } while (0)
#endif

// token -> (batch, position in its sequence) from the exclusive prefix cum[batch+1]
__device__ __forceinline__ void locate_token(const int32_t *__restrict__ cum, int batch, int t, int &b, int &pos) {
    int lo = 0;
    for (int i = 0; i < batch; ++i)
        if (t >= cum[i]) lo = i;
    b = lo;
    pos = t - cum[lo];
}

// KV8: the caches are e4m3 bytes, stored = e4m3(x / scale) (same [L, bs, kvh, max_seq, hs] indexing in elements)

template <int HS, bool KV8>
__global__ __launch_bounds__(256) void prefill_rope_append_kernel(half_t *__restrict__ qkv, const half_t *__restrict__ bias,
                                                                  void *__restrict__ k_cache, void *__restrict__ v_cache,
                                                                  const int32_t *__restrict__ cum, const int32_t *__restrict__ hist,
                                                                  const float2 *__restrict__ rope, int batch, int head_num,
                                                                  int kv_head_num, int max_seq_len, int rotary_dim, size_t layer_off,
                                                                  float k_inv_scale, float v_inv_scale,
                                                                  const int32_t *__restrict__ table /* paged cache or null */,
                                                                  int max_pages) {
    const int t = blockIdx.x;
    int b, pos;
    locate_token(cum, batch, t, b, pos);
    const int tpos = hist[b] + pos;
    if (tpos < 0 || tpos >= max_seq_len) return;  // never write outside the slab
    constexpr int HALF = HS / 2;
    const int heads = head_num + 2 * kv_head_num;
    half_t *row = qkv + static_cast<size_t>(t) * heads * HS;
    const float2 *cs = rope + static_cast<size_t>(tpos) * HALF;
    // one work item = 8 consecutive dims d .. d+7 of the first half of a head and their rotate-half partners d+HS/2 ..:
    // two 16-byte loads, two 16-byte stores (the first version moved 2 bytes per access: 58 us per layer at 2048 tokens)
    constexpr int GROUPS = HALF / 8;
    for (int i = threadIdx.x; i < heads * GROUPS; i += 256) {
        const int h = i / GROUPS, d = (i - h * GROUPS) * 8;
        half_t *src = row + static_cast<size_t>(h) * HS;
        const half8_t lo = *reinterpret_cast<const half8_t *>(src + d), hi = *reinterpret_cast<const half8_t *>(src + d + HALF);
        half8_t blo = {0, 0, 0, 0, 0, 0, 0, 0}, bhi = blo;
        if (bias) {
            blo = *reinterpret_cast<const half8_t *>(bias + static_cast<size_t>(h) * HS + d);
            bhi = *reinterpret_cast<const half8_t *>(bias + static_cast<size_t>(h) * HS + d + HALF);
        }
        const bool rotate = h < head_num + kv_head_num;
        half8_t olo, ohi;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x0 = to_f32(lo[e]) + to_f32(blo[e]), x1 = to_f32(hi[e]) + to_f32(bhi[e]);
            float o0 = x0, o1 = x1;
            if (rotate && d + e < (rotary_dim >> 1)) {
                const float2 v = cs[d + e];
                o0 = x0 * v.x - x1 * v.y;
                o1 = x1 * v.x + x0 * v.y;
            }
            olo[e] = from_f32<half_t>(o0);
            ohi[e] = from_f32<half_t>(o1);
        }
        if (h < head_num) {
            *reinterpret_cast<half8_t *>(src + d) = olo;
            *reinterpret_cast<half8_t *>(src + d + HALF) = ohi;
        } else {
            const bool is_k = h < head_num + kv_head_num;
            const int g = is_k ? h - head_num : h - head_num - kv_head_num;
            // dense [bs, kvh, max_seq, hs] slab, or 128-token pages [num_pages, kvh, 128, hs] through the block table
            const size_t off = table ? layer_off + ((static_cast<size_t>(table[static_cast<size_t>(b) * max_pages + tpos / 128]) * kv_head_num + g) * 128 + tpos % 128) * HS
                                     : layer_off + ((static_cast<size_t>(b) * kv_head_num + g) * max_seq_len + tpos) * HS;
            if constexpr (KV8) {
                // quantise the fp16-rounded value (what the fp16 cache would hold), as the decode kernel does
                uint8_t *dst = static_cast<uint8_t *>(is_k ? k_cache : v_cache) + off;
                const float inv = is_k ? k_inv_scale : v_inv_scale;
                uint2 plo, phi;
                plo.x = pack4_e4m3(to_f32(olo[0]) * inv, to_f32(olo[1]) * inv, to_f32(olo[2]) * inv, to_f32(olo[3]) * inv);
                plo.y = pack4_e4m3(to_f32(olo[4]) * inv, to_f32(olo[5]) * inv, to_f32(olo[6]) * inv, to_f32(olo[7]) * inv);
                phi.x = pack4_e4m3(to_f32(ohi[0]) * inv, to_f32(ohi[1]) * inv, to_f32(ohi[2]) * inv, to_f32(ohi[3]) * inv);
                phi.y = pack4_e4m3(to_f32(ohi[4]) * inv, to_f32(ohi[5]) * inv, to_f32(ohi[6]) * inv, to_f32(ohi[7]) * inv);
                *reinterpret_cast<uint2 *>(dst + d) = plo;
                *reinterpret_cast<uint2 *>(dst + d + HALF) = phi;
            } else {
                half_t *dst = static_cast<half_t *>(is_k ? k_cache : v_cache) + off;
                *reinterpret_cast<half8_t *>(dst + d) = olo;
                *reinterpret_cast<half8_t *>(dst + d + HALF) = ohi;
            }
        }
    }
}

// Short prefills (<= 128 tokens: the QKV projection leaves split-K slabs): the slab consumer of the QKV projection with
// prefill_rope_append_kernel's work folded in -- one launch instead of splitk_finalize + the RoPE / append launch, and the k / v
// columns never travel through the packed QKV buffer.  Arithmetic per element = skinny_finalize_kernel's (s0 + s1 + ..., scale,
// one fp16 rounding) followed by prefill_rope_append_kernel's on that fp16 value: bit-identical to the two launches.
// grid (tokens, ceil(heads / 16)); an item = 4 consecutive dims d .. d + 3 (d < 64) of one head and their rotate-half partners d + 64 ..
template <bool KV8>
__global__ __launch_bounds__(256) void splitk_finalize_qkv_rope_kernel(const float *__restrict__ slab, int KS, int M, const SlabScale scale,
                                                                       half_t *__restrict__ qkv, const half_t *__restrict__ bias,
                                                                       void *__restrict__ k_cache, void *__restrict__ v_cache,
                                                                       const int32_t *__restrict__ cum, const int32_t *__restrict__ hist,
                                                                       const float2 *__restrict__ rope, int batch, int head_num, int kv_head_num,
                                                                       int max_seq_len, int rotary_dim, size_t layer_off, float k_inv_scale,
                                                                       float v_inv_scale, const int32_t *__restrict__ table, int max_pages) {
    constexpr int HS = 128, HALF = 64, GROUPS = HALF / 4;
    const int t = blockIdx.x;
    int b, pos;
    locate_token(cum, batch, t, b, pos);
    const int tpos = hist[b] + pos;
    if (tpos < 0 || tpos >= max_seq_len) return;  // never write outside the slab (prefill_rope_append_kernel)
    const int heads = head_num + 2 * kv_head_num, N = heads * HS;
    const size_t slab_sz = static_cast<size_t>(M) * N;
    const float2 *cs = rope + static_cast<size_t>(tpos) * HALF;
    // blockIdx.y = group of 16 heads: one item per thread, (heads / 16) x tokens workgroups (a single workgroup per token left half
    // the CUs idle at 128 tokens and walked its 6 items per thread one memory round trip after the other)
    {
        const int i = blockIdx.y * 256 + threadIdx.x;
        if (i >= heads * GROUPS) return;
        const int h = i / GROUPS, d = (i - h * GROUPS) * 4, n = h * HS + d;
        floatx4 v[2];
        for (int k0 = 0; k0 < KS; k0 += 8) {
            floatx4 p[8][2];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    p[kk][g] = *reinterpret_cast<const floatx4 *>(slab + static_cast<size_t>(min(k0 + kk, KS - 1)) * slab_sz +
                                                                  static_cast<size_t>(t) * N + n + g * HALF);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                if (k0 + kk < KS) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        if (k0 + kk == 0) v[g] = floatx4{0.f, 0.f, 0.f, 0.f} + p[kk][g];
                        else v[g] += p[kk][g];
                    }
                }
        }
        half4_t blo{0, 0, 0, 0}, bhi{0, 0, 0, 0};
        if (bias) {
            blo = *reinterpret_cast<const half4_t *>(bias + n);
            bhi = *reinterpret_cast<const half4_t *>(bias + n + HALF);
        }
        const bool rotate = h < head_num + kv_head_num;
        half4_t olo, ohi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // the projection's output as splitk_finalize stores it (fp16), then prefill_rope_append_kernel's arithmetic
            const float x0 = to_f32(from_f32<half_t>(scale.apply(v[0][e], t, n + e))) + to_f32(blo[e]);
            const float x1 = to_f32(from_f32<half_t>(scale.apply(v[1][e], t, n + e + HALF))) + to_f32(bhi[e]);
            float o0 = x0, o1 = x1;
            if (rotate && d + e < (rotary_dim >> 1)) {
                const float2 c = cs[d + e];
                o0 = x0 * c.x - x1 * c.y;
                o1 = x1 * c.x + x0 * c.y;
            }
            olo[e] = from_f32<half_t>(o0);
            ohi[e] = from_f32<half_t>(o1);
        }
        if (h < head_num) {
            half_t *dst = qkv + static_cast<size_t>(t) * N + n;
            *reinterpret_cast<half4_t *>(dst) = olo;
            *reinterpret_cast<half4_t *>(dst + HALF) = ohi;
        } else {
            const bool is_k = h < head_num + kv_head_num;
            const int g = is_k ? h - head_num : h - head_num - kv_head_num;
            const size_t off = table ? layer_off + ((static_cast<size_t>(table[static_cast<size_t>(b) * max_pages + tpos / 128]) * kv_head_num + g) * 128 + tpos % 128) * HS
                                     : layer_off + ((static_cast<size_t>(b) * kv_head_num + g) * max_seq_len + tpos) * HS;
            if constexpr (KV8) {
                uint8_t *dst = static_cast<uint8_t *>(is_k ? k_cache : v_cache) + off + d;
                const float inv = is_k ? k_inv_scale : v_inv_scale;
                *reinterpret_cast<unsigned *>(dst) = pack4_e4m3(to_f32(olo[0]) * inv, to_f32(olo[1]) * inv, to_f32(olo[2]) * inv, to_f32(olo[3]) * inv);
                *reinterpret_cast<unsigned *>(dst + HALF) = pack4_e4m3(to_f32(ohi[0]) * inv, to_f32(ohi[1]) * inv, to_f32(ohi[2]) * inv, to_f32(ohi[3]) * inv);
            } else {
                half_t *dst = static_cast<half_t *>(is_k ? k_cache : v_cache) + off + d;
                *reinterpret_cast<half4_t *>(dst) = olo;
                *reinterpret_cast<half4_t *>(dst + HALF) = ohi;
            }
        }
    }
}
bool splitk_finalize_qkv_rope_eligible(const SplitKSlabs &sk, int head_size, const void *qkv, const void *bias) {
    return head_size == 128 && sk.N % 128 == 0 && reinterpret_cast<uintptr_t>(sk.slab) % 16 == 0 &&
           (reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(bias)) % 8 == 0;
}
int splitk_finalize_qkv_rope(const SplitKSlabs &sk, const SlabScale &sc, half_t *qkv, const half_t *qkv_bias, void *k_cache, void *v_cache,
                             const int32_t *cum_seqlens, const int32_t *history_len, const float2 *rope, int layer, int batch, int head_num,
                             int kv_head_num, int max_seq_len, int rotary_dim, hipStream_t st, int kv_fp8, float k_scale, float v_scale,
                             const int32_t *block_table, int max_pages, int num_pages) {
    const size_t layer_off = block_table ? static_cast<size_t>(layer) * num_pages * kv_head_num * 128 * 128
                                         : static_cast<size_t>(layer) * batch * kv_head_num * max_seq_len * 128;
    const dim3 grid(sk.M, (sk.N / 128 + 15) / 16);
    if (kv_fp8)
        splitk_finalize_qkv_rope_kernel<true><<<grid, 256, 0, st>>>(sk.slab, sk.KS, sk.M, sc, qkv, qkv_bias, k_cache, v_cache, cum_seqlens, history_len,
                                                                    rope, batch, head_num, kv_head_num, max_seq_len, rotary_dim, layer_off,
                                                                    1.0f / k_scale, 1.0f / v_scale, block_table, max_pages);
    else
        splitk_finalize_qkv_rope_kernel<false><<<grid, 256, 0, st>>>(sk.slab, sk.KS, sk.M, sc, qkv, qkv_bias, k_cache, v_cache, cum_seqlens, history_len,
                                                                     rope, batch, head_num, kv_head_num, max_seq_len, rotary_dim, layer_off, 1.f, 1.f,
                                                                     block_table, max_pages);
    return launch_status("splitk_finalize(qkv + rope + append)");
}

// Per-token (sequence, cache position) table for the QKV projection's fused RoPE + append epilogue (gemm256.cuh
// g256_store_qkv_rope): tok_b[t] = sequence of packed token t, tok_tpos[t] = history + position.  Once per prefill call.
__global__ __launch_bounds__(256) void prefill_token_table_kernel(const int32_t *__restrict__ cum, const int32_t *__restrict__ hist, int batch,
                                                                  int num_tokens, int32_t *__restrict__ tok_b, int32_t *__restrict__ tok_tpos,
                                                                  const QkvRopeArgs args, QkvRopeArgs *__restrict__ args_dev) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t == 0) *args_dev = args;
    if (t >= num_tokens) return;
    int b, pos;
    locate_token(cum, batch, t, b, pos);
    tok_b[t] = b;
    tok_tpos[t] = hist[b] + pos;
}
int prefill_token_table(const int32_t *cum_seqlens, const int32_t *history_len, int batch, int num_tokens, int32_t *tok_b, int32_t *tok_tpos,
                        QkvRopeArgs args, QkvRopeArgs *args_dev, hipStream_t st) {
    args.tok_b = tok_b;
    args.tok_tpos = tok_tpos;
    prefill_token_table_kernel<<<(num_tokens + 255) / 256, 256, 0, st>>>(cum_seqlens, history_len, batch, num_tokens, tok_b, tok_tpos, args, args_dev);
    return launch_status("prefill_token_table");
}

// grid: (q tiles of NW*16 rows over max_q_len, head_num, batch); block = NW waves x 16 query rows.  NW = 8: a staged 64-key K/V
// tile (global -> LDS, two barriers: 80 of the 175 us of the 4-wave form at 2048 tokens) serves 128 query rows.
// RT = 16-row query tiles per wave (round 3): with RT = 2 every K / V^T fragment a wave reads from LDS feeds two MFMAs, and a
// 4-wave workgroup covers the same 128 query rows -- two such workgroups share a CU and run out of phase with each other.
template <int HS, bool KV8, int NW, int RT = 1>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2))) void prefill_flash_kernel(const half_t *__restrict__ qkv, const void *__restrict__ k_cache,
                                                            const void *__restrict__ v_cache, half_t *__restrict__ out,
                                                            const int32_t *__restrict__ cum, const int32_t *__restrict__ hist,
                                                            int head_num, int kv_head_num, int max_seq_len, size_t layer_off,
                                                            float k_scale, float v_scale,
                                                            const int32_t *__restrict__ table /* paged cache or null */, int max_pages) {
    static_assert(HS == 128, "tuned for head_size 128");
    // V rows are 288 bytes (256 + 32): the PV operand is fetched with ds_read_b64_tr_b16 (gfx950 transposed LDS read: a 16-lane
    // group reads a block of 4 keys x 16 head dims and every lane receives one column of it = 4 consecutive keys of its head
    // dim), whose 16 lanes address 4 rows x 4 eight-byte pieces -- rows 32 bytes apart modulo 256 keep a 32-lane half
    // conflict-free.  (The first version gathered the column with 8 scalar 2-byte LDS reads per fragment: 63 of 175 us.)
    constexpr int BQ = NW * 16 * RT, BT = 64, VSTRIDE = HS + 16, NTHR = NW * 64, WROWS = 16 * RT;   // WROWS = query rows per wave
    // two K/V tile buffers (2 x 34 KiB): tile t+1 is written while tile t is multiplied, ONE barrier per iteration; its
    // global loads are issued a full iteration earlier (registers), pinned ahead of the MFMAs with a scheduling barrier --
    // at 2 resident workgroups per CU nothing else hides an L2/HBM round trip
    __shared__ __attribute__((aligned(16))) half_t Ksb[2][BT * HS];
    __shared__ __attribute__((aligned(16))) half_t Vsb[2][BT * VSTRIDE];
    // Causal work grows with the query tile index (tile i multiplies 2(i+1) key tiles at 128 rows per tile) and one 8-wave
    // workgroup is resident per CU (166 VGPRs), so the 512 workgroups of 2048 tokens x 32 heads run as two dynamic rounds in
    // linear-id order.  Longest first: the ids walk the query tiles from the last (heaviest) to the first, all heads of a tile
    // index together, so the long workgroups start at once and the short ones fill in behind them (in plain order the
    // 32-key-tile workgroups of the last heads started last: 140 -> 92 us per layer at 2048 tokens).
    // (Two workgroups per CU -- K fragments read in halves, 128 VGPRs, 9 spilled -- measured slower: 106 us.)
    const int nx = gridDim.x, rows = gridDim.y * gridDim.z;
    const int lin = blockIdx.x + nx * (blockIdx.y + gridDim.y * blockIdx.z);
    const int kx = lin / rows, rowi = lin - kx * rows;
    // RT = 2 (4-wave workgroups, two resident per CU, all 512 of the 2048-token case at once): workgroups i and i + 256 share a CU,
    // so the second half of the ids walks the query tiles UPWARDS from the first -- heavy tile nx - 1 - k meets light tile k
    // (with the plain longest-first order two 32-key-tile workgroups shared the first CUs: 73 us instead of 61)
    const int nheavy = (nx + 1) / 2;
    const int xq = (RT == 1 || kx < nheavy) ? nx - 1 - kx : kx - nheavy;
    const int b = rowi / gridDim.y, h = rowi - b * gridDim.y;
    const int len = cum[b + 1] - cum[b], history = hist[b];
    const int q0 = xq * BQ;
    if (q0 >= len) return;
    const int ctx = history + len;
    const int g = h / (head_num / kv_head_num);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int heads = head_num + 2 * kv_head_num;
    // e4m3 cache: the scale operand of v_cvt_scalef32_pk_f16_fp8 is E8M0 -- only the exponent bits of the float are used -- so the
    // conversion applies the power-of-two part of each cache scale and the mantissa remainder (in [1, 2)) is applied in fp32:
    // K's in the softmax scale, V's on the output row
    const float k_p2 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, k_scale) & 0x7f800000u);
    const float v_p2 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v_scale) & 0x7f800000u);
    const float k_rem = KV8 ? k_scale / k_p2 : 1.f, v_rem = KV8 ? v_scale / v_p2 : 1.f;
    const float scale = rsqrtf(static_cast<float>(HS)) * k_rem;

    // this lane's query rows, one per row tile (clamped for the tail tile; their results are not stored)
    int qrow[RT], qpos[RT];
    half8_t qf[RT][4];
#pragma unroll
    for (int u = 0; u < RT; ++u) {
        qrow[u] = q0 + wave * WROWS + u * 16 + r;
        const int qrow_c = min(qrow[u], len - 1);
        const half_t *qptr = qkv + (static_cast<size_t>(cum[b] + qrow_c) * heads + h) * HS;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[u][s] = *reinterpret_cast<const half8_t *>(qptr + 32 * s + 8 * q);
        qpos[u] = history + qrow_c;  // keys t <= qpos are visible
    }

    const half_t *kc = static_cast<const half_t *>(k_cache), *vc = static_cast<const half_t *>(v_cache);
    const uint8_t *kc8 = static_cast<const uint8_t *>(k_cache), *vc8 = static_cast<const uint8_t *>(v_cache);
    (void)kc; (void)vc; (void)kc8; (void)vc8;

    floatx4 o[RT][8];  // O^T tiles of row tile u: rows d = 16*dt + 4q + e, col = this lane's query row
    float m_run[RT], l_run[RT];
#pragma unroll
    for (int u = 0; u < RT; ++u) {
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) o[u][dt] = floatx4{0.f, 0.f, 0.f, 0.f};
        m_run[u] = -INFINITY;
        l_run[u] = 0.f;
    }

    const int t_hi = min(ctx, history + min(q0 + BQ, len));  // keys needed by any row of this q tile
    constexpr int NCH = 1024 / NTHR;  // 16-byte chunks of each tile per thread
    half8_t kreg[NCH], vreg[NCH];
    uint2 kreg8[NCH], vreg8[NCH];
    (void)kreg; (void)vreg; (void)kreg8; (void)vreg8;
    auto fetch_tile = [&](int t0) {  // global -> registers
        // One page lookup per TILE, wave-uniform (a scalar load): the 64 keys of a tile lie in one 128-token page, and so does the
        // clamp target ctx - 1 of a tile's rows past the context (t0 <= ctx - 1 < t0 + 64).  Looked up per chunk through
        // row_off(), the table entry was a dependent VECTOR load in front of every K / V load with a vmcnt(0) between them.
        const size_t tile_base = table ? layer_off + (static_cast<size_t>(__builtin_amdgcn_readfirstlane(table[static_cast<size_t>(b) * max_pages + t0 / 128])) * kv_head_num + g) * 128 * HS
                                       : layer_off + (static_cast<size_t>(b) * kv_head_num + g) * max_seq_len * HS;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = tid + NTHR * i, row = id >> 4, ch = id & 15;
            const int t = min(t0 + row, ctx - 1);
            const size_t off = tile_base + static_cast<size_t>(table ? t % 128 : t) * HS;
            if constexpr (KV8) {
                kreg8[i] = *reinterpret_cast<const uint2 *>(kc8 + off + ch * 8);
                vreg8[i] = *reinterpret_cast<const uint2 *>(vc8 + off + ch * 8);
            } else {
                kreg[i] = *reinterpret_cast<const half8_t *>(kc + off + ch * 8);
                vreg[i] = *reinterpret_cast<const half8_t *>(vc + off + ch * 8);
            }
        }
    };
    auto publish_tile = [&](int buf) {  // registers -> LDS: K in swizzled 16-byte chunks, V row-major with padded rows
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = tid + NTHR * i, row = id >> 4, ch = id & 15;
            half8_t kvv, vvv;
            if constexpr (KV8) {
                // 8 e4m3 bytes -> 8 halves (x scale): the tiles in LDS are fp16 either way
                const uint2 kb = kreg8[i], vb = vreg8[i];
                const half2_t k0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(kb.x), k_p2, false);
                const half2_t k1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(kb.x), k_p2, true);
                const half2_t k2 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(kb.y), k_p2, false);
                const half2_t k3 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(kb.y), k_p2, true);
                const half2_t v0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(vb.x), v_p2, false);
                const half2_t v1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(vb.x), v_p2, true);
                const half2_t v2 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(vb.y), v_p2, false);
                const half2_t v3 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(vb.y), v_p2, true);
                kvv = half8_t{k0[0], k0[1], k1[0], k1[1], k2[0], k2[1], k3[0], k3[1]};
                vvv = half8_t{v0[0], v0[1], v1[0], v1[1], v2[0], v2[1], v3[0], v3[1]};
            } else {
                kvv = kreg[i];
                vvv = vreg[i];
            }
            *reinterpret_cast<half8_t *>(Ksb[buf] + row * HS + ((ch ^ (row & 15)) << 3)) = kvv;
            *reinterpret_cast<half8_t *>(Vsb[buf] + row * VSTRIDE + ch * 8) = vvv;
        }
    };
#ifdef FLASH_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    fetch_tile(0);
    publish_tile(0);
    if (BT < t_hi) fetch_tile(BT);
    for (int t0 = 0, it = 0; t0 < t_hi; t0 += BT, ++it) {
        const half_t *Ks = Ksb[it & 1], *Vs = Vsb[it & 1];
        FLASH_T(ta);
        __syncthreads();  // tile `it` published by every thread; every wave done with tile it-1 (the other buffer)
        FLASH_T(tb);
        FLASH_ACC(0, ta, tb);
#ifndef FLASH_SKIP_STAGE   // (timing experiment only, wrong results: no staging inside the loop)
        if (t0 + BT < t_hi) {
            publish_tile((it + 1) & 1);                        // tile it+1: registers (fetched last iteration) -> other buffer
            if (t0 + 2 * BT < t_hi) fetch_tile(t0 + 2 * BT);   // tile it+2: in flight under this iteration's MFMAs
        }
#endif
        __builtin_amdgcn_sched_barrier(0);  // keep the loads above the compute below
        FLASH_T(tc);
        FLASH_ACC(1, tb, tc);
        // key tiles entirely in the future of this wave's query rows are skipped by the whole wave (it still takes part in
        // the staging and the barriers above)
        if (t0 > history + q0 + wave * WROWS + WROWS - 1) continue;
        // ---- S^T = K . Q^T : 4 key tiles of 16, 4 k-steps over the head dim, RT row tiles per fragment ----
        // all 16 K fragments of the tile are read before the first MFMA (the compiler's own order was read -> wait -> MFMA,
        // one LDS round trip per MFMA)
        // RT = 1: all 16 K fragments of the tile are read before the first MFMA (the compiler's own order was read -> wait -> MFMA, one
        // LDS round trip per MFMA).  RT = 2: two batches of 8 fragments -- k-steps 0-1, then 2-3, of all four key tiles: the
        // accumulators stay 4 RT independent chains -- 32 instead of 64 fragment registers beside 64 output and 32 logit accumulators.
        constexpr int KSB = RT == 1 ? 1 : 2, KSS = 4 / KSB;   // batches, k-steps per batch
        floatx4 sacc[RT][4];
#pragma unroll
        for (int sb = 0; sb < KSB; ++sb) {
            half8_t kfr[4][KSS];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int row = tt * 16 + r;
#pragma unroll
                for (int s = 0; s < KSS; ++s)
                    kfr[tt][s] = *reinterpret_cast<const half8_t *>(Ks + row * HS + ((((sb * KSS + s) * 4 + q) ^ (row & 15)) << 3));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KSS; ++s)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int u = 0; u < RT; ++u)   // (first k-step: the zero C operand is an inline constant, no accumulator to clear)
                        sacc[u][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kfr[tt][s], qf[u][sb * KSS + s],
                                                                             (sb == 0 && s == 0) ? floatx4{0.f, 0.f, 0.f, 0.f} : sacc[u][tt], 0, 0, 0);
            if constexpr (KSB > 1) __builtin_amdgcn_sched_barrier(0);
        }
#ifdef FLASH_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(sacc[0][0]), "+v"(sacc[0][1]), "+v"(sacc[0][2]), "+v"(sacc[0][3]));
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // (MFMA results written: the next read of them would wait anyway)
#endif
        FLASH_T(td);
        FLASH_ACC(2, tc, td);
        // lane holds S[qrow[u]][t = t0 + 16 tt + 4 q + e].  Softmax in the log2 domain, p = 2^(S * scale2 - m).  VALU diet (round 3:
        // the loop issued ~330 VALU slots per wave and tile against 32 MFMAs): the row maximum is taken over the RAW
        // logits and scaled once (scale2 > 0), the scale rides in the exponent's FMA, numerators are converted to fp16 in pairs;
        // the causal / length mask is only evaluated on tiles that reach past the wave's first query position or the context end
        // (wave-uniform test), and 2^(-inf) = 0 needs no select.
        const float scale2 = scale * 1.44269504088896341f;
        const bool need_mask = t0 + BT - 1 > history + q0 + wave * WROWS || t0 + BT > ctx;
        half8_t pf[RT][2];  // P^T fragments of row tile u = MFMA B operand of the two 32-key steps
        typedef float float2v_t __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int u = 0; u < RT; ++u) {
            float mloc = -INFINITY;
            if (need_mask) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int t = t0 + tt * 16 + 4 * q + e;
                        const float v = (t <= qpos[u] && t < ctx) ? sacc[u][tt][e] : -INFINITY;
                        sacc[u][tt][e] = v;
                        mloc = fmaxf(mloc, v);
                    }
            } else {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) mloc = fmaxf(mloc, sacc[u][tt][e]);
            }
            // maximum over the 4 lanes of a query row (lanes r, r + 16, r + 32, r + 48) with the gfx950 row-swap instructions (VALU:
            // v_permlane32_swap pairs every lane with the one 32 lanes away, v_permlane16_swap with the one 16 away) instead of two
            // dependent ds_bpermute round trips through the LDS crossbar, each behind an lgkmcnt(0) that also drains the wave's reads
            mloc = lane_xor_max<16>(lane_xor_max<32>(mloc));   // (device_utils.cuh)
            mloc *= scale2;   // (-inf stays -inf)
            const float m_new = fmaxf(m_run[u], mloc);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;  // fully masked so far: keep everything at zero
            const float alpha = (m_run[u] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run[u] - m_use);
            float lsum = 0.f;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    // p = 2^(s * scale2 - m): one FMA + one v_exp_f32 per element (2^(-inf) = 0 for masked keys)
                    const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[u][tt][e], scale2, -m_use));
                    const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[u][tt][e + 1], scale2, -m_use));
                    lsum += p0 + p1;
                    const half2_t hh = __builtin_convertvector(float2v_t{p0, p1}, half2_t);   // v_cvt_pk_f16_f32 (round to nearest even)
                    pf[u][tt >> 1][(tt & 1) * 4 + e] = hh[0];
                    pf[u][tt >> 1][(tt & 1) * 4 + e + 1] = hh[1];
                }
            // l_run is this lane's PARTIAL row sum (its 16 of the tile's 64 keys): alpha is the same for the 4 lanes of a row, so
            // the cross-lane reduction is linear and done once after the last tile
            l_run[u] = l_run[u] * alpha + lsum;
            m_run[u] = m_new;
            // skip the 32 rescaling multiplies when no row of the wave needs them
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) o[u][dt] *= alpha;
            }
        }
#ifdef FLASH_STAMPS
        asm volatile("" : "+v"(pf[0][0]), "+v"(pf[0][1]));
#endif
        FLASH_T(te);
        FLASH_ACC(3, td, te);
        // ---- O^T += V^T . P^T : A fragment = V[t(q,j)][d = 16 dt + r] gathered down a column, shared by the RT row tiles ----
        // lane (r, q) of its 16-lane group supplies the address of block row (r >> 2), columns 4 (r & 3) .. +3, and receives
        // column r of the block's 4 rows; all 16 transposed reads of a 32-key step are issued before its 8 RT MFMAs
        typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
        typedef __attribute__((address_space(3))) fp16x4_t *lds_fp16x4_ptr;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const half_t *blk = Vs + (32 * c + 4 * q + (r >> 2)) * VSTRIDE + 4 * (r & 3);
            fp16x4_t lo[8], hi[8];
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                lo[dt] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4_ptr)(blk + dt * 16));
#ifdef FLASH_SKIP_V   // timing experiment only (wrong results): half of the V^T fragment reads
                hi[dt] = lo[dt];
#else
                hi[dt] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4_ptr)(blk + 16 * VSTRIDE + dt * 16));
#endif
            }
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                struct V8 { fp16x4_t lo, hi; };   // the two transposed reads side by side ARE the 8-half A operand: no repacking
                const half8_t vf = __builtin_bit_cast(half8_t, V8{lo[dt], hi[dt]});
#pragma unroll
                for (int u = 0; u < RT; ++u) o[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[u][c], o[u][dt], 0, 0, 0);
            }
        }
#ifdef FLASH_STAMPS
        asm volatile("" : "+v"(o[0][0]), "+v"(o[0][7]));
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#endif
        FLASH_T(tf);
        FLASH_ACC(4, te, tf);
        FLASH_ACC(5, ta, tf);
    }
#ifdef FLASH_STAMPS
    if (lane == 0 && lin < 64) {
        for (int i = 0; i < 8; ++i) flash_stamp_buf[(lin * 8 + wave) * 8 + i] = stamp_acc[i];
    }
#endif
#pragma unroll
    for (int u = 0; u < RT; ++u) {
        const float l = lane_xor_sum<32>(lane_xor_sum<16>(l_run[u]));
        if (qrow[u] < len) {
            const float inv = v_rem / (l + 1e-6f);  // the reference's denominator epsilon (scale_and_mask_and_softmax.cu:118)
            half_t *optr = out + (static_cast<size_t>(cum[b] + qrow[u]) * head_num + h) * HS;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                const half4_t v = {from_f32<half_t>(o[u][dt][0] * inv), from_f32<half_t>(o[u][dt][1] * inv),
                                   from_f32<half_t>(o[u][dt][2] * inv), from_f32<half_t>(o[u][dt][3] * inv)};
                *reinterpret_cast<half4_t *>(optr + dt * 16 + 4 * q) = v;
            }
        }
    }
}

int prefill_attention_f16(half_t *qkv, const half_t *qkv_bias, void *k_cache, void *v_cache, half_t *out,
                          const int32_t *cum_seqlens, const int32_t *history_len, const float2 *rope, int layer, int batch,
                          int num_tokens, int max_q_len, int head_num, int kv_head_num, int head_size, int max_seq_len,
                          int rotary_dim, hipStream_t st, int kv_fp8, float k_scale, float v_scale, const int32_t *block_table,
                          int max_pages, int num_pages, int rope_done) {
    if (head_size != 128) {
        set_error("prefill attention: head_size %d not supported by the flash kernel (128 only)", head_size);
        return LLMIE_ERR_UNSUPPORTED;
    }
    const size_t layer_off = block_table ? static_cast<size_t>(layer) * num_pages * kv_head_num * 128 * head_size
                                         : static_cast<size_t>(layer) * batch * kv_head_num * max_seq_len * head_size;
    // 128 query rows per workgroup either way: 8 waves x 16 rows (one workgroup per CU at a time, dynamic rounds, longest first), or
    // 4 waves x 2 x 16 rows (every fragment read from LDS feeds two MFMAs; two workgroups per CU, out of phase with each other).
    // Measured (bench, same box, us per layer): 8 x 512 tokens 54.5 -> 47.5 and 16 x 256 tokens 38 with the 4-wave form; 1 x 2048
    // 61 -> 70 -- there all 512 workgroups are resident at once, a CU's pair is one long and one short workgroup and the long one
    // runs most of its tiles alone at one wave per SIMD; 4 x 1024 72 = 72, 2 x 1024 38 -> 41.  Rule: the 4-wave form up to
    // kFlashRt2MaxQ query rows per sequence.
    constexpr int kFlashRt2MaxQ = 512;
    const bool rt2 = max_q_len <= kFlashRt2MaxQ;
    // Fewer 128-row workgroups than CUs (one short prompt): 4 waves x 16 rows = 64 query rows per workgroup, twice the workgroups
    // (interleaved A/B, 7B, one sequence: 128 tokens 27.8k -> 28.1k tok/s, 256 32.6k -> 32.9k, 512 46.0k -> 46.6k, 768 equal)
    const bool bq64 = ((max_q_len + 127) / 128) * head_num * batch < 256;
    const int bq = bq64 ? 64 : 128;
    dim3 grid((max_q_len + bq - 1) / bq, head_num, batch);
    const float ks = kv_fp8 ? k_scale : 1.f, vs = kv_fp8 ? v_scale : 1.f;
    if (rope_done) {   // the QKV projection's epilogue rotated q in `qkv` and wrote k / v into the caches
    } else if (kv_fp8)
        prefill_rope_append_kernel<128, true><<<num_tokens, 256, 0, st>>>(qkv, qkv_bias, k_cache, v_cache, cum_seqlens, history_len,
                                                                          rope, batch, head_num, kv_head_num, max_seq_len,
                                                                          rotary_dim, layer_off, 1.0f / ks, 1.0f / vs, block_table, max_pages);
    else
        prefill_rope_append_kernel<128, false><<<num_tokens, 256, 0, st>>>(qkv, qkv_bias, k_cache, v_cache, cum_seqlens, history_len,
                                                                           rope, batch, head_num, kv_head_num, max_seq_len,
                                                                           rotary_dim, layer_off, 1.f, 1.f, block_table, max_pages);
#define LLMIE_FLASH(KV8_, NW_, RT_)                                                                                                      \
    prefill_flash_kernel<128, KV8_, NW_, RT_><<<grid, NW_ * 64, 0, st>>>(qkv, k_cache, v_cache, out, cum_seqlens, history_len, head_num, \
                                                                         kv_head_num, max_seq_len, layer_off, ks, vs, block_table, max_pages)
    if (kv_fp8 && bq64) LLMIE_FLASH(true, 4, 1);
    else if (bq64) LLMIE_FLASH(false, 4, 1);
    else if (kv_fp8 && rt2) LLMIE_FLASH(true, 4, 2);
    else if (kv_fp8) LLMIE_FLASH(true, 8, 1);
    else if (rt2) LLMIE_FLASH(false, 4, 2);
    else LLMIE_FLASH(false, 8, 1);
#undef LLMIE_FLASH
    return launch_status("prefill_attention");
}

}  // namespace llmie

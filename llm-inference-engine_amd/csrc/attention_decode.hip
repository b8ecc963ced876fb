// Fused decode attention for gfx950: llmie_decoder_mha
// (replaces launchDecoderMaskedMultiHeadAttention, decoder_self_attention.cu:56-270).
//
// Math (reference fp32 kernel, with the batch-stride / step<=head_size / GQA-race defects of
// SURVEY 9-K4 fixed):   q,k,v (+bias)  ->  cache[layer,b,g,step-1,:] = k,v  ->
//   logit[t] = (q . K[t]) / sqrt(hs), t < step  ->  p = exp(l - max) / (sum + 1e-6)  ->  out = sum p V.
//
// Design: flash-decoding.  The KV range [0, step) of one (batch, kv-head) is split into chunks of
// CHUNK tokens; one 256-thread workgroup per chunk streams its K and V rows with 16-byte loads
// straight to VGPRs (a wave instruction covers 64/LPT whole rows = 1 KiB contiguous; every load
// of the chunk is issued before the first use), keeps all REP query heads of the kv head in
// registers (GQA reads each K/V byte once), does the softmax with wave64 shuffles, and writes an
// (m, l, o[hs]) partial.  A second tiny kernel merges the partials.  bs*kvh*splits workgroups
// fill the 256 CUs even at batch 1 (7B, S=2048: 32*16 = 512 workgroups).
// HBM-bound: algorithmic bytes = 2 * step * kvh * hs * sizeof(T) per sequence.
#include "llmie_internal.h"

#include <cstdlib>
#include <type_traits>

namespace llmie {

// KV cache element formats: T itself (fp32 / fp16) or e4m3 bytes (stored = e4m3(x / scale), one static scale per cache)
struct fp8kv_t {
    uint8_t b;
};
template <typename KT> struct CacheVec {
    using type = typename Vec16<KT>::type;
    static constexpr int n = Vec16<KT>::n;
};
template <> struct CacheVec<fp8kv_t> {
    using type = uint4_t;
    static constexpr int n = 16;
};
// the N elements of one 16-byte cache vector as fp32 (fp8: unscaled; the scale is folded into q / the output)
template <typename KT, int N> __device__ __forceinline__ void kv_to_f32(const typename CacheVec<KT>::type &v, float (&o)[N]) {
    if constexpr (std::is_same<KT, fp8kv_t>::value) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8(static_cast<int>(v[w]), false);
            const auto hi = __builtin_amdgcn_cvt_pk_f32_fp8(static_cast<int>(v[w]), true);
            o[4 * w] = lo[0];
            o[4 * w + 1] = lo[1];
            o[4 * w + 2] = hi[0];
            o[4 * w + 3] = hi[1];
        }
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e) o[e] = to_f32(v[e]);
    }
}
// N values of the activation type -> one 16-byte cache vector (fp8: x * inv_scale, saturating e4m3)
template <typename KT, typename T, int N> __device__ __forceinline__ typename CacheVec<KT>::type kv_pack(const T (&x)[N], float inv_scale) {
    typename CacheVec<KT>::type v;
    if constexpr (std::is_same<KT, fp8kv_t>::value) {
#pragma unroll
        for (int w = 0; w < 4; ++w)
            v[w] = pack4_e4m3(to_f32(x[4 * w]) * inv_scale, to_f32(x[4 * w + 1]) * inv_scale, to_f32(x[4 * w + 2]) * inv_scale,
                              to_f32(x[4 * w + 3]) * inv_scale);
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = x[e];
    }
    return v;
}
// N consecutive activation elements (N * sizeof(T) bytes = one or two 16-byte loads)
template <typename T, int N> __device__ __forceinline__ void load_elems(const T *p, T (&dst)[N]) {
    constexpr int PER = Vec16<T>::n;
    static_assert(N % PER == 0, "whole 16-byte loads");
#pragma unroll
    for (int c = 0; c < N / PER; ++c) {
        const typename Vec16<T>::type v = reinterpret_cast<const typename Vec16<T>::type *>(p)[c];
#pragma unroll
        for (int e = 0; e < PER; ++e) dst[c * PER + e] = v[e];
    }
}

// geometry of the split kernel: NWV waves per workgroup, GL K (and V) 16-byte loads in flight per lane
template <typename KT, int HS, int NWV = 4, int GL = 8> struct AttnGeom {
    static constexpr int N = CacheVec<KT>::n;        // cache elements per 16-byte load = head dims per lane
    static constexpr int LPT = HS / N;               // lanes per token row
    static constexpr int TPW = 64 / LPT;             // token rows per wave instruction
    static constexpr int CHUNK = NWV * GL * TPW;     // tokens per workgroup
};

__host__ __device__ inline int attn_min_chunk() { return 32; }

// q/k/v source when the QKV projection's split-K finalize is fused into the attention: fp32 slabs [KS][batch][qkv_dim]
struct QkvSlabs {
    const float *slab;
    int KS;
    size_t stride;          // floats between slabs (= batch * qkv_dim)
    SlabScale sc;           // scales of the projection's weight format (int8: per channel; fp8: channel x token)
};

// Paged KV cache: pool [L][num_pages][kvh][KV_PAGE tokens][hs]; block_table[b * max_pages + p] = pool page of
// the p-th page of sequence b.  table == null: the reference's dense [L][batch][kvh][max_seq][hs] slab.
constexpr int KV_PAGE = 128;
struct PagedKv {
    const int32_t *table;
    int max_pages;
    // (two more launch-wide layout facts ride along with the cache layout)
    int step_stride;   // 0: one position for the whole batch (step / *step_dev); 1: ragged batch, step_dev[b] = context length
                       // of sequence b INCLUDING this step's token
    int out_x32;       // the output rows go to the x32 activation image of the packed projections (<= 32 sequences)
};
// position of sequence b (a value outside [1, max_seq_len] makes every workgroup of that sequence return without touching
// the caches or the output: a corrupt device-resident position must not become an out-of-bounds append)
__device__ __forceinline__ int seq_step(const int32_t *step_dev, int step_arg, int stride, int b, int max_seq_len) {
    const int s = step_dev ? step_dev[static_cast<size_t>(b) * stride] : step_arg;
    return (s >= 1 && s <= max_seq_len) ? s : 0;
}
// element index of (sequence b, feature k) in the attention output: row-major [batch][width] or the x32 image
__device__ __forceinline__ size_t attn_out_index(int x32, int b, int k, int width) {
    if (!x32) return static_cast<size_t>(b) * width + k;
    return (static_cast<size_t>(k >> 5) * 2 + (b >> 4)) * 512 + ((k & 31) >> 3) * 128 + (b & 15) * 8 + (k & 7);
}

// Merge of the per-split (m, l, o[d]) partials of one (batch, head) for output dim d: 16 splits per round,
// every load of a round issued before the first use.  Shared by the stand-alone merge kernel and by the
// in-kernel merge of the last-arriving workgroup, so both give bit-identical results.
// one round of the merge: the 16 (m, l, o[d]) triples of splits s0 .. s0 + 15 (those >= nsplits are ignored) enter (M, L, o)
__device__ __forceinline__ void merge_round(const float (&ms)[16], const float (&ls)[16], const float (&os)[16], int s0, int nsplits,
                                            float &M, float &L, float &o) {
    float Mc = M;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (s0 + i < nsplits) Mc = fmaxf(Mc, ms[i]);
    const float rescale = (M == -INFINITY) ? 0.f : __expf(M - Mc);
    L *= rescale;
    o *= rescale;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (s0 + i < nsplits) {
            const float f = __expf(ms[i] - Mc);
            L = fmaf(f, ls[i], L);
            o = fmaf(f, os[i], o);
        }
    M = Mc;
}
// loads of one round; slots are clamped to `last` (nsplits - 1, or the last slot of the buffer when the loads are issued before
// the number of splits is known: what a clamped-away slot holds is never used)
__device__ __forceinline__ void merge_load(const float *__restrict__ p, int s0, int last, size_t stride, int d, float (&ms)[16],
                                           float (&ls)[16], float (&os)[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int s = min(s0 + i, last);
        ms[i] = p[s * stride];
        ls[i] = p[s * stride + 1];
        os[i] = p[s * stride + 2 + d];
    }
}
__device__ __forceinline__ float merge_splits(const float *__restrict__ p, int nsplits, size_t stride, int d) {
    float M = -INFINITY, L = 0.f, o = 0.f;
    for (int s0 = 0; s0 < nsplits; s0 += 16) {
        float ms[16], ls[16], os[16];
        merge_load(p, s0, nsplits - 1, stride, d, ms, ls, os);
        merge_round(ms, ls, os, s0, nsplits, M, L, o);
    }
    return o / (L + 1e-6f);
}

template <typename T, int HS, int REP, int kAttnWaves = 4, int kAttnG = 8, typename KT = T>
__global__ __launch_bounds__(kAttnWaves * 64) void decode_attn_split_kernel(
    const T *__restrict__ qkv, const T *__restrict__ qkv_bias, KT *k_cache, KT *v_cache,
    float *__restrict__ part, T *__restrict__ out, int head_num, int kv_head_num, int max_seq_len,
    int step_arg, const int32_t *__restrict__ step_dev, int max_splits,
    const float2 *__restrict__ rope /* [max_pos][HS/2] (cos,sin) or null */, int rotary_dim,
    int32_t *tickets /* [batch, kv_head_num] zero-initialised arrival counters, or null = separate merge kernel */,
    const QkvSlabs qs /* qs.slab != null: q/k/v come from the split-K partial slabs of the QKV projection (qkv unused) */,
    const float k_scale, const float v_scale /* fp8 cache: stored = e4m3(x / scale); 1 otherwise */,
    const PagedKv pg /* pg.table != null: k_cache / v_cache are this layer's page pools */,
    const int cpw /* chunks of CHUNK tokens one workgroup walks through (online softmax across them): large batches amortise
                     the per-workgroup prologue / merge (~550 of ~1000 VALU instructions per wave at one chunk) */) {
    using G = AttnGeom<KT, HS, kAttnWaves, kAttnG>;
    using V = typename CacheVec<KT>::type;  // one 16-byte vector of cache elements
    constexpr int N = G::N, LPT = G::LPT, TPW = G::TPW, CHUNK = G::CHUNK;
    constexpr int NT = kAttnWaves * 64;
    constexpr bool FP8KV = std::is_same<KT, fp8kv_t>::value;
    static_assert(HS % N == 0 && LPT >= 1 && LPT <= 64 && (LPT & (LPT - 1)) == 0, "head size");

    const int split = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
    const int step = seq_step(step_dev, step_arg, pg.step_stride, b, max_seq_len);
    const int span = cpw * CHUNK;   // tokens of one workgroup
    const int t0 = split * span;
    if (t0 >= step) return;  // whole workgroup exits together (also: invalid position)
    int tc0 = t0;                             // first token of the chunk being processed
    int t_end = min(step, tc0 + CHUNK);       // end of that chunk
    const int nsplits = (step + span - 1) / span;
    const int batch = gridDim.z;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPT, dl = lane % LPT;
    const int qkv_heads = head_num + 2 * kv_head_num;
    const float scale = rsqrtf(static_cast<float>(HS));

    const T *row = qkv + static_cast<size_t>(b) * qkv_heads * HS;
    // q for the REP heads of this kv head, pre-scaled, fp32
    // Cache rows of this workgroup's chunk: one base pointer per 128-token page the chunk touches (1 or 2).  Dense layout: the
    // page-aligned pieces of the head's contiguous [max_seq][HS] slab; paged: looked up ONCE here in the block table (uniform
    // scalar loads), so the K/V loads below never depend on a table load.
    constexpr int NPG = CHUNK > KV_PAGE ? CHUNK / KV_PAGE : 1;
    static_assert(CHUNK % KV_PAGE == 0 || KV_PAGE % CHUNK == 0, "chunk vs page size");
    int pg0 = 0;
    KT *kbase[NPG], *vbase[NPG];
    auto set_pages = [&]() {   // of the chunk at tc0
        pg0 = tc0 / KV_PAGE;
#pragma unroll
        for (int j = 0; j < NPG; ++j) {
            size_t off;
            if (pg.table) {
                const int page = pg.table[static_cast<size_t>(b) * pg.max_pages + min(pg0 + j, pg.max_pages - 1)];
                off = (static_cast<size_t>(page) * kv_head_num + g) * KV_PAGE * HS;
            } else {
                off = (static_cast<size_t>(b) * kv_head_num + g) * max_seq_len * HS + static_cast<size_t>(pg0 + j) * KV_PAGE * HS;
            }
            kbase[j] = k_cache + off;
            vbase[j] = v_cache + off;
        }
    };
    set_pages();
    auto krow = [&](int t) { return kbase[NPG == 1 ? 0 : (t / KV_PAGE - pg0)] + static_cast<size_t>(t % KV_PAGE) * HS; };
    auto vrow = [&](int t) { return vbase[NPG == 1 ? 0 : (t / KV_PAGE - pg0)] + static_cast<size_t>(t % KV_PAGE) * HS; };
    KT *kc = kbase[0];  // a readable address of this head (dummy source of the unconditional loads below)
    const int t_new = step - 1;
    // small L2-resident operands first (q rows, RoPE row), then the K/V stream; q is processed after the K/V
    // loads have been issued, so its latency hides under theirs (vmcnt retires in order: q is older)
    // Slab mode (q/k/v still in the QKV projection's split-K slabs): the rows this workgroup needs -- REP q heads, plus
    // the new k and v rows when its chunk holds this step's token -- are reduced ONCE per workgroup into LDS by the
    // first threads (one float4 column each, all slab loads in flight together, summed in the finalize kernel's
    // order, scaled and rounded like it) instead of by every lane (16 lanes x 4 waves hold the same q slice).
    T qraw[REP][N];
    constexpr int ITEMS_PER_HEAD = HS / 4;
    constexpr int LROUNDS = ((REP + 2) * ITEMS_PER_HEAD + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) T qkvlds[(REP + 2) * HS];
    const bool wg_has_new = t_new >= t0 && t_new < t0 + span;  // workgroup-uniform
    floatx4 spart[LROUNDS][4];
    half4_t sscale[LROUNDS];
    floatx4 sscalef[LROUNDS];
    const float xs_b = (qs.slab && qs.sc.wf) ? qs.sc.xs[b] : 1.f;
    size_t scol[LROUNDS];
    bool sact[LROUNDS];
    if (qs.slab) {
#pragma unroll
        for (int lr = 0; lr < LROUNDS; ++lr) {
            const int item = lr * NT + threadIdx.x;
            const int hsel = item / ITEMS_PER_HEAD, d4 = item - hsel * ITEMS_PER_HEAD;
            sact[lr] = hsel < REP || (wg_has_new && hsel < REP + 2);
            const int hh = hsel < REP ? g * REP + hsel : (hsel == REP ? head_num + g : head_num + kv_head_num + g);
            scol[lr] = static_cast<size_t>(hh) * HS + d4 * 4;
            if (sact[lr]) {
                const float *p = qs.slab + static_cast<size_t>(b) * qkv_heads * HS + scol[lr];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    spart[lr][kk] = *reinterpret_cast<const floatx4 *>(p + static_cast<size_t>(min(kk, qs.KS - 1)) * qs.stride);
                if (qs.sc.wh) sscale[lr] = *reinterpret_cast<const half4_t *>(qs.sc.wh + scol[lr]);
                if (qs.sc.wf) sscalef[lr] = *reinterpret_cast<const floatx4 *>(qs.sc.wf + scol[lr]);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < REP; ++r) load_elems<T, N>(row + static_cast<size_t>(g * REP + r) * HS + dl * N, qraw[r]);
    }
    // this step's k/v rows and the bias slices: loaded by every workgroup, unconditionally (a load under a divergent
    // or data-dependent branch makes the compiler drain vmcnt at the join -- measured: the K/V stream below used to
    // stall after its second load).  Without a bias / outside slab mode the address is a dummy valid one (the head's
    // first cache row) and the value is dropped by a select.
    const int hk = head_num + g, hv = head_num + kv_head_num + g;
    const T *dummy = reinterpret_cast<const T *>(kc);  // N * sizeof(T) <= 32 readable bytes at the head's first cache rows
    T knraw[N], vnraw[N], qbias[REP][N], kbias[N], vbias[N];
    load_elems<T, N>(qs.slab ? dummy : row + static_cast<size_t>(hk) * HS + dl * N, knraw);
    load_elems<T, N>(qs.slab ? dummy : row + static_cast<size_t>(hv) * HS + dl * N, vnraw);
#pragma unroll
    for (int r = 0; r < REP; ++r)
        load_elems<T, N>(qkv_bias ? qkv_bias + static_cast<size_t>(g * REP + r) * HS + dl * N : dummy, qbias[r]);
    load_elems<T, N>(qkv_bias ? qkv_bias + static_cast<size_t>(hk) * HS + dl * N : dummy, kbias);
    load_elems<T, N>(qkv_bias ? qkv_bias + static_cast<size_t>(hv) * HS + dl * N : dummy, vbias);
    float2 csraw[N];
    if (rope) {
        const float2 *cs = rope + static_cast<size_t>(t_new) * (HS / 2) + (dl % (LPT / 2)) * N;
#pragma unroll
        for (int e = 0; e < N; ++e) csraw[e] = cs[e];
    }
    // ---- issue every K and V load of this wave's token range ----
    V kv[kAttnG], vv[kAttnG];
    int tok[kAttnG];
    auto issue_loads = [&]() {   // of the chunk at tc0 (pages set)
#pragma unroll
        for (int i = 0; i < kAttnG; ++i) {
            const int t = tc0 + (wave * kAttnG + i) * TPW + sub;
            tok[i] = t;
            // rows past the chunk end re-read its last row (masked below); the slot of this step's token is read as it is
            // (stale, replaced below): every load is unconditional so all 2*G of them are in flight together
            kv[i] = load_nt(reinterpret_cast<const V *>(krow(min(t, t_end - 1))) + dl);
        }
#pragma unroll
        for (int i = 0; i < kAttnG; ++i)
            vv[i] = load_nt(reinterpret_cast<const V *>(vrow(min(tok[i], t_end - 1))) + dl);
    };
    issue_loads();
    if (qs.slab) {
#pragma unroll
        for (int lr = 0; lr < LROUNDS; ++lr) {
            if (sact[lr]) {
                const float *p = qs.slab + static_cast<size_t>(b) * qkv_heads * HS + scol[lr];
                floatx4 f = spart[lr][0];
#pragma unroll
                for (int kk = 1; kk < 4; ++kk)
                    if (kk < qs.KS) f += spart[lr][kk];
                for (int k = 4; k < qs.KS; ++k) f += *reinterpret_cast<const floatx4 *>(p + static_cast<size_t>(k) * qs.stride);
                const int item = lr * NT + threadIdx.x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = f[e];
                    if (qs.sc.wh) v *= to_f32(sscale[lr][e]);          // the arithmetic of SlabScale::apply
                    if (qs.sc.wf) v *= sscalef[lr][e] * xs_b;
                    qkvlds[item * 4 + e] = from_f32<T>(v);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < REP; ++r) load_elems<T, N>(&qkvlds[r * HS + dl * N], qraw[r]);
    }
    // RoPE (fused form of launchRope, rope.cu:4-43): rotate-half pairs (d, d+HS/2) live LPT/2 lanes apart
    const bool rope_first = dl < LPT / 2;
    float rc[N], rs[N];
    if (rope) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const bool rot = ((dl % (LPT / 2)) * N + e) < (rotary_dim >> 1);
            const float2 v = rot ? csraw[e] : float2{1.f, 0.f};
            rc[e] = v.x;
            rs[e] = rope_first ? -v.y : v.y;
        }
    }
    auto rotate = [&](float (&x)[N]) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const float partner = (LPT / 2 < 16) ? lane_xor_lt16<(LPT / 2 < 16 ? LPT / 2 : 8)>(x[e]) : __shfl_xor(x[e], LPT / 2, 64);
            x[e] = x[e] * rc[e] + partner * rs[e];
        }
    };
    // e4m3 cache: the scale operand of v_cvt_scalef32_pk_f16_fp8 is E8M0 (only the exponent bits of the float are used), so the
    // conversion gets the power-of-two part of k_scale and q carries the mantissa remainder (in [1, 2)) in fp32
    const float k_p2 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, k_scale) & 0x7f800000u);
    const float k_q = FP8KV ? k_scale / k_p2 : k_scale;
    (void)k_p2;
    float qf[REP][N];
#pragma unroll
    for (int r = 0; r < REP; ++r) {
        float f[N];
#pragma unroll
        for (int e = 0; e < N; ++e) f[e] = to_f32(qraw[r][e]);
        if (rope) {
            rotate(f);
#pragma unroll
            for (int e = 0; e < N; ++e) f[e] = to_f32(from_f32<T>(f[e]));  // as stored by the unfused RoPE kernel
        }
#pragma unroll
        for (int e = 0; e < N; ++e) {
            f[e] += qkv_bias ? to_f32(qbias[r][e]) : 0.f;
            qf[r][e] = f[e] * (scale * k_q);  // the cache holds k / k_scale (fp8: the conversion applies the power-of-two part)
        }
    }
    // e4m3 cache: K.q on packed fp16 -- v_cvt_scalef32_pk_f16_fp8 turns two cache bytes into (k0, k1) * k_scale in one instruction
    // and v_dot2_f32_f16 adds both products to the fp32 logit: 16 instructions per 16-byte vector instead of 8 conversions + 16
    // fp32 FMAs.  q (already scaled by 1/sqrt(hs), O(0.1)) is rounded to fp16 for this: 2^-11 relative, far below the cache's
    // own e4m3 step (2^-4).
    half2_t qh[FP8KV ? REP : 1][FP8KV ? N / 2 : 1];
    if constexpr (FP8KV) {
#pragma unroll
        for (int r = 0; r < REP; ++r)
#pragma unroll
            for (int j = 0; j < N / 2; ++j) qh[r][j] = half2_t{from_f32<half_t>(qf[r][2 * j]), from_f32<half_t>(qf[r][2 * j + 1])};
    }

    // the token of this step comes from the qkv buffer / slabs (RoPE, then +bias, as the reference's rope.cu then
    // decoder_self_attention.cu:111-118) and is appended to the cache; computed by every lane, kept by the token's lanes
    V knv, vnv;   // in the cache's element format: what is stored is what this step attends to
    {
        T kn[N], vn[N];
#pragma unroll
        for (int e = 0; e < N; ++e) {
            kn[e] = knraw[e];
            vn[e] = vnraw[e];
        }
        if (qs.slab) {
            load_elems<T, N>(&qkvlds[REP * HS + dl * N], kn);
            load_elems<T, N>(&qkvlds[(REP + 1) * HS + dl * N], vn);
        }
        if (rope) {
            float f[N];
#pragma unroll
            for (int e = 0; e < N; ++e) f[e] = to_f32(kn[e]);
            rotate(f);
#pragma unroll
            for (int e = 0; e < N; ++e) kn[e] = from_f32<T>(f[e]);
        }
#pragma unroll
        for (int e = 0; e < N; ++e) {
            kn[e] = qkv_bias ? from_f32<T>(to_f32(kn[e]) + to_f32(kbias[e])) : kn[e];
            vn[e] = qkv_bias ? from_f32<T>(to_f32(vn[e]) + to_f32(vbias[e])) : vn[e];
        }
        knv = kv_pack<KT, T, N>(kn, 1.0f / k_scale);
        vnv = kv_pack<KT, T, N>(vn, 1.0f / v_scale);
    }
    auto inject = [&]() {   // chunk at tc0, its loads issued
        bool mine = false;
        if (t_new >= tc0 && t_new < tc0 + CHUNK) {   // workgroup-uniform: every other chunk skips 8 selects per loaded vector
#pragma unroll
        for (int i = 0; i < kAttnG; ++i) {
            const bool is_new = tok[i] == t_new;
            mine |= is_new;
            uint4_t kw = __builtin_bit_cast(uint4_t, kv[i]), vw = __builtin_bit_cast(uint4_t, vv[i]);
            const uint4_t knw = __builtin_bit_cast(uint4_t, knv), vnw = __builtin_bit_cast(uint4_t, vnv);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                kw[w] = is_new ? knw[w] : kw[w];
                vw[w] = is_new ? vnw[w] : vw[w];
            }
            kv[i] = __builtin_bit_cast(V, kw);
            vv[i] = __builtin_bit_cast(V, vw);
        }
        }
        if (mine) {
            reinterpret_cast<V *>(krow(t_new))[dl] = knv;
            reinterpret_cast<V *>(vrow(t_new))[dl] = vnv;
        }
    };
    inject();

    // running softmax state of this wave over the chunks of the workgroup: maximum (wave-uniform), and per lane the sum of
    // numerators and the weighted value sums of ITS token slots (reduced across the lanes once, after the last chunk)
    float mx[REP], acc[REP][N], ls[REP];
#pragma unroll
    for (int r = 0; r < REP; ++r) {
        mx[r] = -INFINITY;
        ls[r] = 0.f;
#pragma unroll
        for (int e = 0; e < N; ++e) acc[r][e] = 0.f;
    }
    auto accumulate = [&](const bool first) {   // chunk at tc0, its vectors landed and patched
    // ---- logits ----
    float lg[REP][kAttnG];
    float cm[REP];
#pragma unroll
    for (int r = 0; r < REP; ++r) cm[r] = -INFINITY;
#pragma unroll
    for (int i = 0; i < kAttnG; ++i) {
        const bool valid = tok[i] < t_end;
        float kf[FP8KV ? 1 : N];
        half2_t kh[FP8KV ? N / 2 : 1];
        if constexpr (FP8KV) {
            const uint4_t kw = __builtin_bit_cast(uint4_t, kv[i]);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                kh[2 * w] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(kw[w]), k_p2, false);
                kh[2 * w + 1] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(kw[w]), k_p2, true);
            }
        } else {
            kv_to_f32<KT, N>(kv[i], kf);
        }
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            float d = 0.f;
            if (valid) {
                if constexpr (FP8KV) {
#pragma unroll
                    for (int j = 0; j < N / 2; ++j) d = __builtin_amdgcn_fdot2(kh[j], qh[r][j], d, false);
                } else {
#pragma unroll
                    for (int e = 0; e < N; ++e) d = fmaf(qf[r][e], kf[e], d);
                }
            }
            d = group_sum<LPT>(d);
            lg[r][i] = valid ? d : -INFINITY;
            cm[r] = fmaxf(cm[r], lg[r][i]);
        }
    }
    // wave max over the TPW token slots (lanes differing in sub), then the running maximum; what was accumulated under the old
    // maximum is rescaled (nothing to rescale in a workgroup's first chunk)
#pragma unroll
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int o = LPT; o < 64; o <<= 1) cm[r] = lane_xor_max_o(cm[r], o);
        const float mnew = fmaxf(mx[r], cm[r]);
        if (!first) {
            const float f = (mx[r] == -INFINITY) ? 0.f : __expf(mx[r] - mnew);
            ls[r] *= f;
#pragma unroll
            for (int e = 0; e < N; ++e) acc[r][e] *= f;
        }
        mx[r] = mnew;
    }
    // ---- softmax numerators and P.V ----
#pragma unroll
    for (int i = 0; i < kAttnG; ++i) {
        const bool valid = tok[i] < t_end;
        if (valid) {
            float vf[N];
            kv_to_f32<KT, N>(vv[i], vf);
#pragma unroll
            for (int r = 0; r < REP; ++r) {
                const float p = __expf(lg[r][i] - mx[r]);
                ls[r] += p;
#pragma unroll
                for (int e = 0; e < N; ++e) acc[r][e] = fmaf(p, vf[e], acc[r][e]);
            }
        }
    }
    };
    accumulate(true);
    // the workgroup's further chunks: same loads, same arithmetic, the prologue above paid once.  e4m3 cache only: there the
    // fixed work is ~550 of ~1000 VALU instructions per wave and batch 32 x ctx 2048 went from 130 to 98 us per layer (4.1 ->
    // 5.5 TB/s); the fp16 cache streams at 6.2-6.3 TB/s with one chunk per workgroup and the loop's longer register live ranges
    // (146 -> 177 VGPRs, two workgroups per CU instead of three) cost 2.5 us per launch at ctx 128, so it is compiled out there.
    if constexpr (FP8KV) {
        for (int c = 1; c < cpw; ++c) {
            tc0 += CHUNK;
            if (tc0 >= step) break;   // workgroup-uniform
            t_end = min(step, tc0 + CHUNK);
            set_pages();
            issue_loads();
            inject();
            accumulate(false);
        }
    }
#pragma unroll
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int o = LPT; o < 64; o <<= 1) {
            // (v + v[lane ^ o] through the row swaps / DPP: no LDS round trips; bit-identical to the __shfl_xor butterfly)
            ls[r] = lane_xor_sum_o(ls[r], o);
#pragma unroll
            for (int e = 0; e < N; ++e) acc[r][e] = lane_xor_sum_o(acc[r][e], o);
        }
    }
    // ---- merge the 4 waves through LDS ----
    __shared__ float s_m[REP][kAttnWaves], s_l[REP][kAttnWaves];
    __shared__ float s_o[REP][kAttnWaves][HS];
    if (sub == 0) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int e = 0; e < N; ++e) s_o[r][wave][dl * N + e] = acc[r][e] * v_scale;  // the cache holds v / v_scale
            if (dl == 0) {
                s_m[r][wave] = mx[r];
                s_l[r][wave] = ls[r];
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < REP * HS; i += NT) {
        const int r = i / HS, d = i - r * HS;
        float M = s_m[r][0];
#pragma unroll
        for (int w = 1; w < kAttnWaves; ++w) M = fmaxf(M, s_m[r][w]);
        float L = 0.f, o = 0.f;
#pragma unroll
        for (int w = 0; w < kAttnWaves; ++w) {
            const float f = (s_m[r][w] == -INFINITY) ? 0.f : __expf(s_m[r][w] - M);
            L += f * s_l[r][w];
            o += f * s_o[r][w][d];
        }
        const int h = g * REP + r;
        if (nsplits == 1) {
            out[attn_out_index(pg.out_x32, b, h * HS + d, head_num * HS)] = from_f32<T>(o / (L + 1e-6f));
        } else {
            float *p = part + ((static_cast<size_t>(b) * head_num + h) * max_splits + split) * (HS + 2);
            p[2 + d] = o;
            if (d == 0) {
                p[0] = M;
                p[1] = L;
            }
        }
    }
    (void)batch;
    // ---- in-launch merge by the last-arriving workgroup of this (batch, kv head) ----
    // Placement-independent hand-off (cdna_hip_programming.md Guideline 16, counter form): plain partial stores ->
    // every storing wave drains vmcnt -> workgroup barrier -> lane 0: agent-scope release, drained, then ONE relaxed
    // agent-scope ticket -> the workgroup that draws nsplits-1 acquires (agent scope), barrier, reads all partials.
    if (tickets && nsplits > 1) {
        __shared__ int s_ticket;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int32_t *cnt = tickets + static_cast<size_t>(b) * kv_head_num + g;
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (s_ticket == nsplits - 1) {  // workgroup-uniform
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next launch
            }
            __syncthreads();
            const size_t stride = static_cast<size_t>(HS) + 2;
            for (int i = threadIdx.x; i < REP * HS; i += NT) {
                const int r = i / HS, d = i - r * HS;
                const int h = g * REP + r;
                const float *p = part + (static_cast<size_t>(b) * head_num + h) * max_splits * stride;
                const float v = merge_splits(p, nsplits, stride, d);
                out[attn_out_index(pg.out_x32, b, h * HS + d, head_num * HS)] = from_f32<T>(v);
            }
        }
    }
}

// merge the per-split partials: grid (head_num, batch), one thread per output dim.  Every thread reads the
// (m, l) pairs itself (broadcast loads) and its own o values, 16 splits per round with all loads issued before
// the first use, so the kernel is about one L2 round trip per 16 splits -- no LDS, no barrier.
template <typename T>
__global__ __launch_bounds__(256) void decode_attn_combine_kernel(const float *__restrict__ part,
                                                                  T *__restrict__ out, int head_num,
                                                                  int head_size, int chunk, int step_arg,
                                                                  const int32_t *__restrict__ step_dev,
                                                                  int max_splits, int step_stride, int max_seq_len, int out_x32) {
    const int h = blockIdx.x, b = blockIdx.y;
    const size_t stride = static_cast<size_t>(head_size) + 2;
    const float *p = part + (static_cast<size_t>(b) * head_num + h) * max_splits * stride;
    // The first round's loads do not wait for the device-resident position: they go out beside its load, clamped to the buffer
    // (a slot past the live splits holds an older step's values, which merge_round never looks at) -- one L2 round trip less in
    // a kernel that is three dependent round trips and a launch.  Same arithmetic as merge_splits: bit-identical.
    int d = threadIdx.x;
    float ms[16], ls[16], os[16];
    merge_load(p, 0, max_splits - 1, stride, min(d, head_size - 1), ms, ls, os);
    const int step = seq_step(step_dev, step_arg, step_stride, b, max_seq_len);
    const int nsplits = (step + chunk - 1) / chunk;
    if (nsplits <= 1) return;  // the split kernel already wrote the final output (or: invalid position)
    if (d < head_size) {
        float M = -INFINITY, L = 0.f, o = 0.f;
        merge_round(ms, ls, os, 0, nsplits, M, L, o);
        for (int s0 = 16; s0 < nsplits; s0 += 16) {
            merge_load(p, s0, nsplits - 1, stride, d, ms, ls, os);
            merge_round(ms, ls, os, s0, nsplits, M, L, o);
        }
        out[attn_out_index(out_x32, b, h * head_size + d, head_num * head_size)] = from_f32<T>(o / (L + 1e-6f));
    }
    for (d += blockDim.x; d < head_size; d += blockDim.x)   // head sizes above the block size
        out[attn_out_index(out_x32, b, h * head_size + d, head_num * head_size)] = from_f32<T>(merge_splits(p, nsplits, stride, d));
}

// Any head size / GQA ratio (e.g. the reference unit test's hs=4): one workgroup per (b, q-head),
// scalar loads, logits in LDS.  step*4 bytes of dynamic LDS.
template <typename T>
__global__ __launch_bounds__(256) void decode_attn_generic_kernel(
    const T *__restrict__ qkv, const T *__restrict__ qkv_bias, T *k_cache, T *v_cache, T *__restrict__ out,
    int head_num, int kv_head_num, int head_size, int max_seq_len, int step_arg,
    const int32_t *__restrict__ step_dev) {
    extern __shared__ float logits[];  // [step]
    __shared__ float red[4];
    const int h = blockIdx.x, b = blockIdx.y;
    const int step = seq_step(step_dev, step_arg, 0, b, max_seq_len);
    if (step == 0) return;   // invalid device-resident position
    const int rep = head_num / kv_head_num, g = h / rep;
    const int qkv_heads = head_num + 2 * kv_head_num;
    const T *row = qkv + static_cast<size_t>(b) * qkv_heads * head_size;
    const int hk = head_num + g, hv = head_num + kv_head_num + g;
    const size_t head_off = (static_cast<size_t>(b) * kv_head_num + g) * max_seq_len * head_size;
    T *kc = k_cache + head_off, *vc = v_cache + head_off;
    const int t_new = step - 1;
    const float scale = rsqrtf(static_cast<float>(head_size));
    auto qv = [&](int d) { return to_f32(row[static_cast<size_t>(h) * head_size + d]) + (qkv_bias ? to_f32(qkv_bias[static_cast<size_t>(h) * head_size + d]) : 0.f); };
    auto knew = [&](int d) { return from_f32<T>(to_f32(row[static_cast<size_t>(hk) * head_size + d]) + (qkv_bias ? to_f32(qkv_bias[static_cast<size_t>(hk) * head_size + d]) : 0.f)); };
    auto vnew = [&](int d) { return from_f32<T>(to_f32(row[static_cast<size_t>(hv) * head_size + d]) + (qkv_bias ? to_f32(qkv_bias[static_cast<size_t>(hv) * head_size + d]) : 0.f)); };
    if (h % rep == 0) {  // one q-head per kv group appends; the others read k/v_new from qkv
        for (int d = threadIdx.x; d < head_size; d += 256) {
            kc[static_cast<size_t>(t_new) * head_size + d] = knew(d);
            vc[static_cast<size_t>(t_new) * head_size + d] = vnew(d);
        }
    }
    float mx = -INFINITY;
    for (int t = threadIdx.x; t < step; t += 256) {
        float acc = 0.f;
        for (int d = 0; d < head_size; ++d) {
            const float kvl = (t == t_new) ? to_f32(knew(d)) : to_f32(kc[static_cast<size_t>(t) * head_size + d]);
            acc = fmaf(qv(d), kvl, acc);
        }
        acc *= scale;
        logits[t] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = block_max<4>(mx, red);
    float sum = 0.f;
    for (int t = threadIdx.x; t < step; t += 256) {
        const float p = expf(logits[t] - mx);
        logits[t] = p;
        sum += p;
    }
    sum = block_sum<4>(sum, red) + 1e-6f;
    __syncthreads();
    for (int d = threadIdx.x; d < head_size; d += 256) {
        float o = 0.f;
        for (int t = 0; t < step; ++t) {
            const float vvl = (t == t_new) ? to_f32(vnew(d)) : to_f32(vc[static_cast<size_t>(t) * head_size + d]);
            o = fmaf(logits[t], vvl, o);
        }
        out[(static_cast<size_t>(b) * head_num + h) * head_size + d] = from_f32<T>(o / sum);
    }
}

struct KvScale {
    float k, v;
};

template <typename T, int HS, int REP, typename KT = T>
static void launch_split(const T *qkv, const T *bias, KT *kc, KT *vc, float *part, T *out, int batch,
                         int head_num, int kv_head_num, int max_seq_len, int step, const int32_t *step_dev,
                         int max_splits_ws, const float2 *rope, int rot_dim, int32_t *tickets, const QkvSlabs &qs,
                         KvScale ks, PagedKv pg, hipStream_t st) {
    const int bound = step_dev ? max_seq_len : step;
    // chunks per workgroup: a function of the batch geometry ONLY (not of the step: the host-step and the device-step form of
    // one call must chunk alike), 1 while the grid needs every chunk as its own workgroup, up to 8 for large batches (batch 32 x 32
    // heads at ctx 2048: one workgroup per (sequence, head), no partials and no merge launch at all)
    int cpw = 1;
    if constexpr (!std::is_same<KT, T>::value) {   // (e4m3 cache only, see the kernel)
        while (cpw < 8 && batch * kv_head_num >= 128 * cpw) cpw *= 2;   // >= 256 workgroups per chunk row are kept
    }
    int CHUNK, splits;
#define LLMIE_ATTN_LAUNCH(NWV_, GL_)                                                                                   \
    do {                                                                                                                \
        CHUNK = cpw * AttnGeom<KT, HS, NWV_, GL_>::CHUNK;                                                               \
        splits = (bound + CHUNK - 1) / CHUNK;                                                                           \
        dim3 grid(splits, kv_head_num, batch);                                                                          \
        decode_attn_split_kernel<T, HS, REP, NWV_, GL_, KT><<<grid, NWV_ * 64, 0, st>>>(                                \
            qkv, bias, kc, vc, part, out, head_num, kv_head_num, max_seq_len, step, step_dev, max_splits_ws, rope, rot_dim, \
            tickets, qs, ks.k, ks.v, pg, cpw);                                                                          \
    } while (0)
    LLMIE_ATTN_LAUNCH(4, 8);   // 4 waves, 8 K + 8 V loads in flight per lane (8 waves x 8 and 4 x 4 measured slower, round 1)
#undef LLMIE_ATTN_LAUNCH
    if (splits > 1 && !tickets) {
        dim3 cgrid(head_num, batch);
        decode_attn_combine_kernel<T><<<cgrid, HS < 64 ? 64 : (HS > 256 ? 256 : HS), 0, st>>>(part, out, head_num, HS, CHUNK,
                                                                                              step, step_dev, max_splits_ws, pg.step_stride,
                                                                                              max_seq_len, pg.out_x32);
    }
}

template <typename T, int HS, typename KT = T>
static bool dispatch_rep(int rep, const T *qkv, const T *bias, KT *kc, KT *vc, float *part, T *out, int batch,
                         int head_num, int kv_head_num, int max_seq_len, int step, const int32_t *step_dev,
                         int max_splits_ws, const float2 *rope, int rot_dim, int32_t *tickets, const QkvSlabs &qs,
                         KvScale ks, PagedKv pg, hipStream_t st) {
    switch (rep) {
        case 1: launch_split<T, HS, 1, KT>(qkv, bias, kc, vc, part, out, batch, head_num, kv_head_num, max_seq_len, step, step_dev, max_splits_ws, rope, rot_dim, tickets, qs, ks, pg, st); return true;
        case 2: launch_split<T, HS, 2, KT>(qkv, bias, kc, vc, part, out, batch, head_num, kv_head_num, max_seq_len, step, step_dev, max_splits_ws, rope, rot_dim, tickets, qs, ks, pg, st); return true;
        case 4: launch_split<T, HS, 4, KT>(qkv, bias, kc, vc, part, out, batch, head_num, kv_head_num, max_seq_len, step, step_dev, max_splits_ws, rope, rot_dim, tickets, qs, ks, pg, st); return true;
        case 8:
            if constexpr (!std::is_same<KT, T>::value) return false;  // fp8 cache: 16 dims per lane x 8 heads does not fit registers
            else launch_split<T, HS, 8, KT>(qkv, bias, kc, vc, part, out, batch, head_num, kv_head_num, max_seq_len, step, step_dev, max_splits_ws, rope, rot_dim, tickets, qs, ks, pg, st); return true;
        default: return false;
    }
}

template <typename T>
static int decoder_mha_impl(const T *qkv, const T *bias, T *k_cache, T *v_cache, T *out, int layer, int batch,
                            int head_num, int kv_head_num, int head_size, int max_seq_len, int step,
                            const int32_t *step_dev, void *workspace, size_t workspace_bytes, const float2 *rope,
                            int rot_dim, int32_t *tickets, const QkvSlabs &qs, hipStream_t st, PagedKv pg = PagedKv{nullptr, 0, 0, 0},
                            int num_pages = 0) {
    const size_t layer_off = pg.table ? static_cast<size_t>(layer) * num_pages * kv_head_num * KV_PAGE * head_size
                                      : static_cast<size_t>(layer) * batch * kv_head_num * max_seq_len * head_size;
    T *kc = k_cache + layer_off, *vc = v_cache + layer_off;
    const int rep = head_num / kv_head_num;
    const int max_splits_ws = (max_seq_len + attn_min_chunk() - 1) / attn_min_chunk();
    const bool aligned = ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(kc) |
                           reinterpret_cast<uintptr_t>(vc)) % 16 == 0) && (!bias || reinterpret_cast<uintptr_t>(bias) % 16 == 0);
    bool done = false;
    if (aligned && batch <= 65535 && kv_head_num <= 65535) {
        const size_t need = llmie_decoder_mha_workspace_bytes(batch, head_num, head_size, max_seq_len);
        float *part = static_cast<float *>(workspace);
        auto ws_ok = [&]() { return workspace && workspace_bytes >= need; };
        if (head_size == 128 && ws_ok())
            done = dispatch_rep<T, 128>(rep, qkv, bias, kc, vc, part, out, batch, head_num, kv_head_num, max_seq_len, step, step_dev, max_splits_ws, rope, rot_dim, tickets, qs, KvScale{1.f, 1.f}, pg, st);
        else if (head_size == 64 && ws_ok())
            done = dispatch_rep<T, 64>(rep, qkv, bias, kc, vc, part, out, batch, head_num, kv_head_num, max_seq_len, step, step_dev, max_splits_ws, rope, rot_dim, tickets, qs, KvScale{1.f, 1.f}, pg, st);
        else if (head_size == 32 && ws_ok())
            done = dispatch_rep<T, 32>(rep, qkv, bias, kc, vc, part, out, batch, head_num, kv_head_num, max_seq_len, step, step_dev, max_splits_ws, rope, rot_dim, tickets, qs, KvScale{1.f, 1.f}, pg, st);
        else if (head_size == 256 && ws_ok())
            done = dispatch_rep<T, 256>(rep, qkv, bias, kc, vc, part, out, batch, head_num, kv_head_num, max_seq_len, step, step_dev, max_splits_ws, rope, rot_dim, tickets, qs, KvScale{1.f, 1.f}, pg, st);
        if (!done && (head_size == 128 || head_size == 64 || head_size == 32 || head_size == 256) && !ws_ok() &&
            (rep == 1 || rep == 2 || rep == 4 || rep == 8)) {
            set_error("decoder_mha: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
            return LLMIE_ERR_WORKSPACE;
        }
    }
    if (!done && pg.table) {
        set_error("decoder_mha: the paged KV cache needs head_size in {32,64,128,256} and head_num/kv_head_num in {1,2,4,8}");
        return LLMIE_ERR_UNSUPPORTED;
    }
    if (!done && qs.slab) {
        set_error("decoder_mha: q/k/v from split-K slabs needs head_size in {32,64,128,256} and head_num/kv_head_num in {1,2,4,8}");
        return LLMIE_ERR_UNSUPPORTED;
    }
    if (!done && rope) {
        set_error("decoder_mha: fused RoPE needs head_size in {32,64,128,256} and head_num/kv_head_num in {1,2,4,8}");
        return LLMIE_ERR_UNSUPPORTED;
    }
    if (!done && (pg.step_stride || pg.out_x32)) {
        set_error("decoder_mha: ragged batches / x32 output need head_size in {32,64,128,256} and head_num/kv_head_num in {1,2,4,8}");
        return LLMIE_ERR_UNSUPPORTED;
    }
    if (!done) {
        const int bound = step_dev ? max_seq_len : step;
        const size_t lds = sizeof(float) * static_cast<size_t>(bound);
        if (lds > 60 * 1024) {
            set_error("decoder_mha: generic path supports at most 15360 tokens (head_size=%d, rep=%d)", head_size, rep);
            return LLMIE_ERR_UNSUPPORTED;
        }
        dim3 grid(head_num, batch);
        decode_attn_generic_kernel<T><<<grid, 256, lds, st>>>(qkv, bias, kc, vc, out, head_num, kv_head_num,
                                                             head_size, max_seq_len, step, step_dev);
    }
    return launch_status("decoder_mha");
}

// fp16 activations over an e4m3 KV cache [L, batch, kvh, max_seq, hs] bytes (stored = e4m3(x / scale)): same kernel with 16
// cache elements per 16-byte load (8 lanes per token row, 256-token chunks); head_size 128 or 64, head ratio 1/2/4
static int decoder_mha_fp8kv(const half_t *qkv, const half_t *bias, uint8_t *k_cache, uint8_t *v_cache, half_t *out, int layer, int batch,
                             int head_num, int kv_head_num, int head_size, int max_seq_len, int step, const int32_t *step_dev,
                             void *workspace, size_t workspace_bytes, const float2 *rope, int rot_dim, const QkvSlabs &qs,
                             KvScale ks, hipStream_t st, PagedKv pg, int num_pages) {
    const size_t layer_off = pg.table ? static_cast<size_t>(layer) * num_pages * kv_head_num * KV_PAGE * head_size
                                      : static_cast<size_t>(layer) * batch * kv_head_num * max_seq_len * head_size;
    fp8kv_t *kc = reinterpret_cast<fp8kv_t *>(k_cache) + layer_off, *vc = reinterpret_cast<fp8kv_t *>(v_cache) + layer_off;
    const int rep = head_num / kv_head_num;
    const int max_splits_ws = (max_seq_len + attn_min_chunk() - 1) / attn_min_chunk();
    const size_t need = llmie_decoder_mha_workspace_bytes(batch, head_num, head_size, max_seq_len);
    if ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(kc) | reinterpret_cast<uintptr_t>(vc) |
         reinterpret_cast<uintptr_t>(bias)) % 16 || !workspace || workspace_bytes < need || batch > 65535 || !(ks.k > 0.f) ||
        !(ks.v > 0.f) || (rep != 1 && rep != 2 && rep != 4) || (head_size != 128 && head_size != 64)) {
        set_error("decoder_mha(fp8 KV): needs head_size 64/128, head ratio 1/2/4, 16-byte aligned buffers, positive scales and "
                  "the llmie_decoder_mha workspace");
        return LLMIE_ERR_UNSUPPORTED;
    }
    float *part = static_cast<float *>(workspace);
    bool done;
    if (head_size == 128)
        done = dispatch_rep<half_t, 128, fp8kv_t>(rep, qkv, bias, kc, vc, part, out, batch, head_num, kv_head_num, max_seq_len, step,
                                                  step_dev, max_splits_ws, rope, rot_dim, nullptr, qs, ks, pg, st);
    else
        done = dispatch_rep<half_t, 64, fp8kv_t>(rep, qkv, bias, kc, vc, part, out, batch, head_num, kv_head_num, max_seq_len, step,
                                                 step_dev, max_splits_ws, rope, rot_dim, nullptr, qs, ks, pg, st);
    (void)done;
    return launch_status("decoder_mha(fp8 KV)");
}

// engine entry: same as llmie_decoder_mha with RoPE (table [max_pos][hs/2] of (cos,sin)) fused in front
int decoder_mha_rope(const void *qkv, const void *qkv_bias, void *k_cache, void *v_cache, void *out, int layer, int batch,
                     int head_num, int kv_head_num, int head_size, int max_seq_len, int step, const int32_t *step_dev,
                     void *workspace, size_t workspace_bytes, const float2 *rope, int rot_dim, int32_t *tickets,
                     llmie_dtype dtype, hipStream_t st, const SplitKSlabs *qkv_slabs, const SlabScale *qkv_scale, int kv_fp8,
                     float k_scale, float v_scale, const int32_t *block_table, int max_pages, int num_pages, int ragged,
                     int out_x32) {
    const PagedKv pg{block_table, max_pages, ragged ? 1 : 0, out_x32 ? 1 : 0};
    if (ragged && !step_dev) {
        set_error("decoder_mha: a ragged batch needs the device array of context lengths");
        return LLMIE_ERR_INVALID_ARG;
    }
    if (out_x32 && (batch > 32 || dtype != LLMIE_F16)) {
        set_error("decoder_mha: the x32 output image holds at most 32 fp16 rows");
        return LLMIE_ERR_UNSUPPORTED;
    }
    if (block_table && (max_pages <= 0 || num_pages <= 0 || static_cast<long long>(max_pages) * KV_PAGE < max_seq_len)) {
        set_error("decoder_mha: paged KV cache needs max_pages * %d >= max_seq_len and num_pages > 0", KV_PAGE);
        return LLMIE_ERR_UNSUPPORTED;
    }
    QkvSlabs qs{nullptr, 0, 0, SlabScale{nullptr, nullptr, nullptr}};
    const SlabScale no_scale{nullptr, nullptr, nullptr};
    const SlabScale &qsc = qkv_scale ? *qkv_scale : no_scale;
    if (qkv_slabs && (reinterpret_cast<uintptr_t>(qsc.wh) % 8 || reinterpret_cast<uintptr_t>(qsc.wf) % 16 ||
                      reinterpret_cast<uintptr_t>(qkv_slabs->slab) % 16 ||
                      (static_cast<size_t>(qkv_slabs->M) * qkv_slabs->N) % 4)) {
        set_error("decoder_mha: q/k/v from split-K slabs needs 16-byte aligned slabs and aligned scale vectors");
        return LLMIE_ERR_UNSUPPORTED;
    }
    if (qkv_slabs) qs = QkvSlabs{qkv_slabs->slab, qkv_slabs->KS, static_cast<size_t>(qkv_slabs->M) * qkv_slabs->N, qsc};
    if (kv_fp8) {
        if (dtype != LLMIE_F16 || tickets) {
            set_error("decoder_mha(fp8 KV): fp16 activations, separate merge kernel only");
            return LLMIE_ERR_UNSUPPORTED;
        }
        return decoder_mha_fp8kv((const half_t *)qkv, (const half_t *)qkv_bias, (uint8_t *)k_cache, (uint8_t *)v_cache, (half_t *)out,
                                 layer, batch, head_num, kv_head_num, head_size, max_seq_len, step, step_dev, workspace,
                                 workspace_bytes, rope, rot_dim, qs, KvScale{k_scale, v_scale}, st, pg, num_pages);
    }
    if (dtype == LLMIE_F32)
        return decoder_mha_impl<float>((const float *)qkv, (const float *)qkv_bias, (float *)k_cache, (float *)v_cache,
                                       (float *)out, layer, batch, head_num, kv_head_num, head_size, max_seq_len, step,
                                       step_dev, workspace, workspace_bytes, rope, rot_dim, tickets, qs, st, pg, num_pages);
    return decoder_mha_impl<half_t>((const half_t *)qkv, (const half_t *)qkv_bias, (half_t *)k_cache, (half_t *)v_cache,
                                    (half_t *)out, layer, batch, head_num, kv_head_num, head_size, max_seq_len, step,
                                    step_dev, workspace, workspace_bytes, rope, rot_dim, tickets, qs, st, pg, num_pages);
}

}  // namespace llmie

using namespace llmie;

extern "C" size_t llmie_decoder_mha_workspace_bytes(int batch, int head_num, int head_size, int max_seq_len) {
    if (batch <= 0 || head_num <= 0 || head_size <= 0 || max_seq_len <= 0) return 0;
    const size_t splits = (static_cast<size_t>(max_seq_len) + attn_min_chunk() - 1) / attn_min_chunk();
    return static_cast<size_t>(batch) * head_num * splits * (head_size + 2) * sizeof(float);
}

extern "C" int llmie_decoder_mha(const void *qkv, const void *qkv_bias, void *k_cache, void *v_cache, void *out,
                                 int layer, int batch, int head_num, int kv_head_num, int head_size,
                                 int max_seq_len, int step, const int32_t *step_dev, void *workspace,
                                 size_t workspace_bytes, llmie_dtype dtype, llmie_stream stream) {
    LLMIE_REQUIRE(qkv && k_cache && v_cache && out, "decoder_mha: NULL pointer");
    LLMIE_REQUIRE(layer >= 0 && batch > 0 && head_num > 0 && kv_head_num > 0 && head_size > 0 && max_seq_len > 0,
                  "decoder_mha: bad shape");
    LLMIE_REQUIRE(head_num % kv_head_num == 0, "decoder_mha: kv_head_num must divide head_num");
    LLMIE_REQUIRE(step_dev || (step >= 1 && step <= max_seq_len), "decoder_mha: step %d outside [1, max_seq_len=%d]",
                  step, max_seq_len);
    if (dtype == LLMIE_F32)
        return decoder_mha_impl<float>((const float *)qkv, (const float *)qkv_bias, (float *)k_cache, (float *)v_cache,
                                       (float *)out, layer, batch, head_num, kv_head_num, head_size, max_seq_len, step,
                                       step_dev, workspace, workspace_bytes, nullptr, 0, nullptr, QkvSlabs{nullptr, 0, 0, SlabScale{nullptr, nullptr, nullptr}}, as_stream(stream));
    if (dtype == LLMIE_F16)
        return decoder_mha_impl<half_t>((const half_t *)qkv, (const half_t *)qkv_bias, (half_t *)k_cache,
                                        (half_t *)v_cache, (half_t *)out, layer, batch, head_num, kv_head_num, head_size,
                                        max_seq_len, step, step_dev, workspace, workspace_bytes, nullptr, 0, nullptr, QkvSlabs{nullptr, 0, 0, SlabScale{nullptr, nullptr, nullptr}}, as_stream(stream));
    LLMIE_UNSUPPORTED("decoder_mha: dtype %d", (int)dtype);
}

extern "C" int llmie_decoder_mha_rope(const void *qkv, const void *qkv_bias, void *k_cache, void *v_cache, void *out,
                                      int layer, int batch, int head_num, int kv_head_num, int head_size,
                                      int max_seq_len, int step, const int32_t *step_dev, void *workspace,
                                      size_t workspace_bytes, const void *rope_table, int rotary_dim, int32_t *tickets,
                                      llmie_dtype dtype, llmie_stream stream) {
    LLMIE_REQUIRE(qkv && k_cache && v_cache && out, "decoder_mha_rope: NULL pointer");
    LLMIE_REQUIRE(layer >= 0 && batch > 0 && head_num > 0 && kv_head_num > 0 && head_size > 0 && max_seq_len > 0,
                  "decoder_mha_rope: bad shape");
    LLMIE_REQUIRE(head_num % kv_head_num == 0, "decoder_mha_rope: kv_head_num must divide head_num");
    LLMIE_REQUIRE(step_dev || (step >= 1 && step <= max_seq_len), "decoder_mha_rope: step %d outside [1, max_seq_len=%d]",
                  step, max_seq_len);
    LLMIE_REQUIRE(!rope_table || (rotary_dim > 0 && rotary_dim % 2 == 0), "decoder_mha_rope: bad rotary_dim");
    if (dtype != LLMIE_F32 && dtype != LLMIE_F16) LLMIE_UNSUPPORTED("decoder_mha_rope: dtype %d", (int)dtype);
    const int rep = head_num / kv_head_num;
    const bool hs_ok = head_size == 32 || head_size == 64 || head_size == 128 || head_size == 256;
    const bool rep_ok = rep == 1 || rep == 2 || rep == 4 || rep == 8;
    if ((rope_table || tickets) && !(hs_ok && rep_ok))
        LLMIE_UNSUPPORTED("decoder_mha_rope: fused RoPE / in-launch merge need head_size in {32,64,128,256} and "
                          "head_num/kv_head_num in {1,2,4,8}");
    return decoder_mha_rope(qkv, qkv_bias, k_cache, v_cache, out, layer, batch, head_num, kv_head_num, head_size,
                            max_seq_len, step, step_dev, workspace, workspace_bytes,
                            static_cast<const float2 *>(rope_table), rotary_dim, tickets, dtype, as_stream(stream));
}

// Ragged batch: RoPE position, append slot and attention span per sequence (ctx_len_dev[b] includes this step's token);
// block_table != NULL: paged cache pools.  Same kernels as llmie_decoder_mha_rope.
extern "C" int llmie_decoder_mha_ragged(const void *qkv, const void *qkv_bias, void *k_cache, void *v_cache, void *out, int layer,
                                        int batch, int head_num, int kv_head_num, int head_size, int max_seq_len,
                                        const int32_t *ctx_len_dev, void *workspace, size_t workspace_bytes, const void *rope_table,
                                        int rotary_dim, const int32_t *block_table, int max_pages, int num_pages,
                                        llmie_dtype dtype, llmie_stream stream) {
    LLMIE_REQUIRE(qkv && k_cache && v_cache && out && ctx_len_dev, "decoder_mha_ragged: NULL pointer");
    LLMIE_REQUIRE(layer >= 0 && batch > 0 && head_num > 0 && kv_head_num > 0 && head_size > 0 && max_seq_len > 0,
                  "decoder_mha_ragged: bad shape");
    LLMIE_REQUIRE(head_num % kv_head_num == 0, "decoder_mha_ragged: kv_head_num must divide head_num");
    LLMIE_REQUIRE(!rope_table || (rotary_dim > 0 && rotary_dim % 2 == 0), "decoder_mha_ragged: bad rotary_dim");
    if (dtype != LLMIE_F32 && dtype != LLMIE_F16) LLMIE_UNSUPPORTED("decoder_mha_ragged: dtype %d", (int)dtype);
    return decoder_mha_rope(qkv, qkv_bias, k_cache, v_cache, out, layer, batch, head_num, kv_head_num, head_size, max_seq_len, -1,
                            ctx_len_dev, workspace, workspace_bytes, static_cast<const float2 *>(rope_table), rotary_dim, nullptr,
                            dtype, as_stream(stream), nullptr, nullptr, 0, 1.f, 1.f, block_table, max_pages, num_pages, 1, 0);
}

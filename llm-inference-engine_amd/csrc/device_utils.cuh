// Device helpers: 64-lane wave reductions, 16-byte vector types, fp16<->fp32 packs.
// Written for gfx950 (wave64, 4 SIMD-32 per CU); no 32-lane assumptions anywhere.
#pragma once
#include "llmie_common.h"

namespace llmie {

using half_t = _Float16;
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef half_t half2_t __attribute__((ext_vector_type(2)));
typedef half_t half4_t __attribute__((ext_vector_type(4)));
typedef half_t half8_t __attribute__((ext_vector_type(8)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef int int4_t __attribute__((ext_vector_type(4)));

// element traits: 16-byte vector of T
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    using type = float4_t;
    static constexpr int n = 4;
};
template <> struct Vec16<half_t> {
    using type = half8_t;
    static constexpr int n = 8;
};

template <typename T> __device__ __forceinline__ float to_f32(T v) { return static_cast<float>(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return static_cast<T>(v); }

// ---- wave64 reductions on DPP (VALU cross-lane moves, no LDS traffic; __shfl_xor lowers to ds_bpermute) ----
// gfx9 DPP controls: quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140,
// row_bcast15 = 0x142, row_bcast31 = 0x143.  All 64 lanes must be active at the call.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_or(float v, float masked) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, masked), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_or<0xB1, 0xF>(v, 0.f);
    v += dpp_or<0x4E, 0xF>(v, 0.f);
    v += dpp_or<0x141, 0xF>(v, 0.f);
    v += dpp_or<0x140, 0xF>(v, 0.f);   // every lane: sum of its row of 16
    v += dpp_or<0x142, 0xA>(v, 0.f);   // rows 1,3 += row 0,2
    v += dpp_or<0x143, 0xC>(v, 0.f);   // rows 2,3 += (row 0 + row 1): lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_or<0xB1, 0xF>(v, v));
    v = fmaxf(v, dpp_or<0x4E, 0xF>(v, v));
    v = fmaxf(v, dpp_or<0x141, 0xF>(v, v));
    v = fmaxf(v, dpp_or<0x140, 0xF>(v, v));
    v = fmaxf(v, dpp_or<0x142, 0xA>(v, v));
    v = fmaxf(v, dpp_or<0x143, 0xC>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// ---- exact xor-lane exchanges without the LDS crossbar (round 3).  __shfl_xor lowers to ds_bpermute_b32: an LDS-path round trip
// of ~100+ cycles behind an lgkmcnt wait that also drains every other LDS / scalar-memory access of the wave; the decode attention
// kernel held 68 of them, several of them in dependent chains.  All 64 lanes must be active at the call.
//   xor 1, 2: DPP quad_perm;  xor 4: DPP row_shl:4 into banks 0 / 2 + row_shr:4 into banks 1 / 3;  xor 8: DPP row_ror:8;
//   xor 16, 32: the gfx950 row swaps v_permlane16_swap / v_permlane32_swap (both registers of the swap hold v: afterwards one
//   holds the lane's own-or-partner row, the other the other one -- enough for commutative reductions, see lane_xor_sum / _max).
template <int O> __device__ __forceinline__ float lane_xor_lt16(float v) {
    static_assert(O == 1 || O == 2 || O == 4 || O == 8, "DPP forms: xor 1, 2, 4, 8");
    const int x = __builtin_bit_cast(int, v);
    int r;
    if constexpr (O == 1) r = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);
    else if constexpr (O == 2) r = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false);
    else if constexpr (O == 8) r = __builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, false);   // row_ror:8 = lane ^ 8 inside a row of 16
    else {
        r = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);   // row_shl:4 -> lanes 0-3, 8-11 of every row take lane + 4
        r = __builtin_amdgcn_update_dpp(r, x, 0x114, 0xF, 0xA, false);   // row_shr:4 -> lanes 4-7, 12-15 take lane - 4
    }
    return __builtin_bit_cast(float, r);
}
// hipcc (ROCm 7.2) mis-lowers the SECOND result of __builtin_amdgcn_permlane{16,32}_swap: both elements of the returned pair come
// out of the first register -- `max(s[0], s[1])` compiled to `s[0]`, "the value of the lane's even row", no maximum at all (found
// with tools/micro/lane_xor_check.hip; the flash prefill kernel had run for a few hours on a row "maximum" that was merely the
// same for the four lanes of a row -- mathematically still a valid softmax stabiliser, but no bound on the numerators).  An empty
// asm on both extracted values makes the register allocator hand out the second register; the check tool and
// tests/test_abi_cpu.py (disassembly: every row swap is followed by an instruction that reads BOTH of its registers) pin it.
template <int O> __device__ __forceinline__ void lane_row_swap(float v, float &a, float &b) {
    static_assert(O == 16 || O == 32, "row swaps: distance 16 or 32");
    unsigned x = __builtin_bit_cast(unsigned, v), y = x;
    asm volatile("" : "+v"(y));
    const auto s = O == 16 ? __builtin_amdgcn_permlane16_swap(x, y, false, false) : __builtin_amdgcn_permlane32_swap(x, y, false, false);
    unsigned ua = s[0], ub = s[1];
    asm volatile("" : "+v"(ua), "+v"(ub));
    a = __builtin_bit_cast(float, ua);
    b = __builtin_bit_cast(float, ub);
}
// v[lane] + v[lane ^ O] / max(v[lane], v[lane ^ O]) in every lane, O a power of two < 64: the operands of the xor butterfly, so
// the results are bit-identical to the __shfl_xor forms (fp add and max are commutative)
template <int O> __device__ __forceinline__ float lane_xor_sum(float v) {
    if constexpr (O < 16) return v + lane_xor_lt16<O>(v);
    else {
        float a, b;
        lane_row_swap<O>(v, a, b);
        return a + b;
    }
}
template <int O> __device__ __forceinline__ float lane_xor_max(float v) {
    if constexpr (O < 16) return fmaxf(v, lane_xor_lt16<O>(v));
    else {
        float a, b;
        lane_row_swap<O>(v, a, b);
        return fmaxf(a, b);
    }
}
// the partner's value itself, any distance (one select more than the combined forms for 16 / 32): for arg-max style butterflies
template <int O> __device__ __forceinline__ float lane_xor(float v) {
    if constexpr (O < 16) return lane_xor_lt16<O>(v);
    else {
        float a, b;   // a = the even row's (lower half's) value of the lane's pair, b = the odd row's (upper half's)
        lane_row_swap<O>(v, a, b);
        const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        return (lane & O) ? a : b;
    }
}
template <int O> __device__ __forceinline__ int lane_xor(int v) { return __builtin_bit_cast(int, lane_xor<O>(__builtin_bit_cast(float, v))); }
// the same for a distance known after unrolling (o in {1, 2, 4, 8, 16, 32})
__device__ __forceinline__ float lane_xor_sum_o(float v, int o) {
    switch (o) {
        case 1: return lane_xor_sum<1>(v);
        case 2: return lane_xor_sum<2>(v);
        case 4: return lane_xor_sum<4>(v);
        case 8: return lane_xor_sum<8>(v);
        case 16: return lane_xor_sum<16>(v);
        default: return lane_xor_sum<32>(v);
    }
}
__device__ __forceinline__ float lane_xor_max_o(float v, int o) {
    switch (o) {
        case 1: return lane_xor_max<1>(v);
        case 2: return lane_xor_max<2>(v);
        case 4: return lane_xor_max<4>(v);
        case 8: return lane_xor_max<8>(v);
        case 16: return lane_xor_max<16>(v);
        default: return lane_xor_max<32>(v);
    }
}
template <typename V> __device__ __forceinline__ V lane_xor_o(V v, int o) {
    switch (o) {
        case 1: return lane_xor<1>(v);
        case 2: return lane_xor<2>(v);
        case 4: return lane_xor<4>(v);
        case 8: return lane_xor<8>(v);
        case 16: return lane_xor<16>(v);
        default: return lane_xor<32>(v);
    }
}
// reduce within groups of `width` consecutive lanes (width power of two <= 64): the xor butterfly, largest distance first
template <int WIDTH> __device__ __forceinline__ float group_sum(float v) {
    if constexpr (WIDTH >= 64) v = lane_xor_sum<32>(v);
    if constexpr (WIDTH >= 32) v = lane_xor_sum<16>(v);
    if constexpr (WIDTH >= 16) v = lane_xor_sum<8>(v);
    if constexpr (WIDTH >= 8) v = lane_xor_sum<4>(v);
    if constexpr (WIDTH >= 4) v = lane_xor_sum<2>(v);
    if constexpr (WIDTH >= 2) v = lane_xor_sum<1>(v);
    return v;
}

// Block-wide sum for blocks of NW waves; `lds` needs NW floats. All threads get the result.
template <int NW> __device__ __forceinline__ float block_sum(float v, float *lds) {
    v = wave_sum(v);
    if constexpr (NW == 1) return v;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();  // protect lds reuse across consecutive calls
    if (lane == 0) lds[wid] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) r += lds[i];
    return r;
}
template <int NW> __device__ __forceinline__ float block_max(float v, float *lds) {
    v = wave_max(v);
    if constexpr (NW == 1) return v;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) lds[wid] = v;
    __syncthreads();
    float r = lds[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) r = fmaxf(r, lds[i]);
    return r;
}

// streaming (read-once) 16-byte load: non-temporal so weight streams do not evict reused lines
template <typename V> __device__ __forceinline__ V load_nt(const V *p) {
    return __builtin_nontemporal_load(p);
}

// dot of 8 halves with fp32 accumulate on v_dot2_f32_f16
__device__ __forceinline__ float dot8(half8_t a, half8_t b, float acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        half2_t x = {a[2 * i], a[2 * i + 1]};
        half2_t y = {b[2 * i], b[2 * i + 1]};
        acc = __builtin_amdgcn_fdot2(x, y, acc, false);
    }
    return acc;
}

// 4 floats -> 4 e4m3fn bytes (OCP fp8, the gfx950 conversion), saturating
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

// scales a split-K slab consumer applies to the reduced value of output (m, n):
//   fp16 weights: none;  int8: wh[n] (fp16 per output channel);  fp8: wf[n] * xs[m] (fp32 weight-row x token scales)
struct SlabScale {
    const half_t *wh;
    const float *wf;
    const float *xs;
    __device__ __forceinline__ float apply(float v, int m, int n) const {
        if (wh) return v * static_cast<float>(wh[n]);
        if (wf) return v * (wf[n] * xs[m]);
        return v;
    }
};

// operands of the QKV projection's RoPE + KV-append epilogue (gemm256.cuh g256_store_qkv_rope; filled by the engine's prefill)
struct QkvRopeArgs {           // layer-invariant; lives in device memory (prefill_token_table writes it once per prefill call)
    void *k_cache, *v_cache;   // whole caches (all layers)
    const int32_t *tok_b;      // [T] sequence of packed token t
    const int32_t *tok_tpos;   // [T] cache position history + position of packed token t (< 0 or >= max_seq_len: not written)
    const float2 *rope;        // [max_seq_len][64] (cos, sin)
    const int32_t *table;      // paged cache: [batch, max_pages] or null
    size_t layer_stride;       // elements of one layer's cache slab (dense: batch * kvh * max_seq * 128; paged: num_pages * kvh * 128 * 128)
    int head_num, kv_head_num, max_seq_len, rotary_dim, max_pages, kv8;
    float k_inv_scale, v_inv_scale;
};

// ---- int8 weight tiles in LDS (gemm8p.cuh WQ form, gemm_mid.cuh int8 form): 64-byte rows, 16-byte slot c of row r holds source chunk
// c ^ ((r >> 2) & 3); lane (r, q) multiplies k = 32 s + 8 q .. + 7 in k-step s = bytes [32 s + 8 q, + 8) of its row, one ds_read_b64;
// de-quantisation in registers: byte ^ 0x80 under the fp16 exponent 0x64 is 1152 + w, a packed subtract of 1152 leaves w exactly
__device__ __forceinline__ unsigned g8_q8_slot(int row, int chunk) { return static_cast<unsigned>((chunk ^ (row >> 2)) & 3); }
// byte offset, inside its 64-byte row, of the 8 weights lane (r, q) multiplies in k-step s
__device__ __forceinline__ unsigned g8_q8_piece(int r, int q, int s) { return (g8_q8_slot(r, 2 * s + (q >> 1)) << 4) + 8u * (q & 1); }
__device__ __forceinline__ half8_t g8_dequant8(const uint2 w) {
    const half2_t off = {static_cast<half_t>(1152.f), static_cast<half_t>(1152.f)};
    const unsigned v0 = w.x ^ 0x80808080u, v1 = w.y ^ 0x80808080u;
    const half2_t h0 = __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(0x64646464u, v0, 0x04010400u)) - off;   // {0x64, b1, 0x64, b0}
    const half2_t h1 = __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(0x64646464u, v0, 0x04030402u)) - off;   // {0x64, b3, 0x64, b2}
    const half2_t h2 = __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(0x64646464u, v1, 0x04010400u)) - off;
    const half2_t h3 = __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(0x64646464u, v1, 0x04030402u)) - off;
    return half8_t{h0[0], h0[1], h1[0], h1[1], h2[0], h2[1], h3[0], h3[1]};
}

}  // namespace llmie

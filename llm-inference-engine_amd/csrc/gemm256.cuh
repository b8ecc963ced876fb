// 256 x 256 x 64 MFMA GEMM for prefill-sized projections (gfx950):  C[M,N] = X[M,K] . W[N,K]^T (+bias)(+residual), fp16 in,
// fp32 accumulate -- the linear of launchLinearGemm (linear.cu:10-87) at M >= a few hundred tokens, where the op is
// MFMA-bound instead of weight-bandwidth-bound.
//
// Why a second tiled kernel: the 128 x 128 kernel in gemm_kernels.cuh stages global -> VGPR -> LDS (ds_write_b128 costs 13
// LDS-path cycles per wave instruction) and gives each wave a 64 x 64 tile, i.e. 8 fragment reads per 16 MFMAs; its
// LDS traffic is longer than its MFMA time and it tops out at ~0.9 PFLOP/s (35% of the dense fp16 peak).  Here
//   * both operand tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write), double buffered:
//     the DMA of k-tile t+1 runs under the MFMAs of k-tile t, one barrier per k-tile;
//   * 8 waves (2 x 4) per workgroup, each wave owns 128 x 64 of C = 8 x 4 MFMA tiles (128 accumulator registers) and
//     reads 12 fragments per 32 MFMAs;
//   * an LDS-DMA wave instruction writes 1 KiB linearly (wave-uniform base + lane * 16 B), so the XOR swizzle that makes the
//     fragment reads conflict-free is applied on the SOURCE address: LDS slot (row, s) receives chunk s ^ (row & 7).
// LDS image per stage: X half 0 | X half 1 | W half 0 | W half 1, each 128 rows x 128 B (64 k) = 16 KiB; 2 stages = 128 KiB.
#pragma once
#include "device_utils.cuh"

namespace llmie {

// FP8 form: X and W are e4m3 bytes (X quantised per token, W per output row), a k-tile is 128 k = the same 128-byte rows,
// one v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales, E8M0 127) per output tile and k-tile -- twice the fp16 MFMA
// rate -- and the epilogue applies xscale[m] * wscale[n].
typedef int intx8 __attribute__((ext_vector_type(8)));

// Epilogue shared by the 256-row kernels (this file and gemm8p.cuh): acc[i][j] of wave (wr, wc): lane (r, q) holds
// C[m0 + wr*128 + i*16 + r][n0 + wcol + j*16 + 4q + e]; SwiGLU: acc[i][jj] = gate, acc[i][2 + jj] = up of column n0 + wcol + 16 jj + 4q + e.
// WQ = 8 (gemm8p.cuh, int8 weights): `wscale` points at the fp16 per-row scales of W, applied to the fp32 sums before bias / residual.
template <bool FP8, bool HAS_EPI, int WN, bool SWIGLU, int WQ = 0>
__device__ __forceinline__ void g256_store(floatx4 (&acc)[8][WN], half_t *C, int M, int N, size_t ldc, int m0, int n0, int wr, int wcol,
                                           int r, int q, const half_t *__restrict__ bias, const half_t *residual,
                                           const float *__restrict__ xscale, const float *__restrict__ wscale) {
    const int half_n = N >> 1;
    const half_t *hscale = reinterpret_cast<const half_t *>(wscale);
    (void)hscale;
    if constexpr (SWIGLU) {
        // acc[i][jj] = gate, acc[i][WN / 2 + jj] = up of C columns n0 + wcol + 16 jj + 4q + e
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + wr * 128 + i * 16 + r;
            if (m >= M) continue;
#pragma unroll
            for (int jj = 0; jj < WN / 2; ++jj) {
                const int n = n0 + wcol + jj * 16 + 4 * q;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float gt = acc[i][jj][e], up = acc[i][WN / 2 + jj][e];
                    if constexpr (FP8) {
                        const float xsm = xscale[m];
                        gt *= wscale[min(n + e, half_n - 1)] * xsm;
                        up *= wscale[half_n + min(n + e, half_n - 1)] * xsm;
                    }
                    if constexpr (WQ != 0) {
                        gt *= to_f32(hscale[min(n + e, half_n - 1)]);
                        up *= to_f32(hscale[half_n + min(n + e, half_n - 1)]);
                    }
                    // the unfused sequence rounds gate and up to fp16 before SiluAndMul: keep the same roundings
                    gt = to_f32(from_f32<half_t>(gt));
                    up = to_f32(from_f32<half_t>(up));
                    o[e] = (gt / (1.0f + expf(-gt))) * up;
                }
                if (n + 3 < half_n && (half_n & 3) == 0) {
                    const half4_t o4 = {from_f32<half_t>(o[0]), from_f32<half_t>(o[1]), from_f32<half_t>(o[2]), from_f32<half_t>(o[3])};
                    *reinterpret_cast<half4_t *>(C + static_cast<size_t>(m) * half_n + n) = o4;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < half_n) C[static_cast<size_t>(m) * half_n + n + e] = from_f32<half_t>(o[e]);
                }
            }
        }
        return;
    }
    // acc[i][j]: lane holds C[m0 + wr*128 + i*16 + r][n0 + wcol + j*16 + 4q + e]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wr * 128 + i * 16 + r;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = n0 + wcol + j * 16 + 4 * q;
            if (n + 3 < N && (N & 3) == 0 && (ldc & 3) == 0) {
                floatx4 v = acc[i][j];
                if constexpr (FP8) {
                    const floatx4 ws = *reinterpret_cast<const floatx4 *>(wscale + n);
                    const float xsm = xscale[m];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= ws[e] * xsm;
                }
                if constexpr (WQ != 0) {
                    const half4_t ws = *reinterpret_cast<const half4_t *>(hscale + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= to_f32(ws[e]);
                }
                if (HAS_EPI) {
                    if (bias) {
                        const half4_t b4 = *reinterpret_cast<const half4_t *>(bias + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += to_f32(b4[e]);
                    }
                    if (residual) {
                        const half4_t r4 = *reinterpret_cast<const half4_t *>(residual + static_cast<size_t>(m) * ldc + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += to_f32(r4[e]);
                    }
                }
                const half4_t o = {from_f32<half_t>(v[0]), from_f32<half_t>(v[1]), from_f32<half_t>(v[2]), from_f32<half_t>(v[3])};
                *reinterpret_cast<half4_t *>(C + static_cast<size_t>(m) * ldc + n) = o;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < N) {
                        float v = acc[i][j][e];
                        if constexpr (FP8) v *= wscale[n + e] * xscale[m];
                        if constexpr (WQ != 0) v *= to_f32(hscale[n + e]);
                        if (HAS_EPI) {
                            if (bias) v += to_f32(bias[n + e]);
                            if (residual) v += to_f32(residual[static_cast<size_t>(m) * ldc + n + e]);
                        }
                        C[static_cast<size_t>(m) * ldc + n + e] = from_f32<half_t>(v);
                    }
            }
        }
    }
}

// ---- QKV projection with RoPE + KV-cache append in the epilogue (round 3; context_attention.cpp:158-205: the QKV GEMM,
// launchFusedQKVAddBiasAndTransposeAndRope and launchConcatKVCache of the prefill as ONE launch) ----
// The eight-phase kernels (gemm8p.cuh, ROPE = true) give every wave the 64 (256-wide tile) or 32 (128-wide tile) output columns
// {d, d + 64} x 32 / 16 of ONE head -- the weight rows of a tile are fetched in a permuted order, qkv_rope_col() -- so both halves
// of every rotate-half pair sit in the accumulators of one lane: acc[i][j] and acc[i][j + WN / 2].  The epilogue then does what
// prefill_rope_append_kernel (prefill.hip) does to the fp16 QKV row, on the fp16-ROUNDED accumulator (bit-identical results):
// + bias, rotate q and k heads by the token's (cos, sin) row, store q into the packed QKV buffer, k and v straight into the cache
// slot history + position (dense slab or 128-token pages; fp16 or e4m3).  K / V never travel through the QKV buffer.
// (struct QkvRopeArgs: device_utils.cuh)
// tile column c (0 .. 64 WN - 1, in units of the workgroup tile) -> column inside its 128-wide head
template <int WN> __device__ __forceinline__ int qkv_rope_col(int c) {
    if constexpr (WN == 4) return (c & ~0x7f) | ((c & 0x20) << 1) | ((c & 0x40) >> 1) | (c & 0x1f);   // 256-wide: swap bits 5 and 6
    else return ((c & 0x10) << 2) | ((c & 0x60) >> 1) | (c & 0xf);                                     // 128-wide: [wc:2][jj][4] -> [jj][wc:2][4]
}
// acc[i][j]: lane (r, q) holds row m0 + wr*128 + i*16 + r, PERMUTED tile column wcol + j*16 + 4q + e; n0 = first column of the
// tile in the whole [T, (nh + 2 kvh) * 128] output (a multiple of 128)
template <bool FP8, int WN, int WQ>
__device__ __forceinline__ void g256_store_qkv_rope(floatx4 (&acc)[8][WN], half_t *qkv, int M, size_t ldc, int m0, int n0, int wr, int wcol,
                                                    int r, int q, const float *__restrict__ xscale, const float *__restrict__ wscale,
                                                    const half_t *__restrict__ bias, const QkvRopeArgs *__restrict__ rap, int layer) {
    const QkvRopeArgs ra = *rap;   // (wave-uniform address: scalar loads, issued here -- behind the main loop)
    // the lane's (r, q) re-derived from the lane id the hardware counts (mbcnt) instead of the values the main loop was built on:
    // nothing of this epilogue's address arithmetic can be computed in front of the loop and kept in registers across it (the
    // e4m3 256 x 256 form sits at 256 VGPRs and spilled two such values around its loop)
    {
        const int lane_now = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        r = lane_now & 15;
        q = lane_now >> 4;
    }
    // pointers that came out of the struct are global memory (the compiler would issue flat accesses, which also count on lgkmcnt)
    typedef __attribute__((address_space(1))) const int32_t *gi32_t;
    typedef __attribute__((address_space(1))) const floatx4 *gf4_t;
    const gi32_t tok_tpos = (gi32_t)ra.tok_tpos, tok_b = (gi32_t)ra.tok_b, table = (gi32_t)ra.table;
    const size_t layer_off = static_cast<size_t>(layer) * ra.layer_stride;
    constexpr int HS = 128, HP = WN / 2;   // pairs (j, j + HP)
    const half_t *hscale = reinterpret_cast<const half_t *>(wscale);
    (void)hscale;
    // the head a wave's columns belong to is wave-uniform: (n0 + permuted wcol) / 128 for every j (the 16 j + 4 q + e part stays below 64)
    const int head = __builtin_amdgcn_readfirstlane((n0 + qkv_rope_col<WN>(wcol)) >> 7);
    const bool is_q = head < ra.head_num, is_k = !is_q && head < ra.head_num + ra.kv_head_num, rotate = is_q || is_k;
    const int g = is_k ? head - ra.head_num : head - ra.head_num - ra.kv_head_num;
    const int half_rot = rotate ? (ra.rotary_dim >> 1) : 0;
    // this lane's 8 rows: cache position and sequence, all loads in flight at once
    int tposv[8], bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = min(m0 + wr * 128 + i * 16 + r, M - 1);
        tposv[i] = tok_tpos[m];
        bv[i] = is_q ? 0 : tok_b[m];
    }
    // per column pair j: column-in-head d, scales and bias (the same for every row)
    int dj[HP];
    half4_t blo[HP], bhi[HP];
    floatx4 sl[HP], sh[HP];
#pragma unroll
    for (int j = 0; j < HP; ++j) {
        const int cn = n0 + qkv_rope_col<WN>(wcol + j * 16 + 4 * q);   // output column of e = 0, first half of the pair: head * 128 + d, d < 64
        dj[j] = cn & 127;
        blo[j] = half4_t{0, 0, 0, 0};
        bhi[j] = blo[j];
        if (bias) {
            blo[j] = *reinterpret_cast<const half4_t *>(bias + cn);
            bhi[j] = *reinterpret_cast<const half4_t *>(bias + cn + 64);
        }
        if constexpr (FP8) {
            sl[j] = *reinterpret_cast<const floatx4 *>(wscale + cn);
            sh[j] = *reinterpret_cast<const floatx4 *>(wscale + cn + 64);
        }
        if constexpr (WQ != 0) {
            const half4_t wl = *reinterpret_cast<const half4_t *>(hscale + cn), wh = *reinterpret_cast<const half4_t *>(hscale + cn + 64);
            sl[j] = floatx4{to_f32(wl[0]), to_f32(wl[1]), to_f32(wl[2]), to_f32(wl[3])};
            sh[j] = floatx4{to_f32(wh[0]), to_f32(wh[1]), to_f32(wh[2]), to_f32(wh[3])};
        }
    }
    // (cos, sin) of the lane's 4 HP rotation pairs of a row: two 16-byte loads per j.  Row i + 1's are requested before row i is
    // rotated and stored (straight-line code: a row that must not be written -- past M, or a cache position outside the slab -- loads
    // from a clamped position and skips its stores), so the L2 round trip of one row hides under the arithmetic of another; v heads
    // load nothing
    floatx4 cs[2][HP][2];
    auto load_cs = [&](int i, floatx4 (&c)[HP][2]) {
        const int tp = min(max(tposv[i], 0), ra.max_seq_len - 1);
#pragma unroll
        for (int j = 0; j < HP; ++j) {
            const gf4_t rp = (gf4_t)(ra.rope + static_cast<size_t>(tp) * (HS / 2) + dj[j]);
            c[j][0] = rp[0];   // cos(d), sin(d), cos(d + 1), sin(d + 1)
            c[j][1] = rp[1];
        }
    };
#pragma unroll
    for (int j = 0; j < HP; ++j) cs[0][j][0] = cs[0][j][1] = cs[1][j][0] = cs[1][j][1] = floatx4{1.f, 0.f, 1.f, 0.f};
    if (rotate) load_cs(0, cs[0]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wr * 128 + i * 16 + r, tpos = tposv[i];
        const bool valid = m < M && tpos >= 0 && tpos < ra.max_seq_len;   // (prefill_rope_append_kernel: never write outside the slab)
        if (rotate && i + 1 < 8) load_cs(i + 1, cs[(i + 1) & 1]);
        const float xsm = FP8 ? xscale[min(m, M - 1)] : 1.f;
        const int tpc = min(max(tpos, 0), ra.max_seq_len - 1);
        size_t row_off = 0;   // element offset of this token's row in its kv head's cache slab
        if (!is_q)
            row_off = table ? layer_off + ((static_cast<size_t>(table[static_cast<size_t>(bv[i]) * ra.max_pages + tpc / 128]) * ra.kv_head_num + g) * 128 + tpc % 128) * HS
                            : layer_off + ((static_cast<size_t>(bv[i]) * ra.kv_head_num + g) * ra.max_seq_len + tpc) * HS;
#pragma unroll
        for (int j = 0; j < HP; ++j) {
            floatx4 lo = acc[i][j], hi = acc[i][j + HP];
            const floatx4 cs0 = cs[i & 1][j][0], cs1 = cs[i & 1][j][1];
            const int d = dj[j];
            if constexpr (FP8) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] *= sl[j][e] * xsm;
                    hi[e] *= sh[j][e] * xsm;
                }
            }
            if constexpr (WQ != 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] *= sl[j][e];
                    hi[e] *= sh[j][e];
                    // keep "fp32 product, then one conversion" (what g256_store's packed multiplies + v_cvt_pk_f16_f32 compute): left
                    // to itself the compiler folds a low-half fp16 scale, the product and the conversion into v_fma_mixlo_f16, which
                    // rounds the exact product ONCE -- 1 ulp away from the two-launch sequence in ~1e-4 of the elements
                    asm volatile("" : "+v"(lo[e]), "+v"(hi[e]));
                }
            }
            half4_t olo, ohi;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // the projection's output as the unfused sequence stores it (fp16), then prefill_rope_append_kernel's arithmetic
                const float x0 = to_f32(from_f32<half_t>(lo[e])) + to_f32(blo[j][e]), x1 = to_f32(from_f32<half_t>(hi[e])) + to_f32(bhi[j][e]);
                const float cv = e < 2 ? cs0[2 * e] : cs1[2 * e - 4], sv = e < 2 ? cs0[2 * e + 1] : cs1[2 * e - 3];
                const float r0 = x0 * cv - x1 * sv, r1 = x1 * cv + x0 * sv;
                const bool rot = d + e < half_rot;
                olo[e] = from_f32<half_t>(rot ? r0 : x0);
                ohi[e] = from_f32<half_t>(rot ? r1 : x1);
            }
            if (!valid) {
            } else if (is_q) {
                half_t *dst = qkv + static_cast<size_t>(m) * ldc + (head << 7) + d;
                *reinterpret_cast<half4_t *>(dst) = olo;
                *reinterpret_cast<half4_t *>(dst + 64) = ohi;
            } else if (ra.kv8) {
                typedef __attribute__((address_space(1))) unsigned *gu32_t;
                const gu32_t dst = (gu32_t)(static_cast<uint8_t *>(is_k ? ra.k_cache : ra.v_cache) + row_off + d);
                const float inv = is_k ? ra.k_inv_scale : ra.v_inv_scale;
                dst[0] = pack4_e4m3(to_f32(olo[0]) * inv, to_f32(olo[1]) * inv, to_f32(olo[2]) * inv, to_f32(olo[3]) * inv);
                dst[16] = pack4_e4m3(to_f32(ohi[0]) * inv, to_f32(ohi[1]) * inv, to_f32(ohi[2]) * inv, to_f32(ohi[3]) * inv);
            } else {
                typedef __attribute__((address_space(1))) half4_t *gh4_t;
                const gh4_t dst = (gh4_t)(static_cast<half_t *>(is_k ? ra.k_cache : ra.v_cache) + row_off + d);
                dst[0] = olo;
                dst[16] = ohi;
            }
        }
    }
}

// WN = MFMA column tiles per wave: 4 -> 256 x 256 workgroup tile, 2 -> 256 x 128 (one W half per stage, 96 KiB of LDS) for
// projections whose 256-wide grid would leave CUs idle (N = 4096 at 2048 tokens: 128 vs 256 workgroups).
// SWIGLU (WN = 4, W = fused gate_up [2I, K]): the workgroup's 256 weight rows are 128 gate rows n0 .. and the 128 up rows
// I + n0 .. of the same columns; every wave multiplies 32 gate and the matching 32 up columns of its 128 tokens, so
// act = silu(gate) * up is formed in registers and C is [M, I] -- the [M, 2I] intermediate and the SiluAndMul pass
// (silu_and_mul.cu:61-82) never touch memory.
template <bool FP8, bool HAS_EPI, int WN = 4, bool SWIGLU = false>
__global__ __launch_bounds__(512) void gemm256_kernel(const void *__restrict__ Xv, const void *__restrict__ Wv, half_t *C,
                                                      int M, int N, int K, const half_t *__restrict__ bias,
                                                      const half_t *residual, int tiles_n, const float *__restrict__ xscale,
                                                      const float *__restrict__ wscale, int ldc_arg = 0, int group_m = 0) {
    // ldc_arg != 0: C (and residual) rows are ldc_arg elements apart -- a launch over a column range [n_begin, n_begin + N)
    // of a wider output, with W / bias / wscale / C / residual pointers already advanced to n_begin
    const size_t ldc = ldc_arg ? ldc_arg : N;
    constexpr int ES = FP8 ? 1 : 2;            // bytes per element
    constexpr int BK = 128 / ES;               // k per tile: rows of 128 bytes either way
    static_assert(!SWIGLU || (WN == 4 && !HAS_EPI), "SwiGLU form: 256 weight rows = 128 gate + 128 up, no bias/residual");
    constexpr int BN = SWIGLU ? 128 : 64 * WN;  // workgroup tile columns of C
    const int half_n = N >> 1;                 // SWIGLU: I (N = 2I weight rows)
    constexpr int NHALF = SWIGLU ? 4 : 2 + BN / 128;  // 128-row half tiles per stage: X0 X1 W0 [W1]  (SWIGLU: X0 X1 gate up)
    constexpr int HALF_BYTES = 128 * 128, STAGE_BYTES = NHALF * HALF_BYTES;
    const unsigned char *X = static_cast<const unsigned char *>(Xv), *W = static_cast<const unsigned char *>(Wv);
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];  // [2][4][128 * 128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;  // wave grid 2 (M) x 4 (N)
    const int r = lane & 15, q = lane >> 4;
    // consecutive workgroups walk down M inside one 256-wide column of W: the W tile is shared through L2 by the
    // workgroups that are resident together, X (the smaller operand at prefill) is re-read per column
    int tile_m, tile_n;
    if (group_m > 0) {
        // XCD-aware tile order.  The dispatcher hands workgroup id i to XCD i % 8 and each XCD has its own L2: give every XCD a
        // CONTIGUOUS range of logical tiles, and walk that range in groups of group_m row tiles x all column tiles, M fastest --
        // the ~32 workgroups an XCD runs together then cover a compact block (group_m X tiles x 8 W tiles) whose operands
        // they share through that L2 instead of 32 unrelated tiles.
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int qd = nwg >> 3, rem = nwg & 7;
        const int L = xcd * qd + min(xcd, rem) + idx;  // XCD x owns qd + (x < rem) consecutive logical ids
        const int tm = (M + 255) / 256, per_group = group_m * tiles_n;
        const int grp = L / per_group, in = L - grp * per_group;
        const int first_m = grp * group_m, gsz = min(tm - first_m, group_m);
        tile_m = first_m + in % gsz;
        tile_n = in / gsz;
    } else {
        tile_m = blockIdx.x / tiles_n;
        tile_n = blockIdx.x - tile_m * tiles_n;
    }
    const int m0 = tile_m * 256, n0 = tile_n * BN;

    // ---- LDS-DMA plan: half-tile h (0,1 = X rows m0 + 128 h ..; 2,3 = W rows n0 + 128 (h-2) ..), instruction i (0,1):
    //      this wave fills rows (i*8 + wave)*8 .. +8 of the half; lane -> row + lane/8, slot lane%8 <- chunk slot ^ (row & 7)
    const unsigned char *src[NHALF][2];
#pragma unroll
    for (int h = 0; h < NHALF; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (i * 8 + wave) * 8 + (lane >> 3), slot = lane & 7, chunk = slot ^ (row & 7);
            int grow;  // clamped: edge rows are never stored
            if (h < 2) grow = min(m0 + h * 128 + row, M - 1);
            else if (SWIGLU) grow = (h - 2) * half_n + min(n0 + row, half_n - 1);  // half 2: gate rows, half 3: up rows
            else grow = min(n0 + (h - 2) * 128 + row, N - 1);
            src[h][i] = (h < 2 ? X : W) + static_cast<size_t>(grow) * K * ES + chunk * 16;
        }
    auto dma_tile = [&](int kt, int stage) {
#pragma unroll
        for (int h = 0; h < NHALF; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                unsigned char *dst = lds + stage * STAGE_BYTES + h * HALF_BYTES + (i * 8 + wave) * 1024;  // wave-uniform
                typedef const __attribute__((address_space(1))) void *gptr_t;
                typedef __attribute__((address_space(3))) void *lptr_t;
                __builtin_amdgcn_global_load_lds((gptr_t)(src[h][i] + static_cast<size_t>(kt) * 128), (lptr_t)dst, 16, 0, 0);
            }
    };

    floatx4 acc[8][WN];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses inside a stage: X row-tile i of this wave -> half wr, row i*16 + r; W col-tile j -> half 2 + (wc >> 1),
    // row (wc & 1) * 64 + j*16 + r; chunk c = ks*4 + q sits in slot c ^ (row & 7)
    const int wcol = SWIGLU ? wc * 32 : wc * 16 * WN;  // first column of this wave inside the workgroup tile
    const int a_row0 = r, b_row0 = (wcol & 127) + r;
    const unsigned char *a_base = lds + wr * HALF_BYTES, *b_base = lds + (2 + (SWIGLU ? 0 : (wcol >> 7))) * HALF_BYTES;
    // B fragment j of this wave: plain -> row b_row0 + 16 j of its half; SWIGLU -> j = 0,1 gate rows (half 2), j = 2,3 the same
    // rows of the up half (half 3)
    auto b_off = [&](int j) { return SWIGLU ? (j >> 1) * HALF_BYTES : 0; };
    auto b_row = [&](int j) { return SWIGLU ? b_row0 + (j & 1) * 16 : b_row0 + j * 16; };
    auto frag = [&](const unsigned char *base, int row, int c) {
        return *reinterpret_cast<const half8_t *>(base + row * 128 + ((c ^ (row & 7)) << 4));
    };

    const int KT = K / BK;
    dma_tile(0, 0);
    __syncthreads();  // drains the DMA (vmcnt(0)) and publishes stage 0
    for (int kt = 0; kt < KT; ++kt) {
        const int stage = kt & 1;
        if (kt + 1 < KT) dma_tile(kt + 1, stage ^ 1);  // lands while this k-tile is multiplied
        const unsigned char *ab = a_base + stage * STAGE_BYTES, *bb = b_base + stage * STAGE_BYTES;
        if constexpr (FP8) {
            // lane (r, q) supplies k bytes [16q, +16) and [64 + 16q, +16) of its row: chunks q and 4 + q (any lane -> k assignment is
            // fine as long as both operands use the same one; this one is conflict-free under the row & 7 swizzle)
            auto frag8 = [&](const unsigned char *base, int row) {
                const uint4_t lo = *reinterpret_cast<const uint4_t *>(base + row * 128 + ((q ^ (row & 7)) << 4));
                const uint4_t hi = *reinterpret_cast<const uint4_t *>(base + row * 128 + (((4 + q) ^ (row & 7)) << 4));
                return intx8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            };
            intx8 bf[WN];
#pragma unroll
            for (int j = 0; j < WN; ++j) bf[j] = frag8(bb + b_off(j), b_row(j));
#pragma unroll
            for (int ih = 0; ih < 2; ++ih) {
                intx8 af[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = frag8(ab, a_row0 + (ih * 4 + i) * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[ih * 4 + i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                            bf[j], af[i], acc[ih * 4 + i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8_t bf[WN];
#pragma unroll
            for (int j = 0; j < WN; ++j) bf[j] = frag(bb + b_off(j), b_row(j), ks * 4 + q);
#pragma unroll
            for (int ih = 0; ih < 2; ++ih) {
                half8_t af[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = frag(ab, a_row0 + (ih * 4 + i) * 16, ks * 4 + q);
                // D[n-row, m-col] convention: W as the MFMA A operand gives 4 consecutive n per lane (8-byte stores)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[ih * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[ih * 4 + i][j], 0, 0, 0);
            }
        }
        }
        __syncthreads();  // every wave done with this stage; the next stage's DMA has landed (the barrier drains vmcnt)
    }

    g256_store<FP8, HAS_EPI, WN, SWIGLU>(acc, C, M, N, ldc, m0, n0, wr, wcol, r, q, bias, residual, xscale, wscale);
}

}  // namespace llmie

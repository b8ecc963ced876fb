// Dense linear kernels for gfx950 behind llmie_linear / llmie_batched_gemm.
//
//   gemv_ksplit_kernel     M <= 8 tokens, fp16 W[N,K]: weight streaming; a workgroup owns RPW rows and
//                          splits K over its 256 threads, 16-byte non-temporal loads straight to VGPRs,
//                          activation slice (and the fused RMSNorm) in registers, v_dot2_f32_f16, wave64
//                          butterfly + one LDS exchange per iteration.  HBM-bound: bytes = N*K*2.
//   gemv_lds_kernel        same contract for K outside the register budget (x staged in LDS).
//   skinny_mfma_f16_kernel 1 <= M <= 64, fp16 W[N,K]: one 16-row weight tile per workgroup, the
//                          waves split K, v_mfma_f32_16x16x32_f16 with W as the A operand (so the
//                          weight fragment is one 16-byte load per lane) and x^T as B, LDS reduce
//                          across the K-split.  HBM-bound.
//   skinny_splitk_kernel   8 < M <= 64 per pass (fp16 / int8 / int4 / fp8): 64 weight rows x one K slice per workgroup,
//                          fp32 slabs [KS][M][N] reduced by their consumer.  (64 < M <= 128: gemm_mid.cuh;
//                          prefill-sized M: gemm256.cuh.)
//   tiled_mfma_f16_kernel  M > 64 (prefill): 128x128x32 LDS-tiled MFMA GEMM.  MFMA-bound.
//   generic_gemm_kernel    any dtype/transposition/shape: 64x64x16 LDS-tiled fp32 FMA.
//
// Epilogues fused here (the reference runs them as separate kernels): + bias[N], + residual[M,N],
// SwiGLU over row pairs (i, I+i) of a gate_up matrix.
#pragma once
#include "device_utils.cuh"

namespace llmie {

enum : int { EPI_NONE = 0, EPI_SWIGLU = 1 };


// ------------------------------------------------------------------------------------------
// Decode GEMV family:  y[m, r] = sum_k xn[m,k] * W[r,k]   (fp16, M <= 8 tokens, W row-major [N,K])
//   xn = x                                 (norm == 0)
//   xn = rmsnorm(x + pre_bias) * gamma     (norm != 0: the RMSNorm in front of the projection is computed
//        by every workgroup from the L2-resident activation row instead of by its own launch)
// epi EPI_NONE  : y[M,N] (+bias[N]) (+residual[M,N], may alias y)
// epi EPI_SWIGLU: y[M,N/2] = silu(gate) * up, rows (i, N/2+i) of a fused gate_up matrix
// ------------------------------------------------------------------------------------------
struct GemvArgs {
    const half_t *x;         // [M,K]
    const void *W;           // [N,K] fp16 | int8 | packed int4 (K/2 bytes per row)
    half_t *y;
    int K, N;
    const half_t *bias;      // [N] or null (EPI_NONE)
    const half_t *residual;  // [M,N] or null (EPI_NONE)
    const half_t *gamma;     // [K]   (norm)
    const half_t *pre_bias;  // [K] or null (norm)
    float eps;
    int epi, norm;
    const void *scale;       // int8: fp16 [N]; int4: fp16 [N, K/group]; fp16 weights: null
    int group;               // int4 group size (multiple of 32)
};

// ---- weight formats of the K-split GEMV: a 16-byte chunk holds 8 fp16, 16 int8 or 32 int4 weights ----
// int8 / int4 are de-quantised in registers with the fp16 "magic number" trick (0x6400 | u = 1024 + u exactly),
// two weights per v_perm/v_and + one packed subtract, and fed to v_dot2_f32_f16 like the fp16 stream; the
// per-row (int8) or per-(row, group) (int4) scale is applied to the fp32 partial sum.

template <int WBITS> struct WFmt;
template <> struct WFmt<16> { static constexpr int XE = 1; };  // half8 of x per chunk
template <> struct WFmt<8> { static constexpr int XE = 2; };
template <> struct WFmt<4> { static constexpr int XE = 4; };

__device__ __forceinline__ half2_t as_half2(unsigned int u) { return __builtin_bit_cast(half2_t, u); }

// dot of one 16-byte weight chunk with its activation chunk(s); x is in natural k order for 16/8 bit and in the
// permuted order produced by permute_x_int4() for 4 bit.
template <int WBITS, bool FP8 = false>
__device__ __forceinline__ float chunk_dot(const uint4_t w, const half8_t (&x)[WFmt<WBITS>::XE], float acc) {
    if constexpr (WBITS == 16) {
        return dot8(__builtin_bit_cast(half8_t, w), x[0], acc);
    } else if constexpr (WBITS == 8 && FP8) {
        // OCP e4m3 -> packed fp16 is one gfx950 instruction per pair (v_cvt_scalef32_pk_f16_fp8, scale 1.0)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const half2_t wlo = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(w[i]), 1.0f, false);
            const half2_t whi = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(static_cast<int>(w[i]), 1.0f, true);
            const half8_t xv = x[i >> 1];
            const int b = (i & 1) * 4;
            acc = __builtin_amdgcn_fdot2(wlo, half2_t{xv[b], xv[b + 1]}, acc, false);
            acc = __builtin_amdgcn_fdot2(whi, half2_t{xv[b + 2], xv[b + 3]}, acc, false);
        }
        return acc;
    } else if constexpr (WBITS == 8) {
        // bytes b0..b3 of a word -> halves (b0,b1), (b2,b3) as 1024 + (b ^ 0x80) = 1152 + int8, multiplied AS THEY ARE: the
        // caller starts `acc` at -1152 * (sum of this thread's activation slice), once per token instead of a packed subtract
        // per weight pair (products of two fp16 values are exact in the fp32 dot; the partial sums stay below 2^24 ulps of
        // the result's scale: |x| * 1279 * k-slice against fp32's 24 bits)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned int v = w[i] ^ 0x80808080u;
            const unsigned int lo = __builtin_amdgcn_perm(0x64646464u, v, 0x04010400u);  // {0x64,b1,0x64,b0}
            const unsigned int hi = __builtin_amdgcn_perm(0x64646464u, v, 0x04030402u);  // {0x64,b3,0x64,b2}
            const half2_t wlo = as_half2(lo), whi = as_half2(hi);
            const half8_t xv = x[i >> 1];
            const int b = (i & 1) * 4;
            acc = __builtin_amdgcn_fdot2(wlo, half2_t{xv[b], xv[b + 1]}, acc, false);
            acc = __builtin_amdgcn_fdot2(whi, half2_t{xv[b + 2], xv[b + 3]}, acc, false);
        }
        return acc;
    } else {
        // word i holds k = 8i..8i+7 as nibbles n0..n7.  (w & 0x000F000F)|0x6400.. = (1024+n0, 1024+n4); the nibbles at
        // bits 4-7 land 4 mantissa bits higher: (w & 0x00F000F0)|0x6400.. = (1024+16 n1, 1024+16 n5); one shift by 8 exposes
        // n2,n6 / n3,n7 to the same two masks.  The halves are multiplied AS THEY ARE: the caller has divided the activations that
        // meet the 16 n halves by 16 (prescale_x_int4) and starts `acc` at the matching offset term (int4_offset_term), so the
        // "- 8" of every nibble costs one fp32 term per 32-weight chunk and token instead of a packed subtract / fma per weight
        // pair -- the GEMV is VALU-bound in this format (profiles/r02_int4_int8_valu_pmc.csv).
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const half8_t xv = x[i];  // permuted and prescaled: (k0,k4,k1/16,k5/16,k2,k6,k3/16,k7/16) of this word
            const unsigned int w0 = w[i], w8 = w[i] >> 8;
            const half2_t h0 = as_half2((w0 & 0x000F000Fu) | 0x64006400u);
            const half2_t h1 = as_half2((w0 & 0x00F000F0u) | 0x64006400u);
            const half2_t h2 = as_half2((w8 & 0x000F000Fu) | 0x64006400u);
            const half2_t h3 = as_half2((w8 & 0x00F000F0u) | 0x64006400u);
            acc = __builtin_amdgcn_fdot2(h0, half2_t{xv[0], xv[1]}, acc, false);
            acc = __builtin_amdgcn_fdot2(h1, half2_t{xv[2], xv[3]}, acc, false);
            acc = __builtin_amdgcn_fdot2(h2, half2_t{xv[4], xv[5]}, acc, false);
            acc = __builtin_amdgcn_fdot2(h3, half2_t{xv[6], xv[7]}, acc, false);
        }
        return acc;
    }
}
// activation order matching chunk_dot<4>: within each group of 8: (0,4,1,5,2,6,3,7); the elements that meet the nibbles read as
// 1024 + 16 n (k1, k5, k3, k7) are divided by 16 (exact in fp16 above 2^-10; smaller activations round in the subnormal range)
__device__ __forceinline__ half8_t prescale_x_int4(const half8_t v) {
    const float s = 0.0625f;
    return half8_t{v[0], v[1], from_f32<half_t>(to_f32(v[2]) * s), from_f32<half_t>(to_f32(v[3]) * s),
                   v[4], v[5], from_f32<half_t>(to_f32(v[6]) * s), from_f32<half_t>(to_f32(v[7]) * s)};
}
// sum over one 8-weight word of what chunk_dot<4> adds beyond x . (n - 8): (1024 + 8) x for the plain halves, (64 + 8) x =
// 1152 (x / 16) for the prescaled ones; the negative of the chunk's total is the start value of its dot
__device__ __forceinline__ float int4_offset_term(const half8_t v) {
    return 1032.0f * (to_f32(v[0]) + to_f32(v[1]) + to_f32(v[4]) + to_f32(v[5])) +
           1152.0f * (to_f32(v[2]) + to_f32(v[3]) + to_f32(v[6]) + to_f32(v[7]));
}
// activation order matching chunk_dot<4>: within each group of 8: (0,4,1,5,2,6,3,7)
__device__ __forceinline__ half8_t permute_x_int4(const half8_t v) {
    return half8_t{v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7]};
}

// K-split GEMV (the hot kernel).  One workgroup = 256 threads owns RPW rows per iteration and splits
// K: thread t streams the 16-byte chunks t, t+256, ... of every row (a workgroup instruction covers
// 4 KiB contiguous of one row), so the activation slice a thread needs is XC chunks and lives in
// registers -- no LDS staging, no barrier in front of the stream; RPW*XC loads are in flight per lane.
// The first group's loads are issued before the (fused) RMSNorm prologue and the next group's loads
// right after the dot products, before the cross-wave reduction.  The 4 waves meet once per iteration to
// add their partial sums through a double-buffered LDS slot.
// Measured fp16 (tools/kbench.py, MI355X, incl. ~1 us launch gap): QKV 100.7 MB 16.7 us, O 33.6 MB
// 7.6 us, down 90.2 MB 16.1 us, LM head 262 MB 42 us = 6.0 / 4.4 / 5.6 / 6.2 TB/s.
// FP8 (WBITS == 8): e4m3 weights with fp32 per-row scales, and the activation row is quantised per token to the e4m3
// grid in the prologue (scale amax/448, exactly quantize_rows_fp8_kernel's arithmetic) so that the result equals the
// fp8 MFMA path's  wscale[n] * xscale[m] * sum_k Wq[n,k] xq[m,k]  up to summation order.
// RI = rows per workgroup instruction: a row of at most 2 KiB (int4 at K = 4096: 128 chunks) would leave half of the 256 threads
// multiplying a clamped chunk with zero activations -- with RI = 2 the two 128-thread halves of the workgroup stream two
// CONSECUTIVE rows (still one contiguous 4-KiB instruction), waves 0-1 reduce the even row and waves 2-3 the odd one.  (Also measured for the 7B down projection, 344 chunks as
// 3 x 128 - 40 slots instead of 2 x 256 - 168: 12.0 -> 11.6 us, within the noise -- that launch is not VALU-bound; not dispatched.)
template <int M, int RPW, int XC, int WBITS, bool DB = false, bool FP8 = false, int RI = 1>
__global__ __launch_bounds__(256) void gemv_ksplit_kernel(const GemvArgs a) {
    // DB (quantised weights): two weight register sets -- the next group's loads are issued BEFORE the current group's
    // de-quantise + dot phase (~2 us of VALU per group for int4), which would otherwise run with nothing in flight.
    static_assert(RPW % 2 == 0, "rows come in pairs");
    constexpr int XE = WFmt<WBITS>::XE;   // half8 activations per weight chunk
    constexpr int EPC = 8 * XE;           // weights per 16-byte chunk
    __shared__ float red[2][4][M * RPW];
    __shared__ float ssq[4][M];
    const int K = a.K, N = a.N;
    const int nch = K / EPC;              // chunks per row
    const int nx8 = K >> 3;               // half8 per activation row
    const size_t row_bytes = static_cast<size_t>(K) * WBITS / 8;
    static_assert(RI == 1 || RI == 2, "one or two rows per workgroup instruction");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int TPR = 256 / RI, WPR = 4 / RI;   // threads / waves per row
    const int sub = tid / TPR, ct = tid % TPR;    // which row of the instruction, chunk index inside the row
    const bool swiglu = a.epi == EPI_SWIGLU;
    const int half_n = N >> 1;
    const int out_n = swiglu ? half_n : N;
    // a group = RPW instructions of RI rows: EPI_NONE rows [g*RPW*RI, ...); SWIGLU RPW/2 instructions of gate rows + their up rows
    const int ngroups = swiglu ? (half_n + (RPW / 2) * RI - 1) / ((RPW / 2) * RI) : (N + RPW * RI - 1) / (RPW * RI);
    auto row_of = [&](int grp, int r) {
        int row = swiglu ? (grp * (RPW / 2) + (r >> 1)) * RI + sub + (r & 1) * half_n : (grp * RPW + r) * RI + sub;
        const int lim = swiglu ? ((r & 1) ? N : half_n) : N;
        return row < lim ? row : lim - 1;  // clamp: result of a clamped row is never stored
    };

    // ---- activation slice of this thread (issued first: L2), then the first weight group (HBM) ----
    half8_t xr[M][XC][XE];
    const half8_t *xg = reinterpret_cast<const half8_t *>(a.x);
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int j = 0; j < XC; ++j) {
            const int cc = j * TPR + ct;
#pragma unroll
            for (int e = 0; e < XE; ++e)
                xr[m][j][e] = cc < nch ? xg[m * nx8 + cc * XE + e] : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
    half8_t g[XC][XE];
    if (a.norm) {
        const half8_t *gm = reinterpret_cast<const half8_t *>(a.gamma);
#pragma unroll
        for (int j = 0; j < XC; ++j) {
            const int cc = j * TPR + ct;
#pragma unroll
            for (int e = 0; e < XE; ++e) g[j][e] = cc < nch ? gm[cc * XE + e] : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    constexpr int SR = WBITS == 4 ? RPW : 1, SX = WBITS == 4 ? XC : 1;
    uint4_t wbA[RPW][XC], wbB[DB ? RPW : 1][DB ? XC : 1];
    half_t wscA[SR][SX], wscB[DB ? SR : 1][DB ? SX : 1];  // int4: per-(row, k-group) scales, loaded with the weights
    const unsigned char *Wb = reinterpret_cast<const unsigned char *>(a.W);
    const half_t *scale = reinterpret_cast<const half_t *>(a.scale);
    const int sgroups = (WBITS == 4) ? K / a.group : 1;  // scales per row
    auto load_group = [&](int grp, auto &wb, auto &wsc) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const uint4_t *w = reinterpret_cast<const uint4_t *>(Wb + static_cast<size_t>(row_of(grp, r)) * row_bytes);
            if constexpr (WBITS == 4) {
                // a 32-weight chunk lies inside one scale group (group % 32 == 0); issued with the weight loads so
                // the scale's L2 round trip overlaps the HBM one instead of following it
                const half_t *srow = scale + static_cast<size_t>(row_of(grp, r)) * sgroups;
#pragma unroll
                for (int j = 0; j < XC; ++j) wsc[r][j] = srow[(min(j * TPR + ct, nch - 1) * EPC) / a.group];
            }
#pragma unroll
            for (int j = 0; j < XC; ++j) {
                // unconditional load (a predicated one would put every load in its own exec-masked region); chunks past
                // the row end re-read the last chunk and meet an all-zero activation slice
                const int cc = min(j * TPR + ct, nch - 1);
                wb[r][j] = load_nt(w + cc);
            }
        }
    };
    int grp = blockIdx.x;
    if (grp < ngroups) load_group(grp, wbA, wscA);  // in flight while the norm below waits only for x / gamma

    if (a.norm) {
        const half8_t *pb = reinterpret_cast<const half8_t *>(a.pre_bias);
        if (pb) {
#pragma unroll
            for (int j = 0; j < XC; ++j) {
                const int cc = j * TPR + ct;
                if (cc < nch) {
#pragma unroll
                    for (int e = 0; e < XE; ++e) {
                        const half8_t b = pb[cc * XE + e];
#pragma unroll
                        for (int m = 0; m < M; ++m)
#pragma unroll
                            for (int i = 0; i < 8; ++i) xr[m][j][e][i] = from_f32<half_t>(to_f32(xr[m][j][e][i]) + to_f32(b[i]));
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < XC; ++j)
#pragma unroll
                for (int e = 0; e < XE; ++e) ss = dot8(xr[m][j][e], xr[m][j][e], ss);
            ss = wave_sum(ss);
            if (lane == 0) ssq[wave][m] = ss;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const float inv = rsqrtf((RI == 1 ? ssq[0][m] + ssq[1][m] + ssq[2][m] + ssq[3][m] : ssq[0][m] + ssq[1][m]) / static_cast<float>(K) + a.eps);   // RI = 2: both halves hold the whole row
#pragma unroll
            for (int j = 0; j < XC; ++j)
#pragma unroll
                for (int e = 0; e < XE; ++e)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        xr[m][j][e][i] = from_f32<half_t>(to_f32(xr[m][j][e][i]) * to_f32(g[j][e][i]) * inv);
        }
    }
    if constexpr (WBITS == 4) {
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int j = 0; j < XC; ++j)
#pragma unroll
                for (int e = 0; e < XE; ++e) xr[m][j][e] = prescale_x_int4(permute_x_int4(xr[m][j][e]));
    }
    float xcorr4[WBITS == 4 ? M : 1][WBITS == 4 ? XC : 1];   // int4: start value of every 32-weight chunk's dot
    if constexpr (WBITS == 4) {
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int j = 0; j < XC; ++j) {
                float t = 0.f;
#pragma unroll
                for (int e = 0; e < XE; ++e) t += int4_offset_term(xr[m][j][e]);
                xcorr4[m][j] = -t;
            }
    }
    __shared__ float amx[FP8 ? 4 : 1][M], xscale_sh[M];  // FP8: per-token activation amax / scale
    if constexpr (FP8) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float amax = 0.f;
#pragma unroll
            for (int j = 0; j < XC; ++j)
#pragma unroll
                for (int e = 0; e < XE; ++e)
#pragma unroll
                    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(to_f32(xr[m][j][e][i])));
            amax = wave_max(amax);
            if (lane == 0) amx[wave][m] = amax;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const float amax = fmaxf(fmaxf(amx[0][m], amx[1][m]), fmaxf(amx[2][m], amx[3][m]));
            const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
            if (tid == 0) xscale_sh[m] = sc;  // read in the epilogue, after at least one more barrier
#pragma unroll
            for (int j = 0; j < XC; ++j)
#pragma unroll
                for (int e = 0; e < XE; ++e)
#pragma unroll
                    for (int i = 0; i < 8; i += 2) {
                        const float f0 = fminf(fmaxf(to_f32(xr[m][j][e][i]) / sc, -448.f), 448.f);
                        const float f1 = fminf(fmaxf(to_f32(xr[m][j][e][i + 1]) / sc, -448.f), 448.f);
                        const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(f0, f1, 0, false);
                        const half2_t h = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(pk, 1.0f, false);
                        xr[m][j][e][i] = h[0];
                        xr[m][j][e][i + 1] = h[1];
                    }
        }
    }

    // int8: chunk_dot<8> multiplies 1152 + w; every dot of this thread starts at -1152 * sum(x slice)
    float xcorr[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        xcorr[m] = 0.f;
        if constexpr (WBITS == 8 && !FP8) {
            float sx = 0.f;
#pragma unroll
            for (int j = 0; j < XC; ++j)
#pragma unroll
                for (int e = 0; e < XE; ++e)
#pragma unroll
                    for (int i = 0; i < 8; ++i) sx += to_f32(xr[m][j][e][i]);
            xcorr[m] = -1152.0f * sx;
        }
    }
    auto step = [&](auto &wb, auto &wsc, auto &wbn, auto &wscn, const int cur, const int it) {
        const int nxt = cur + gridDim.x;
        if constexpr (DB) {
            if (nxt < ngroups) load_group(nxt, wbn, wscn);
        }
        float acc[M][RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            float sc[XC];
            if constexpr (WBITS == 4) {
#pragma unroll
                for (int j = 0; j < XC; ++j) sc[j] = to_f32(wsc[r][j]);  // out-of-range chunks meet zero activations
            }
#pragma unroll
            for (int m = 0; m < M; ++m) {
                float sdot = xcorr[m];
#pragma unroll
                for (int j = 0; j < XC; ++j) {
                    if constexpr (WBITS == 4) sdot = fmaf(sc[j], chunk_dot<4>(wb[r][j], xr[m][j], xcorr4[m][j]), sdot);
                    else sdot = chunk_dot<WBITS, FP8>(wb[r][j], xr[m][j], sdot);
                }
                acc[m][r] = sdot;
            }
        }
        // single buffer: the weight registers are dead now -- next group's loads go out before the reduction/barrier
        if constexpr (!DB) {
            if (nxt < ngroups) load_group(nxt, wb, wsc);
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int m = 0; m < M; ++m) acc[m][r] = wave_sum(acc[m][r]);
        float *slot = &red[it & 1][0][0];
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < M; ++m)
#pragma unroll
                for (int r = 0; r < RPW; ++r) slot[wave * (M * RPW) + m * RPW + r] = acc[m][r];
        }
        __syncthreads();  // one barrier per iteration: the other parity slot is free for the next iteration
        auto row_scale = [&](int row) {
            if constexpr (FP8) return reinterpret_cast<const float *>(a.scale)[row];
            else return (WBITS == 8) ? to_f32(scale[row]) : 1.0f;
        };
        // row h (< RI) of instruction r was summed by waves h*WPR .. h*WPR + WPR - 1
        if (swiglu) {
            if (tid < M * (RPW / 2) * RI) {
                const int m = tid / ((RPW / 2) * RI), qh = tid % ((RPW / 2) * RI), q = qh / RI, h = qh % RI;
                const int col = (cur * (RPW / 2) + q) * RI + h;
                if (col < half_n) {
                    float gt = 0.f, up = 0.f;
#pragma unroll
                    for (int w = 0; w < WPR; ++w) {
                        gt += slot[(h * WPR + w) * (M * RPW) + m * RPW + 2 * q];
                        up += slot[(h * WPR + w) * (M * RPW) + m * RPW + 2 * q + 1];
                    }
                    gt *= row_scale(col) * (FP8 ? xscale_sh[m] : 1.0f);
                    up *= row_scale(col + half_n) * (FP8 ? xscale_sh[m] : 1.0f);
                    a.y[static_cast<size_t>(m) * out_n + col] = from_f32<half_t>((gt / (1.0f + expf(-gt))) * up);
                }
            }
        } else {
            if (tid < M * RPW * RI) {
                const int m = tid / (RPW * RI), rh = tid % (RPW * RI), r = rh / RI, h = rh % RI;
                const int col = (cur * RPW + r) * RI + h;
                if (col < N) {
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < WPR; ++w) v += slot[(h * WPR + w) * (M * RPW) + m * RPW + r];
                    v *= row_scale(col) * (FP8 ? xscale_sh[m] : 1.0f);
                    if (a.bias) v += to_f32(a.bias[col]);
                    if (a.residual) v += to_f32(a.residual[static_cast<size_t>(m) * N + col]);
                    a.y[static_cast<size_t>(m) * N + col] = from_f32<half_t>(v);
                }
            }
        }
    };
    for (int it = 0; grp < ngroups;) {
        step(wbA, wscA, wbB, wscB, grp, it);
        grp += gridDim.x;
        ++it;
        if constexpr (DB) {
            if (grp < ngroups) {
                step(wbB, wscB, wbA, wscA, grp, it);
                grp += gridDim.x;
                ++it;
            }
        }
    }
}

// Fallback for K outside the K-split kernel's register budget (any K % 8 == 0 with M*K*2 <= 64 KB):
// x staged once per workgroup in LDS, one wave per row pair, 16 loads single shot.
template <int M>
__global__ __launch_bounds__(256) void gemv_lds_kernel(const GemvArgs a) {
    constexpr int U = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half8_t *xs = reinterpret_cast<half8_t *>(smem_raw);  // [M][K/8]
    __shared__ float inv_rms[M];
    const int K = a.K, N = a.N;
    const int nch = K >> 3, total = M * nch;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool swiglu = a.epi == EPI_SWIGLU;
    {
        const half8_t *xg = reinterpret_cast<const half8_t *>(a.x);
        const half8_t *pb = reinterpret_cast<const half8_t *>(a.pre_bias);
        for (int i = tid; i < total; i += 256) {
            half8_t v = xg[i];
            if (a.norm && pb) {
                const half8_t b = pb[i % nch];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = from_f32<half_t>(to_f32(v[e]) + to_f32(b[e]));
            }
            xs[i] = v;
        }
    }
    __syncthreads();
    if (a.norm) {
        for (int m = wave; m < M; m += 4) {
            float ss = 0.f;
            for (int c = lane; c < nch; c += 64) {
                const half8_t v = xs[m * nch + c];
                ss = dot8(v, v, ss);
            }
            ss = wave_sum(ss);
            if (lane == 0) inv_rms[m] = rsqrtf(ss / static_cast<float>(K) + a.eps);
        }
        __syncthreads();
        const half8_t *gm = reinterpret_cast<const half8_t *>(a.gamma);
        for (int i = tid; i < total; i += 256) {
            const int m = i / nch, c = i - m * nch;
            half8_t v = xs[i];
            const half8_t g = gm[c];
            const float sc = inv_rms[m];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = from_f32<half_t>(to_f32(v[e]) * to_f32(g[e]) * sc);
            xs[i] = v;
        }
        __syncthreads();
    }
    const int half_n = N >> 1;
    const int npairs = swiglu ? half_n : ((N + 1) >> 1);
    for (int pair = blockIdx.x * 4 + wave; pair < npairs; pair += gridDim.x * 4) {
        const int r0 = swiglu ? pair : 2 * pair;
        const int r1 = swiglu ? pair + half_n : min(2 * pair + 1, N - 1);  // odd N: duplicate row, result discarded
        const half8_t *w0 = reinterpret_cast<const half8_t *>(static_cast<const half_t *>(a.W) + static_cast<size_t>(r0) * K);
        const half8_t *w1 = reinterpret_cast<const half8_t *>(static_cast<const half_t *>(a.W) + static_cast<size_t>(r1) * K);
        float acc0[M], acc1[M];
#pragma unroll
        for (int m = 0; m < M; ++m) acc0[m] = acc1[m] = 0.f;
        for (int c = lane; c < nch; c += 64 * U) {
            half8_t a0[U], a1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cc = c + 64 * u;
                if (cc < nch) {
                    a0[u] = load_nt(w0 + cc);
                    a1[u] = load_nt(w1 + cc);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cc = c + 64 * u;
                if (cc < nch) {
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        const half8_t xv = xs[m * nch + cc];
                        acc0[m] = dot8(a0[u], xv, acc0[m]);
                        acc1[m] = dot8(a1[u], xv, acc1[m]);
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            acc0[m] = wave_sum(acc0[m]);
            acc1[m] = wave_sum(acc1[m]);
        }
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                if (swiglu) {
                    const float gt = acc0[m], up = acc1[m];
                    a.y[static_cast<size_t>(m) * half_n + pair] = from_f32<half_t>((gt / (1.0f + expf(-gt))) * up);
                } else {
                    float v0 = acc0[m], v1 = acc1[m];
                    const bool has1 = 2 * pair + 1 < N;
                    if (a.bias) {
                        v0 += to_f32(a.bias[r0]);
                        v1 += to_f32(a.bias[r1]);
                    }
                    if (a.residual) {
                        v0 += to_f32(a.residual[static_cast<size_t>(m) * N + r0]);
                        if (has1) v1 += to_f32(a.residual[static_cast<size_t>(m) * N + r1]);
                    }
                    a.y[static_cast<size_t>(m) * N + r0] = from_f32<half_t>(v0);
                    if (has1) a.y[static_cast<size_t>(m) * N + r1] = from_f32<half_t>(v1);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Skinny MFMA: D[n, m] = sum_k W[n,k] x[m,k]; one workgroup = NT 16-row weight tiles, NW waves
// split K in interleaved 32-wide steps.  MT = ceil(M/16) column tiles.
// A fragment (weights): lane l -> W[n0 + (l&15)][k + 8*(l>>4) .. +8]   (16-byte global load)
// B fragment (x^T)    : lane l -> x[m0 + (l&15)][k + 8*(l>>4) .. +8]   (16-byte load, L2 resident)
// D: lane l holds rows n0 + 4*(l>>4) + {0..3}, column m0 + (l&15).
// ------------------------------------------------------------------------------------------
template <int MT, int NT, int NW, int EPI>
__global__ __launch_bounds__(NW * 64) void skinny_mfma_f16_kernel(const half_t *__restrict__ x,
                                                                  const half_t *__restrict__ W,
                                                                  half_t *__restrict__ y, int M, int K,
                                                                  int N, const half_t *__restrict__ bias,
                                                                  const half_t *residual) {
    __shared__ floatx4 red[NW][NT * MT][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int half_n = N >> 1;
    // tile rows: EPI_SWIGLU pairs tile t (gate rows) with the same rows + N/2 (up rows): NT must be 2
    int nrow[NT];
    if constexpr (EPI == EPI_SWIGLU) {
        static_assert(NT == 2, "swiglu epilogue pairs a gate tile with its up tile");
        nrow[0] = blockIdx.x * 16 + r;
        nrow[1] = nrow[0] + half_n;
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) nrow[t] = (blockIdx.x * NT + t) * 16 + r;
    }
    const int nlimit = (EPI == EPI_SWIGLU) ? half_n : N;
    const half_t *wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int rr = (EPI == EPI_SWIGLU) ? min(nrow[0], half_n - 1) + t * half_n : min(nrow[t], N - 1);
        wp[t] = W + static_cast<size_t>(rr) * K + 8 * q;
    }
    const half_t *xp[MT];
    bool xok[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        xok[j] = (16 * j + r) < M;
        xp[j] = x + static_cast<size_t>(xok[j] ? 16 * j + r : 0) * K + 8 * q;
    }
    floatx4 acc[NT][MT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[t][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    constexpr int U = 4;  // k-steps in flight per wave
    const int ksteps = K >> 5;
    // all loads unconditional: rows past M read row 0 and k-steps past the end re-read the last step; their B
    // fragments are zeroed by a select after the load, so no load sits in its own exec-masked region
    // de-phase the workgroups along K: all workgroups start together and advance at the same rate, and a weight
    // fragment load touches 16 rows that are K*2 bytes apart -- without the rotation every workgroup hits the same
    // HBM channels at the same time
    const int rot = (blockIdx.x * 5) % ksteps;
    for (int s0 = wave * U; s0 < ksteps; s0 += NW * U) {
        half8_t a[U][NT], b[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int s = min(s0 + u, ksteps - 1) + rot;
            s = s >= ksteps ? s - ksteps : s;
#pragma unroll
            for (int t = 0; t < NT; ++t) a[u][t] = load_nt(reinterpret_cast<const half8_t *>(wp[t] + 32 * s));
#pragma unroll
            for (int j = 0; j < MT; ++j) b[u][j] = *reinterpret_cast<const half8_t *>(xp[j] + 32 * s);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = s0 + u < ksteps;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const half8_t bz = (live && xok[j]) ? b[u][j] : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u][t], bz, acc[t][j], 0, 0, 0);
            }
        }
    }
    // cross-wave (K-split) reduction through LDS
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < MT; ++j) red[wave][t * MT + j][lane] = acc[t][j];
    __syncthreads();
    // wave w finishes tiles w, w+NW, ...
    constexpr int TILES = (EPI == EPI_SWIGLU) ? MT : NT * MT;
    for (int tile = wave; tile < TILES; tile += NW) {
        const int t = (EPI == EPI_SWIGLU) ? 0 : tile / MT;
        const int j = (EPI == EPI_SWIGLU) ? tile : tile % MT;
        floatx4 s = red[0][t * MT + j][lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) s += red[w][t * MT + j][lane];
        const int m = 16 * j + r;
        if constexpr (EPI == EPI_SWIGLU) {
            floatx4 up = red[0][1 * MT + j][lane];
#pragma unroll
            for (int w = 1; w < NW; ++w) up += red[w][1 * MT + j][lane];
            const int n0 = blockIdx.x * 16 + 4 * q;
            if (m < M) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n0 + e < nlimit) {
                        const float g = s[e];
                        y[static_cast<size_t>(m) * half_n + n0 + e] = from_f32<half_t>((g / (1.0f + expf(-g))) * up[e]);
                    }
            }
        } else {
            const int n0 = (blockIdx.x * NT + t) * 16 + 4 * q;
            if (m < M) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n0 + e < nlimit) {
                        float v = s[e];
                        if (bias) v += to_f32(bias[n0 + e]);
                        if (residual) v += to_f32(residual[static_cast<size_t>(m) * N + n0 + e]);
                        y[static_cast<size_t>(m) * N + n0 + e] = from_f32<half_t>(v);
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Split-K skinny MFMA GEMM for decode batches / short prefills (8 < M <= 64 per pass), fp16 or int8 weights.
//   Every workgroup must read the activation slice it multiplies, so activation traffic = (#workgroups per full
//   K sweep) x |x|.  With one 16/32-row tile per workgroup (first skinny kernel above) that is 4x the weight
//   stream at M = 64 and the kernel is L2-bound (measured 1.6-2.2 TB/s).  Here a workgroup owns 64 weight rows
//   (one 16-row MFMA tile per wave) x one K slice: its 4 waves read the SAME x fragments (L1 hits after the first),
//   so L2 activation traffic is (N/64) x |x|, and the grid is filled by splitting K over workgroups (KS slices);
//   partial tiles go to an fp32 slab [KS][M][N] that skinny_finalize_kernel sums (+ scale, bias, residual, SwiGLU).
//   A fragment (weights): lane l -> W[n0 + (l&15)][k + 8*(l>>4) .. +8] (int8: 16 consecutive k -> two fragments)
//   D: lane l holds rows n0 + 4*(l>>4) + {0..3}, column m0 + (l&15) -> 16-byte fp32 stores into the slab.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ half8_t dequant_i8x8(unsigned int w0, unsigned int w1) {
    const half2_t off = {static_cast<half_t>(1152.f), static_cast<half_t>(1152.f)};
    const unsigned int v0 = w0 ^ 0x80808080u, v1 = w1 ^ 0x80808080u;
    const half2_t a = as_half2(__builtin_amdgcn_perm(0x64646464u, v0, 0x04010400u)) - off;
    const half2_t b = as_half2(__builtin_amdgcn_perm(0x64646464u, v0, 0x04030402u)) - off;
    const half2_t c = as_half2(__builtin_amdgcn_perm(0x64646464u, v1, 0x04010400u)) - off;
    const half2_t d = as_half2(__builtin_amdgcn_perm(0x64646464u, v1, 0x04030402u)) - off;
    return half8_t{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}

// FP8 (WBITS == 8): e4m3 weights AND e4m3 activations (x is [M,K] bytes, quantised per token by the producer), fed to
// v_mfma_f32_16x16x32_fp8_fp8 without any conversion; the scales (wscale[n] * xscale[m]) are applied by the slab consumer.
// int4 (WBITS == 4, group size 128): a 16-byte load = 32 weights = 4 MFMA k-steps of one row; a sub-block (512 k) spans 4
// scale groups, one per fragment: each group's products are accumulated from zero on the MFMA and folded into the running
// total with the fp32 scale of (row, group) -- exact group scaling, 8 fp32 FMAs per output tile instead of a multiply per
// weight.  Nibbles come out of the word in the order (0,4,1,5,2,6,3,7): the activation chunks are permuted to match when
// they are staged.  The slabs hold fully scaled values (no consumer-side scale).  K % 32 == 0; a ragged last sub-block
// multiplies clamped (in-bounds) weight bytes with zeroed activations.
template <int MT, int WBITS, bool FP8 = false>
__global__ __launch_bounds__(256) void skinny_splitk_kernel(const void *__restrict__ xv, const void *__restrict__ W,
                                                            float *__restrict__ slab, int M, int K, int N, int KS,
                                                            int blocks_per_slice, const half_t *__restrict__ gscale) {
    // One sub-block = 4 weight loads per lane: fp16 BK = 128 (4 steps of 32), int8 / fp8 BK = 256 (4 steps of 64).
    // The activation tile [16*MT rows][BK] of the sub-block is staged in LDS (coalesced 16-byte loads, chunks
    // XOR-swizzled with the row so a fragment read -- 16 rows x one chunk column -- is conflict-free) and shared by
    // the 4 waves: B fragments are ds_read_b128 (measured: fragment-shaped global loads of x, even L1-resident,
    // bound the first versions of this kernel at ~90 cycles per wave-load).  Weight fragments go HBM -> VGPR with
    // the next sub-block's loads in flight while the current one is multiplied; x tiles are double buffered.
    static_assert(!FP8 || WBITS == 8, "fp8 is an 8-bit format");
    constexpr int KSTEP = (WBITS == 16) ? 32 : (WBITS == 4 ? 128 : 64);
    constexpr int BK = 4 * KSTEP;
    constexpr int ROWS = 16 * MT;
    constexpr int XB = FP8 ? 1 : 2;                // bytes per activation element
    constexpr int RB = BK * XB;                    // bytes per tile row
    constexpr int CPR = RB / 16;                   // 16-byte chunks per tile row
    // activation staging group: G sub-blocks at once (all their loads in flight together, one barrier pair per group)
    // G * ROWS * RB bytes <= 32 KB of LDS so that several workgroups stay resident per CU
    constexpr int G = 32768 / (ROWS * RB) >= 1 ? (32768 / (ROWS * RB) > 8 ? 8 : 32768 / (ROWS * RB)) : 1;
    constexpr int GCH = G * ROWS * CPR;               // 16-byte chunks per group
    constexpr int XCH = (GCH + 255) / 256;            // staging chunks per thread
    __shared__ __attribute__((aligned(16))) unsigned char xs[G][ROWS * RB];
    const unsigned char *x = static_cast<const unsigned char *>(xv);
    const size_t x_row_bytes = static_cast<size_t>(K) * XB;
    // per-wave weight transposer: a sub-block of the wave's 16 rows is 16 x 256 B; it is LOADED row-contiguously
    // (one wave instruction = 4 rows x 256 B: full cache lines, DRAM-friendly -- fragment-shaped global loads of
    // 16 rows x 64 B capped this kernel at ~3.8 TB/s) and re-read as MFMA fragments through 4 KB of private LDS.
    __shared__ __attribute__((aligned(16))) unsigned char wsm[4][16 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int tile = blockIdx.x / KS, ks = blockIdx.x - tile * KS;
    const int n0 = tile * 64 + wave * 16;
    const size_t row_bytes = static_cast<size_t>(K) * WBITS / 8;
    // load u of a sub-block: lane -> row 4u + (lane >> 4), 16-byte chunk (lane & 15)
    const unsigned char *wp = static_cast<const unsigned char *>(W) + 16 * r;
    size_t wrow_off[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) wrow_off[u] = static_cast<size_t>(min(n0 + 4 * u + q, N - 1)) * row_bytes;
    const int nblocks = (K + BK - 1) / BK;  // int4: the last sub-block may be ragged
    const int b_begin = ks * blocks_per_slice, b_end = min(nblocks, b_begin + blocks_per_slice);
    const int sgroups = K / 128;            // int4: scales per weight row

    // weight fragments: ring of 4 sub-blocks (16 loads in flight per lane), refilled as soon as a slot is consumed
#ifndef LLMIE_SK_RING_MIN
#define LLMIE_SK_RING_MIN 2
#endif
    // ring depth: 4 sub-blocks, but no deeper than the staging group unless that is a single sub-block (measured on
    // MI355X: a 4-deep ring with G = 2 costs registers/occupancy and ~4 us per launch at M = 32)
    constexpr int R = G >= 4 ? 4 : (G > LLMIE_SK_RING_MIN ? G : LLMIE_SK_RING_MIN);
    constexpr int U = G > R ? G : R;     // main-loop unroll: ring slot (i % R) and group slot (i % G) both compile-time
    uint4_t a[R][4];
    uint2 gsc[WBITS == 4 ? R : 1][4];  // int4: scales of D rows n0 + 4q + e for the sub-block's 4 groups
    uint4_t xr[XCH];
    auto load_a = [&](int blk, uint4_t(&dst)[4], uint2(&sc)[4]) {
        if constexpr (WBITS == 4) {
            // chunk r of the sub-block may lie past the row end in the last one: clamp into the row (its activations are zero)
            const size_t off = min(static_cast<size_t>(blk) * 256 + 16 * r, row_bytes - 16) - 16 * r;
#pragma unroll
            for (int u = 0; u < 4; ++u) dst[u] = load_nt(reinterpret_cast<const uint4_t *>(wp + wrow_off[u] + off));
            const int g0 = min(blk * 4, sgroups - 4);  // 4 consecutive groups, clamped in-bounds (sgroups >= 4)
#pragma unroll
            for (int e = 0; e < 4; ++e) {  // K % 256 == 0 -> sgroups and g0 even -> 4-byte aligned pairs
                const unsigned int *sp = reinterpret_cast<const unsigned int *>(gscale + static_cast<size_t>(min(n0 + 4 * q + e, N - 1)) * sgroups + g0);
                sc[e] = uint2{sp[0], sp[1]};
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) dst[u] = load_nt(reinterpret_cast<const uint4_t *>(wp + wrow_off[u] + static_cast<size_t>(blk) * 256));
        }
    };
    unsigned char *wmine = &wsm[wave][0];
    // registers (row-contiguous image) -> LDS -> MFMA fragments; same-wave LDS ops execute in order, the fences only
    // stop the compiler from reordering them
    auto transpose_w = [&](const uint4_t(&src)[4], uint4_t(&frag)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = 4 * u + q;
            *reinterpret_cast<uint4_t *>(wmine + row * 256 + ((r ^ (row & 15)) << 4)) = src[u];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int u = 0; u < 4; ++u) frag[u] = *reinterpret_cast<const uint4_t *>(wmine + r * 256 + (((4 * u + q) ^ (r & 15)) << 4));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // group staging: chunk id -> (sub-block g, row, chunk); sub-blocks past the slice end re-read the last one (unused)
    auto load_xg = [&](int blk0, int last_blk) {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int id = min(tid + 256 * i, GCH - 1);
            const int g = id / (ROWS * CPR), rem = id - g * (ROWS * CPR), row = rem / CPR, ch = rem - row * CPR;
            const int blk = min(blk0 + g, last_blk);
            const size_t koff = static_cast<size_t>(blk) * RB + ch * 16;  // byte offset inside the activation row
            xr[i] = *reinterpret_cast<const uint4_t *>(x + static_cast<size_t>(min(row, M - 1)) * x_row_bytes + min(koff, x_row_bytes - 16));
            if (WBITS == 4 && koff >= x_row_bytes) xr[i] = uint4_t{0u, 0u, 0u, 0u};  // past K in the ragged last sub-block
        }
    };
    auto store_xg = [&]() {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int id = tid + 256 * i;
            if (id < GCH) {
                const int g = id / (ROWS * CPR), rem = id - g * (ROWS * CPR), row = rem / CPR, ch = rem - row * CPR;
                uint4_t v = xr[i];
                if constexpr (WBITS == 4) {  // k order of the de-quantised nibbles: (0,4,1,5,2,6,3,7)
                    const half8_t h = __builtin_bit_cast(half8_t, v);
                    v = __builtin_bit_cast(uint4_t, half8_t{h[0], h[4], h[1], h[5], h[2], h[6], h[3], h[7]});
                }
                *reinterpret_cast<uint4_t *>(&xs[g][row * RB + ((ch ^ (row & 15)) << 4)]) = v;
            }
        }
    };
    floatx4 acc[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](const uint4_t(&ab)[4], const uint2(&sc)[4], int buf, int blk) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (WBITS == 4) {
                // fragment u = 32 weights of row r in group (4 blk + u): word s -> MFMA k-step s
                floatx4 part[MT];
#pragma unroll
                for (int j = 0; j < MT; ++j) part[j] = floatx4{0.f, 0.f, 0.f, 0.f};
                const half2_t off8 = {static_cast<half_t>(1032.f), static_cast<half_t>(1032.f)};
                const half2_t sixteenth = {static_cast<half_t>(0.0625f), static_cast<half_t>(0.0625f)};
                const half2_t off72 = {static_cast<half_t>(72.f), static_cast<half_t>(72.f)};
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const unsigned int w0 = ab[u][st], w8 = w0 >> 8;
                    const half2_t h0 = as_half2((w0 & 0x000F000Fu) | 0x64006400u) - off8;                    // n0, n4
                    const half2_t h1 = as_half2((w0 & 0x00F000F0u) | 0x64006400u) * sixteenth - off72;      // n1, n5
                    const half2_t h2 = as_half2((w8 & 0x000F000Fu) | 0x64006400u) - off8;                    // n2, n6
                    const half2_t h3 = as_half2((w8 & 0x00F000F0u) | 0x64006400u) * sixteenth - off72;      // n3, n7
                    const half8_t af = {h0[0], h0[1], h1[0], h1[1], h2[0], h2[1], h3[0], h3[1]};
#pragma unroll
                    for (int j = 0; j < MT; ++j) {
                        const int row = 16 * j + r, ch = (u * 4 + q) * 4 + st;
                        const half8_t bf = *reinterpret_cast<const half8_t *>(&xs[buf][row * RB + ((ch ^ (row & 15)) << 4)]);
                        part[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, part[j], 0, 0, 0);
                    }
                }
                // D rows n0 + 4q + e: scale of (row, group); groups clamped at the matrix end were multiplied with zeros
                const int gi = blk * 4 + u - min(blk * 4, sgroups - 4);  // position inside the 4 loaded scales (0..3, or past)
                float sv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const half2_t lo = as_half2(sc[e].x), hi = as_half2(sc[e].y);
                    const half_t hsel = gi == 0 ? lo[0] : (gi == 1 ? lo[1] : (gi == 2 ? hi[0] : hi[1]));
                    sv[e] = gi < 4 ? to_f32(hsel) : 0.f;
                }
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(part[j][e], sv[e], acc[j][e]);
            } else if constexpr (WBITS == 16) {
                const half8_t af = __builtin_bit_cast(half8_t, ab[u]);
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const int row = 16 * j + r, ch = u * 4 + q;
                    const half8_t bf = *reinterpret_cast<const half8_t *>(&xs[buf][row * RB + ((ch ^ (row & 15)) << 4)]);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc[j], 0, 0, 0);
                }
            } else if constexpr (FP8) {
                // the lane's 16 consecutive k of this 64-wide step: two fp8 MFMA operands, no conversion
                const long a0 = (static_cast<long>(ab[u][1]) << 32) | ab[u][0];
                const long a1 = (static_cast<long>(ab[u][3]) << 32) | ab[u][2];
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const int row = 16 * j + r, ch = u * 4 + q;
                    const uint4_t bv = *reinterpret_cast<const uint4_t *>(&xs[buf][row * RB + ((ch ^ (row & 15)) << 4)]);
                    const long b0 = (static_cast<long>(bv[1]) << 32) | bv[0];
                    const long b1 = (static_cast<long>(bv[3]) << 32) | bv[2];
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, b0, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a1, b1, acc[j], 0, 0, 0);
                }
            } else {
                const half8_t a0 = dequant_i8x8(ab[u][0], ab[u][1]), a1 = dequant_i8x8(ab[u][2], ab[u][3]);
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const int row = 16 * j + r, ch = u * 8 + 2 * q;  // 16 consecutive k of this lane = chunks ch, ch+1
                    const half8_t b0 = *reinterpret_cast<const half8_t *>(&xs[buf][row * RB + ((ch ^ (row & 15)) << 4)]);
                    const half8_t b1 = *reinterpret_cast<const half8_t *>(&xs[buf][row * RB + (((ch + 1) ^ (row & 15)) << 4)]);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc[j], 0, 0, 0);
                }
            }
        }
    };
    const int nb = max(0, b_end - b_begin);
    if (nb > 0) load_xg(b_begin, b_end - 1);  // activations first (L2), then the weight ring (HBM)
#pragma unroll
    for (int i = 0; i < R; ++i)
        if (i < nb) load_a(b_begin + i, a[i], gsc[WBITS == 4 ? i : 0]);
    static_assert(U % R == 0 && U % G == 0, "unroll vs ring depth / group size");
    for (int u0 = 0; u0 < nb; u0 += U) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int b = u0 + i;  // wave-uniform
            if (b < nb) {
                if (i % G == 0) {  // group boundary: publish the staged activations, start fetching the next group
                    if (b) __syncthreads();  // previous group consumed by every wave
                    store_xg();
                    __syncthreads();
                    if (b + G < nb) load_xg(b_begin + b + G, b_end - 1);
                }
                uint4_t frag[4];
                uint2 scur[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) scur[e] = gsc[WBITS == 4 ? i % R : 0][e];
                transpose_w(a[i % R], frag);
                if (b + R < nb) load_a(b_begin + b + R, a[i % R], gsc[WBITS == 4 ? i % R : 0]);  // ring slot free again
                compute(frag, scur, i % G, b_begin + b);
            }
        }
    }
    // rows past M were computed on clamped (duplicate) activations: never stored
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = 16 * j + r, n = n0 + 4 * q;
        if (m < M && n < N) {
            float *dst = slab + (static_cast<size_t>(ks) * M + m) * N + n;
            if (n + 3 < N && (N & 3) == 0) {
                *reinterpret_cast<floatx4 *>(dst) = acc[j];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < N) dst[e] = acc[j][e];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Tiled MFMA GEMM (prefill / large batches):  C[M,N] = A[M,K] . B[N,K]^T, fp16 in, fp32 accumulate.
// Both operands are K-contiguous, so every MFMA fragment is one 16-byte LDS read.
//   block tile 128 x 128, BK = 64, 256 threads = 4 waves as 2 x 2, each wave 64 x 64 = 4 x 4 tiles of
//   v_mfma_f32_16x16x32_f16 (64 accumulator registers);
//   LDS: A and B tiles of 128 rows x 128 B, 16-byte chunks XOR-swizzled with (row & 7) so a fragment read
//   (16 rows x one chunk column) is conflict-free; global -> registers -> LDS with the next k-tile's global
//   loads issued before the MFMAs of the current one (register double buffering, two barriers per k-tile).
// Batched through blockIdx.z with dense strides (QK^T of the prefill attention).  MFMA-bound.
// ------------------------------------------------------------------------------------------
template <bool HAS_EPI>
__global__ __launch_bounds__(256) void tiled_mfma_f16_kernel(const half_t *__restrict__ A, const half_t *__restrict__ B,
                                                             half_t *C, int M, int N, int K, size_t strideA,
                                                             size_t strideB, size_t strideC,
                                                             const half_t *__restrict__ bias, const half_t *residual) {
    constexpr int BM = 128, BN = 128, BK = 64;
    __shared__ __attribute__((aligned(16))) half_t As[BM * BK];
    __shared__ __attribute__((aligned(16))) half_t Bs[BN * BK];
    A += blockIdx.z * strideA;
    B += blockIdx.z * strideB;
    C += blockIdx.z * strideC;
    if (residual) residual += blockIdx.z * strideC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // 2 x 2 waves
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int r = lane & 15, q = lane >> 4;

    // staging: a tile is 128 rows x 8 chunks of 16 B = 1024 chunks, 4 per thread: chunk id = tid + 256*i
    half8_t ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + 256 * i, row = id >> 3, ch = id & 7;
            const int am = min(m0 + row, M - 1), bn = min(n0 + row, N - 1);  // clamp: edge rows are never stored
            ra[i] = *reinterpret_cast<const half8_t *>(A + static_cast<size_t>(am) * K + k0 + ch * 8);
            rb[i] = *reinterpret_cast<const half8_t *>(B + static_cast<size_t>(bn) * K + k0 + ch * 8);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + 256 * i, row = id >> 3, ch = id & 7;
            const int off = row * BK + ((ch ^ (row & 7)) << 3);
            *reinterpret_cast<half8_t *>(As + off) = ra[i];
            *reinterpret_cast<half8_t *>(Bs + off) = rb[i];
        }
    };
    floatx4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();  // previous tile fully consumed
        lstore();
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);  // in flight under the MFMAs below
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {  // two 32-wide k-steps per tile
            half8_t af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + i * 16 + r;
                af[i] = *reinterpret_cast<const half8_t *>(As + row * BK + (((ks * 4 + q) ^ (row & 7)) << 3));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wn * 64 + j * 16 + r;
                bf[j] = *reinterpret_cast<const half8_t *>(Bs + row * BK + (((ks * 4 + q) ^ (row & 7)) << 3));
            }
            // D[n-row, m-col] convention: weights/B as the MFMA A operand gives 4 consecutive n per lane (8-byte stores)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
    // acc[i][j]: lane holds C[m = m0 + wm*64 + i*16 + (lane&15)][n = n0 + wn*64 + j*16 + 4*(lane>>4) + e]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + r;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + 4 * q;
            if (n + 3 < N && !HAS_EPI && (N & 3) == 0) {
                half4_t o = {from_f32<half_t>(acc[i][j][0]), from_f32<half_t>(acc[i][j][1]), from_f32<half_t>(acc[i][j][2]),
                             from_f32<half_t>(acc[i][j][3])};
                *reinterpret_cast<half4_t *>(C + static_cast<size_t>(m) * N + n) = o;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < N) {
                        float v = acc[i][j][e];
                        if (HAS_EPI) {
                            if (bias) v += to_f32(bias[n + e]);
                            if (residual) v += to_f32(residual[static_cast<size_t>(m) * N + n + e]);
                        }
                        C[static_cast<size_t>(m) * N + n + e] = from_f32<half_t>(v);
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Generic fallback: C[M,N] = A[M,K] * op(B), any T in {float, half}, fp32 accumulate,
// 64x64 block tile, 16-deep k tile, 256 threads each owning a 4x4 micro tile.
// Batched via blockIdx.z with dense strides.  B is [N,K] if TRANS_B else [K,N].
// ------------------------------------------------------------------------------------------
template <typename T, bool TRANS_B>
__global__ __launch_bounds__(256) void generic_gemm_kernel(const T *__restrict__ A, const T *__restrict__ B,
                                                           T *__restrict__ C, int M, int N, int K,
                                                           size_t strideA, size_t strideB, size_t strideC,
                                                           const T *__restrict__ bias,
                                                           const T *residual) {
    __shared__ float As[16][64 + 1];
    __shared__ float Bs[16][64 + 1];
    A += blockIdx.z * strideA;
    B += blockIdx.z * strideB;
    C += blockIdx.z * strideC;
    if (residual) residual += blockIdx.z * strideC;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        // A tile: 64 rows x 16 k ; B tile: 16 k x 64 cols
        for (int i = threadIdx.x; i < 64 * 16; i += 256) {
            const int mm = i >> 4, kk = i & 15;
            const int gm = m0 + mm, gk = k0 + kk;
            As[kk][mm] = (gm < M && gk < K) ? to_f32(A[static_cast<size_t>(gm) * K + gk]) : 0.f;
        }
        for (int i = threadIdx.x; i < 64 * 16; i += 256) {
            int nn, kk;
            if constexpr (TRANS_B) {
                nn = i >> 4;
                kk = i & 15;
            } else {
                kk = i >> 6;
                nn = i & 63;
            }
            const int gn = n0 + nn, gk = k0 + kk;
            float v = 0.f;
            if (gn < N && gk < K)
                v = TRANS_B ? to_f32(B[static_cast<size_t>(gn) * K + gk]) : to_f32(B[static_cast<size_t>(gk) * N + gn]);
            Bs[kk][nn] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gm = m0 + ty * 4 + i;
        if (gm >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gn = n0 + tx * 4 + j;
            if (gn >= N) continue;
            float v = acc[i][j];
            if (bias) v += to_f32(bias[gn]);
            if (residual) v += to_f32(residual[static_cast<size_t>(gm) * N + gn]);
            C[static_cast<size_t>(gm) * N + gn] = from_f32<T>(v);
        }
    }
}

}  // namespace llmie

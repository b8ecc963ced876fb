// Split-K MFMA GEMM for 32 < M <= 128 activation rows (decode batches of 33..128 sequences, short prefills; XR = 64 or 128
// staged activation rows):
//   slab[ks][m][n] = sum_{k in slice ks} X[m,k] * W[n,k]        (fp32 partials, consumed like skinny_splitk_kernel's)
// -- the linear of launchLinearGemm (linear.cu:10-87) where the weight stream still bounds the op (128 flop per weight
// byte) but the 64-row skinny kernel re-stages the 128-row activation tile through VGPRs once per 64 weight rows (L2
// activation traffic = 2x the weight stream, one barrier pair per 128 k: measured 2.2-2.4 TB/s at M = 128).
// Here a 512-thread workgroup multiplies ALL (<= 128) activation rows with 64*WN weight rows over one K slice:
//   * both operands go HBM/L2 -> LDS by global_load_lds_dwordx4 (no VGPR staging, no ds_write); the XOR swizzle that makes
//     fragment reads conflict-free is applied on the source address (as in gemm256.cuh);
//   * NS-stage ring with COUNTED waits: tile kt+NS-1 is issued while tile kt is multiplied, s_waitcnt vmcnt leaves the
//     younger tiles in flight across the one barrier per k-tile, so NS-1 tiles (2 x 32 KiB of weights at WN = 4) are in
//     flight per CU -- what a bandwidth-bound stream needs;
//   * 8 waves as 2 (M) x 4 (N), wave tile 64 x 16*WN: L2 activation traffic = |X| per 64*WN weight rows (0.5x the weight stream).
// LDS stage: X (XR rows x 128 B) | W (64*WN rows x 128 B).  WN = 2, 3 or 4 (128 / 192 / 256 weight rows): the width is chosen so
// that tiles x K slices come close to the 256 CUs (N = 22016: 115 tiles of 192 rows x 2 slices = 230 workgroups, where 256-row tiles
// give 172 or 258; N = 12288: 64 x 4 = 256 with one slab less than 48 x 5).
#pragma once
#include "device_utils.cuh"

namespace llmie {

typedef int mid_intx8 __attribute__((ext_vector_type(8)));

#ifdef MID_STAMPS   // diagnostic build only (tools/micro/mid_probe.hip): where a workgroup's time goes; never defined in the product build
__device__ unsigned long long mid_stamp_buf[1024 * 8];
#define MID_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) mid_stamp_buf[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MID_STAMP(i) do { } while (0)
#endif

// WQ = 8 (round 3): W is int8 [N, K] under fp16 activations (decode batches of 33..128 sequences and short prefills of int8 engines:
// the 64-row skinny kernel that served them streamed at 2.2-2.4 TB/s at 128 rows -- bench r03: int8 128-token prefill 18.1k tok/s
// against 27.1k for fp16).  The weight tile travels as RAW bytes (64 B per row and 64-deep k-tile: half the DMA bytes, 16 rows per
// wave instruction), a lane fetches its 8 weights of a k-step with one ds_read_b64 and de-quantises them in registers
// (device_utils.cuh, g8_*); the slabs hold unscaled sums, the consumers apply the row scales as for the skinny kernel's slabs.
template <bool FP8, int WN, int NS, int XR = 128, int WQ = 0>
__global__ __launch_bounds__(512) void mid_splitk_kernel(const void *__restrict__ Xv, const void *__restrict__ Wv,
                                                         float *__restrict__ slab, int M, int N, int K, int KS, int kt_per_slice) {
    static_assert(WQ == 0 || (WQ == 8 && !FP8), "int8 weights under fp16 activations");
    constexpr int ES = FP8 ? 1 : 2;   // bytes per element
    constexpr int BK = 128 / ES;      // k per tile: rows of 128 bytes either way
    // XR = activation rows staged per k-tile: 128, or 64 for M <= 64 (half the X tile: a 4th stage fits the 160 KiB and
    // three k-tiles of weights stay in flight per CU)
    constexpr int WROW = WQ ? 64 : 128;   // bytes of a weight row per k-tile
    constexpr int X_INSTR = XR / 64;  // LDS-DMA instructions per wave for the X tile (8 rows each)
    // ... for the W tile: 8 rows of 128 B per instruction, or (int8) 16 rows of 64 B -- 4 WN pieces over the 8 waves; at WN = 3 every
    // wave still issues two (the counted waits are per-wave immediates): the last four pieces re-fetch clamped rows into padding
    constexpr int W_INSTR = WQ ? (4 * WN + 7) / 8 : WN;
    constexpr int X_BYTES = XR * 128, W_BYTES = WQ ? W_INSTR * 8 * 1024 : 64 * WN * WROW, STAGE_BYTES = X_BYTES + W_BYTES;
    constexpr int IPT = X_INSTR + W_INSTR; // LDS-DMA instructions per wave per k-tile
    constexpr int MI = XR / 32;       // 16-row activation tiles per wave (wave grid 2 x 4)
    static_assert(XR == 128 || XR == 64, "activation rows per stage");
    static_assert(WN >= 2 && WN <= 4, "128, 192 or 256 weight rows per workgroup");
    static_assert(NS >= 2 && (NS - 2) * IPT < 64, "vmcnt is a 6-bit counter");
    const unsigned char *X = static_cast<const unsigned char *>(Xv), *W = static_cast<const unsigned char *>(Wv);
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int r = lane & 15, q = lane >> 4;
    const int tile = blockIdx.x / KS, ks = blockIdx.x - tile * KS;
    const int n0 = tile * 64 * WN;
    const int KT = K / BK;
    const int kt0 = ks * kt_per_slice, nk = min(KT, kt0 + kt_per_slice) - kt0;  // host: every slice has >= 1 k-tile

    // DMA plan (gemm256.cuh): half h, instruction i: this wave fills rows (i*8 + wave)*8 .. +8; lane -> row + lane/8,
    // LDS slot lane%8 receives source chunk slot ^ (row & 7)
    // X: instruction i (< X_INSTR) fills X rows (i*8 + wave)*8 .. +8; W: instruction i (< WN) fills W rows (i*8 + wave)*8 .. +8
    const unsigned char *xsrc[X_INSTR], *wsrc[W_INSTR];
#pragma unroll
    for (int i = 0; i < (X_INSTR > W_INSTR ? X_INSTR : W_INSTR); ++i) {
        const int row = (i * 8 + wave) * 8 + (lane >> 3), slot = lane & 7, chunk = slot ^ (row & 7);
        if (i < X_INSTR)   // clamped rows are never stored
            xsrc[i] = X + (static_cast<size_t>(min(min(row, XR - 1), M - 1)) * K + static_cast<size_t>(kt0) * BK) * ES + chunk * 16;
        if constexpr (WQ != 0) {   // int8: 16 rows x 64 B per instruction; slot c of row r <- source chunk c ^ ((r >> 2) & 3)
            if (i < W_INSTR) {
                const int wrow = min((i * 8 + wave) * 16 + (lane >> 2), 64 * WN - 1);   // (WN = 3: pieces 12..15 are padding)
                wsrc[i] = W + static_cast<size_t>(min(n0 + wrow, N - 1)) * K + static_cast<size_t>(kt0) * 64 + g8_q8_slot(wrow, lane & 3) * 16;
            }
        } else {
            if (i < W_INSTR) wsrc[i] = W + (static_cast<size_t>(min(n0 + row, N - 1)) * K + static_cast<size_t>(kt0) * BK) * ES + chunk * 16;
        }
    }
    // Workgroups of one K slice run in lock step and a k-tile of 128-byte row pieces at an 8 KiB row pitch lands in ONE L2
    // channel: every workgroup starts its slice at a different k-tile (sum order is irrelevant: fp32 partials), so that the
    // concurrent tiles spread over the channels.
    const int rot = nk > 0 ? (tile * 3 + ks) % nk : 0;
    auto dma_tile = [&](int tt, int stage) {
        const int t = tt + rot < nk ? tt + rot : tt + rot - nk;
        typedef const __attribute__((address_space(1))) void *gptr_t;
        typedef __attribute__((address_space(3))) void *lptr_t;
#pragma unroll
        for (int i = 0; i < X_INSTR; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(xsrc[i] + static_cast<size_t>(t) * 128), (lptr_t)(lds + stage * STAGE_BYTES + (i * 8 + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < W_INSTR; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + static_cast<size_t>(t) * WROW), (lptr_t)(lds + stage * STAGE_BYTES + X_BYTES + (i * 8 + wave) * 1024), 16, 0, 0);
    };

    floatx4 acc[MI][WN];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int wcol = wc * 16 * WN;  // first weight row of this wave inside the workgroup tile
    const int a_row0 = wr * (XR / 2) + r, b_row0 = wcol + r;   // (b_row0 + 16 j) & 7 == r & 7: the swizzle term is per lane
    const unsigned char *a_base = lds, *b_base = lds + X_BYTES;
    auto frag = [&](const unsigned char *base, int row, int c) {
        return *reinterpret_cast<const half8_t *>(base + row * 128 + ((c ^ (row & 7)) << 4));
    };
    auto frag8 = [&](const unsigned char *base, int row) {  // fp8: lane (r, q) supplies k bytes [16q, +16) and [64 + 16q, +16) of its row (chunks q, 4 + q: conflict-free under the row & 7 swizzle; round 2 read chunks 2q, 2q + 1: a 2-way bank conflict on every ds_read_b128)
        const uint4_t lo = *reinterpret_cast<const uint4_t *>(base + row * 128 + ((q ^ (row & 7)) << 4));
        const uint4_t hi = *reinterpret_cast<const uint4_t *>(base + row * 128 + (((4 + q) ^ (row & 7)) << 4));
        return mid_intx8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    };

    MID_STAMP(0);
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nk) dma_tile(t, t);
    MID_STAMP(1);
    int stage = 0, fill = NS - 1;  // fill = stage that receives tile kt + NS - 1
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed when at most the (NS - 2) younger tiles of this wave are outstanding
        if (kt + NS - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * IPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave's part of tile kt landed; every wave is done with stage `fill` (tile kt - 1)
        if (kt == 0) MID_STAMP(2);
        if (kt == nk / 2) MID_STAMP(3);
        if (kt + NS - 1 < nk) dma_tile(kt + NS - 1, fill);
        const unsigned char *ab = a_base + stage * STAGE_BYTES, *bb = b_base + stage * STAGE_BYTES;
        if constexpr (FP8) {
            mid_intx8 bf[WN], af[MI];
#pragma unroll
            for (int j = 0; j < WN; ++j) bf[j] = frag8(bb, b_row0 + j * 16);
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = frag8(ab, a_row0 + i * 16);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[j], af[i], acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0,
                                                                                 0x7F7F7F7F);
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                half8_t bf[WN], af[MI];
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    if constexpr (WQ != 0)   // (b_row0 + 16 j) >> 2 & 3 == r >> 2 & 3: the piece offset is per lane
                        bf[j] = g8_dequant8(*reinterpret_cast<const uint2 *>(bb + (b_row0 + j * 16) * 64 + g8_q8_piece(r, q, s)));
                    else
                        bf[j] = frag(bb, b_row0 + j * 16, s * 4 + q);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = frag(ab, a_row0 + i * 16, s * 4 + q);
                // D[n-row, m-col]: W as the MFMA A operand -> 4 consecutive n per lane (16-byte fp32 stores)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
            }
        }
        stage = stage + 1 == NS ? 0 : stage + 1;
        fill = fill + 1 == NS ? 0 : fill + 1;
    }
    MID_STAMP(4);
    // acc[i][j]: lane holds rows m = wr*(XR/2) + i*16 + r, columns n0 + wcol + j*16 + 4q + e
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = wr * (XR / 2) + i * 16 + r;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = n0 + wcol + j * 16 + 4 * q;
            float *dst = slab + (static_cast<size_t>(ks) * M + m) * N + n;
            if (n + 3 < N && (N & 3) == 0) {
                *reinterpret_cast<floatx4 *>(dst) = acc[i][j];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < N) dst[e] = acc[i][j][e];
            }
        }
    }
    MID_STAMP(5);
}

}  // namespace llmie
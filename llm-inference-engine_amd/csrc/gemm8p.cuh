// 256 x 256 x 64 MFMA GEMM, eight-phase ping-pong schedule (gfx950):  C[M,N] = X[M,K] . W[N,K]^T (+bias)(+residual) or the fused
// SwiGLU form -- the same operation, operands, tile geometry, LDS image and epilogue as gemm256_kernel<.., WN = 4> (gemm256.cuh: the
// linear of launchLinearGemm, linear.cu:10-87, at prefill sizes), with a different main loop.
//
// gemm256_kernel runs its eight waves in lock step: every wave reads its 12 fragments of a k-tile, multiplies, and all meet at one
// barrier per k-tile that also drains the LDS-DMA of the next tile (vmcnt(0)); its MFMA pipes are busy 51 % of the time
// (profiles/r02_gemm256_pmc.csv).  Here
//   * the two wave rows (wr = 0 / 1: one wave of each on every SIMD) run half a phase apart: while one group issues its 16 MFMAs of
//     a phase (one 64 x 32 quadrant of the wave's 128 x 64 tile over the 64-deep k-tile) the other reads the fragments of its next
//     quadrant and issues its share of the LDS-DMA, and the two swap at every barrier -- the MFMA pipe of a SIMD always has a
//     wave to serve and the LDS reads of one group hide under the arithmetic of the other;
//   * a k-tile is four phases (quadrants (0,0) (0,1) (1,1) (1,0)); a phase reads ONE operand piece -- A rows 0-63, B columns 32-63,
//     A rows 64-127, B columns 0-31 of the NEXT k-tile (8 / 4 / 8 / 4 ds_read_b128 per wave) -- so only the piece that changes
//     between two consecutive quadrants is fetched, and the B piece of the first quadrant is already in registers when a tile starts;
//   * the LDS-DMA runs in the same four pieces ("groups" of 16 KiB = 2 wave instructions per wave, filled by all eight waves), one
//     group per phase, five groups (80 KiB) in flight: phase p reads group p + 1, waits until group p + 2 has landed (a counted
//     s_waitcnt vmcnt(10), never 0 in the steady state) and issues group p + 7 into the slot that group p - 1 occupied;
//   * the DMA is issued from inline asm and counted by hand (hipcc orders LDS accesses of its own behind a pending builtin DMA with
//     vmcnt(0)); raw s_barrier, no fence.
// Ordering rules this schedule is built on (MI355X guide, "Read a staged buffer one phase AFTER the wait that retires it"): with
// the groups staggered by one barrier, a group waited for in phase p (before the phase's first barrier) is visible to every wave's
// reads from phase p + 1; a slot may be refilled two phases after its last read (the reads are retired by the lgkmcnt(0) behind the
// reading phase's first barrier).  Both hold with equality here: read p + 1 / retire p + 2 / refill (p - 1)'s slot, whose last
// read was phase p - 2.
#pragma once
#include "gemm256.cuh"

namespace llmie {

__device__ __forceinline__ void g8_dma16(const unsigned voff, const void *sbase, const unsigned lds_dst) {
    // 1 KiB per wave: 16 bytes per lane from sbase + voff -> LDS lds_dst + 16 lane (M0 written in the statement that uses it).
    // hipcc pads no hazard inside an asm string: "VALU writes SGPR -> VMEM reads it" needs 5 wait states, which matters only if
    // the base pair were produced by a VALU instruction right in front of the statement -- an SGPR restored from a spill lane
    // with v_readlane (pk_gemm.cuh met exactly that).  Here the base comes out of scalar adds, and tests/test_abi_cpu.py refuses
    // any instantiation that spills SGPRs; a leading `s_nop 4` as in pk_gemm.cuh was measured and costs 2-13 % of the kernel
    // (the statement sits in the issue stream of the reading group of every phase), so the guard test stands in for it.
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// one operand fragment of a k-tile: fp16 -> the two 32-deep k-steps, e4m3 -> one 128-deep step (8 registers the MFMA takes whole)
template <bool FP8> struct G8Frag;
template <> struct G8Frag<false> { half8_t k[2]; };
template <> struct G8Frag<true> { intx8 v; };
template <bool FP8> __device__ __forceinline__ void g8_read(G8Frag<FP8> &f, const unsigned char *p, unsigned sw0, unsigned sw1) {
    const uint4_t lo = *reinterpret_cast<const uint4_t *>(p + sw0), hi = *reinterpret_cast<const uint4_t *>(p + sw1);
    if constexpr (FP8) {
        f.v = intx8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    } else {
        f.k[0] = __builtin_bit_cast(half8_t, lo);
        f.k[1] = __builtin_bit_cast(half8_t, hi);
    }
}
template <bool FP8> __device__ __forceinline__ void g8_mma(floatx4 &acc, const G8Frag<FP8> &w, const G8Frag<FP8> &x, int k) {
    if constexpr (FP8) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w.v, x.v, acc, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    else acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.k[k], x.k[k], acc, 0, 0, 0);
}
// ---- weight-only int8 B operand (WQ = 8; round 3: prefill of int8 engines, context_decoder.cpp:58-199 on int8 [N, K] weights) ----
// The weight tile travels HBM/L2 -> LDS as RAW int8 (64 bytes per row and 64-deep k-tile: half the DMA and LDS-fill bytes of the
// fp16 form) and is de-quantised when a wave reads its fragment.  Lane (r, q) multiplies, in k-step s, the k = 32 s + 8 q .. + 7 of
// the k-tile -- the assignment of the fp16 form, so the activation fragments are read exactly as there (16-byte chunks q and 4 + q
// of the row under the row & 7 swizzle: conflict-free; the "k = 16 q .. 16 q + 15 per lane" assignment that a single 16-byte weight
// read per lane would want costs every ACTIVATION read a 2-way bank conflict under the same swizzle: profiles/r03, 0.44 conflict
// cycles per LDS cycle, MFMA busy 52 % against 67 %) -- and fetches its 8 weights of a k-step with one ds_read_b64: bytes
// [32 s + 8 q, + 8) of the row.  De-quantisation is exact: byte ^ 0x80 under the fp16 exponent 0x64 is 1152 + w, and a packed fp16
// subtract of 1152 leaves w (integers below 2048 are exact in fp16); the per-row scale meets the fp32 accumulator in the epilogue,
// so the numerics are "fp16 activations x integer weights, fp32 accumulate, one scale, one rounding" -- what the int8 decode
// kernels compute.  LDS image of a B half: 128 rows x 64 B; the 16-byte slot c of row r holds source chunk c ^ ((r >> 2) & 3)
// (applied on the DMA source address): the 32 lanes of a ds_read_b64 half-wave (16 rows x 2 values of q) then hit 32 different
// 8-byte bank pairs.
// p = the lane's row in the LDS image; off0 / off1 = g8_q8_piece of the two k-steps
__device__ __forceinline__ void g8_read_q8(G8Frag<false> &f, const unsigned char *p, unsigned off0, unsigned off1) {
    const uint2 w0 = *reinterpret_cast<const uint2 *>(p + off0), w1 = *reinterpret_cast<const uint2 *>(p + off1);
    f.k[0] = g8_dequant8(w0);
    f.k[1] = g8_dequant8(w1);
}
// s_waitcnt vmcnt(n) for a run-time n (tails and prologues only; the steady state uses immediates)
__device__ __forceinline__ void g8_wait_vm(int n) {
    switch (n < 0 ? 0 : (n > 10 ? 10 : n)) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    }
}

// Placement pins (no instructions): the MFMAs of a phase read fragments that pass through g8_pin AFTER the phase's lgkmcnt(0) and
// write accumulators that pass through g8_pin BEFORE the closing barrier, so no IR pass can sink them out of the phase or hoist
// them into the read segment (sched_barrier only binds the machine scheduler; without the pins the e4m3 form had its MFMAs sunk
// several phases down and spilled the fragments they kept alive).
__device__ __forceinline__ void g8_pin(floatx4 &a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void g8_pin(G8Frag<false> &f) { asm volatile("" : "+v"(f.k[0]), "+v"(f.k[1])); }
__device__ __forceinline__ void g8_pin(G8Frag<true> &f) { asm volatile("" : "+v"(f.v)); }
// wait until all but the newest `groups` DMA groups (2 wave instructions each) of this wave have landed
__device__ __forceinline__ void g8_wait_groups(int groups) {
    if (groups >= 5) {   // the steady state: one compare
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        return;
    }
    if (groups == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (groups == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (groups == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (groups == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// WQ = 8: W is int8 [N, K] (row pitch K bytes) and wscale points at the fp16 per-row scales; X stays fp16 (FP8 must be false).
// DMA groups of the weight pieces are then ONE wave instruction per wave (8 KiB) instead of two, so the counted waits differ per
// phase: behind phase (u, s) the groups 4u+s+3 .. 4u+s+7 stay in flight = A B A B A (8 instructions) for even s, B A B A B (7) for odd.
// ROPE (round 3): the QKV projection of a prefill with RoPE + KV-cache append as its epilogue (g256_store_qkv_rope, gemm256.cuh):
// the weight rows of a tile are fetched in the permuted order qkv_rope_col() so that a lane holds both halves of every rotate-half
// pair; col0 = first output column of this launch's range (a multiple of 128), C = the packed QKV buffer (un-offset), ldc its row
// pitch; K / V columns go to the caches only.  `rap` points at the DEVICE-resident, layer-invariant operands of the epilogue (read
// after the main loop: as by-value kernel arguments they sat in ~25 SGPRs through the loop and the int8 forms spilled scalars,
// which the hand-issued DMA cannot tolerate), `bias` = this layer's QKV bias or null, `layer` selects the cache slab.
template <bool FP8, bool HAS_EPI, bool SWIGLU = false, int WQ = 0, bool ROPE = false>
__global__ __launch_bounds__(512) void gemm8p_kernel(const void *__restrict__ Xv, const void *__restrict__ Wv, half_t *C, int M, int N,
                                                     int K, const half_t *__restrict__ bias, const half_t *residual, int tiles_n,
                                                     const float *__restrict__ xscale, const float *__restrict__ wscale,
                                                     int ldc_arg = 0, int group_m = 0, int col0 = 0, const QkvRopeArgs *__restrict__ rap = nullptr, int layer = 0) {
    static_assert(!ROPE || (!HAS_EPI && !SWIGLU), "ROPE form: the plain projection");
    // col0 (SwiGLU form): first output column of this launch's range (a launch over columns [col0, col0 + tiles_n * 128) of C[M, I])
    static_assert(!SWIGLU || !HAS_EPI, "SwiGLU form: no bias / residual");
    static_assert(WQ == 0 || (WQ == 8 && !FP8), "WQ: 0 (operands as they are) or 8 (int8 weights under fp16 activations)");
    const size_t ldc = ldc_arg ? ldc_arg : N;
    constexpr int ES = FP8 ? 1 : 2, BK = 128 / ES;
    constexpr int BN = SWIGLU ? 128 : 256;
    const int half_n = N >> 1;
    constexpr unsigned HALF_BYTES = 128 * 128, STAGE_BYTES = 4 * HALF_BYTES;   // stage: A half 0 | A half 1 | B half 0 | B half 1
    const unsigned char *X = static_cast<const unsigned char *>(Xv), *W = static_cast<const unsigned char *>(Wv);
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];  // [2 stages][4 halves][128 rows x 128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r = lane & 15, q = lane >> 4;
    int tile_m, tile_n;
    if (group_m > 0) {   // XCD-aware tile order, as in gemm256_kernel
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int qd = nwg >> 3, rem = nwg & 7;
        const int L = xcd * qd + min(xcd, rem) + idx;
        const int tm = (M + 255) / 256, per_group = group_m * tiles_n;
        const int grp = L / per_group, in = L - grp * per_group;
        const int first_m = grp * group_m, gsz = min(tm - first_m, group_m);
        tile_m = first_m + in % gsz;
        tile_n = in / gsz;
    } else {
        tile_m = blockIdx.x / tiles_n;
        tile_n = blockIdx.x - tile_m * tiles_n;
    }
    const int m0 = tile_m * 256, n0 = (SWIGLU ? col0 : 0) + tile_n * BN;

    // ---- DMA plan.  Group kinds: 0 = A rows 0-63 of both halves, 1 = B first column piece, 2 = B second column piece,
    //      3 = A rows 64-127.  A group = 16 pieces of 8 rows x 128 B; this wave moves pieces pi = i*8 + wave, i = 0, 1.
    //      lane -> row lane/8 of the piece, LDS slot lane%8 <- source chunk slot ^ (row & 7)  (swizzle on the source side)
    const size_t row_bytes = static_cast<size_t>(K) * ES;
    const size_t wrow_bytes = WQ ? static_cast<size_t>(K) : row_bytes;
    constexpr unsigned WKT_BYTES = WQ ? 64 : 128;   // bytes of a weight row per k-tile
    const unsigned char *xbase = X + static_cast<size_t>(m0) * row_bytes;   // wave-uniform bases; the per-lane part is 32-bit
    const unsigned char *wbase = W + static_cast<size_t>(n0) * wrow_bytes;
    // the two DMA bases live in scalar registers from here on: no VALU producer (v_readfirstlane / v_readlane of a spill) can sit
    // within the 5 wait states in front of a global_load_lds that reads them (g8_dma16 carries only `s_nop 0`)
    asm volatile("" : "+s"(xbase), "+s"(wbase));
    typedef __attribute__((address_space(3))) void *lptr_t;
    const unsigned lds_addr = static_cast<unsigned>(reinterpret_cast<size_t>((lptr_t)lds));   // LDS byte address of the image
    unsigned voff[4][2], ldst[4][2];
#pragma unroll
    for (int kind = 0; kind < 4; ++kind)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (WQ != 0) {
                if (kind == 1 || kind == 2) {   // int8 weight piece: 16 rows x 64 B per wave instruction, one per wave (i = 0)
                    const int pi = wave;
                    int half, row0;
                    if (SWIGLU) {
                        half = 2 + (kind - 1);
                        row0 = pi * 16;
                    } else {
                        half = 2 + (pi >> 2);
                        row0 = ((pi >> 1) & 1) * 64 + (kind - 1) * 32 + (pi & 1) * 16;
                    }
                    const int row = row0 + (lane >> 2), chunk = static_cast<int>(g8_q8_slot(row, lane & 3));
                    long grow;
                    if (SWIGLU) grow = static_cast<long>(half - 2) * half_n + min(n0 + row, half_n - 1) - n0;
                    else if (ROPE) grow = qkv_rope_col<4>((half - 2) * 128 + row);
                    else grow = min(n0 + (half - 2) * 128 + row, N - 1) - n0;
                    voff[kind][i] = static_cast<unsigned>(grow * static_cast<long>(wrow_bytes) + chunk * 16);
                    ldst[kind][i] = __builtin_amdgcn_readfirstlane(lds_addr + half * HALF_BYTES + row0 * 64);
                    continue;
                }
            }
            const int pi = i * 8 + wave;
            int half, row0;   // LDS half (0..3) and first row of the piece inside it
            if (kind == 0 || kind == 3) {
                half = pi >> 3;
                row0 = (pi & 7) * 8 + (kind == 3 ? 64 : 0);
            } else if (SWIGLU) {
                half = 2 + (kind - 1);   // gate half / up half, all 128 rows
                row0 = pi * 8;
            } else {
                half = 2 + (pi >> 3);
                row0 = ((pi >> 2) & 1) * 64 + (kind - 1) * 32 + (pi & 3) * 8;
            }
            const int row = row0 + (lane >> 3), chunk = (lane & 7) ^ (row & 7);
            long grow;   // row relative to the tile base, clamped to the matrix (edge rows are never stored)
            if (half < 2) grow = min(m0 + half * 128 + row, M - 1) - m0;
            else if (SWIGLU) grow = static_cast<long>(half - 2) * half_n + min(n0 + row, half_n - 1) - n0;
            else if (ROPE) grow = qkv_rope_col<4>((half - 2) * 128 + row);
            else grow = min(n0 + (half - 2) * 128 + row, N - 1) - n0;
            voff[kind][i] = static_cast<unsigned>(grow * static_cast<long>(row_bytes) + chunk * 16);
            ldst[kind][i] = __builtin_amdgcn_readfirstlane(lds_addr + half * HALF_BYTES + row0 * 128);
        }
    auto issue = [&](auto kind_, int kt, unsigned stage) {
        constexpr int kind = decltype(kind_)::value;
        constexpr bool is_a = kind == 0 || kind == 3;
        const unsigned char *base = is_a ? xbase + static_cast<size_t>(kt) * 128 : wbase + static_cast<size_t>(kt) * WKT_BYTES;
#pragma unroll
        for (int i = 0; i < ((WQ != 0 && !is_a) ? 1 : 2); ++i) g8_dma16(voff[kind][i], base, ldst[kind][i] + stage * STAGE_BYTES);
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;
    // WQ: DMA wave instructions of the groups (first, last] still in flight; group g is a weight piece (1) for even g, an
    // activation piece (2) for odd g (order {1, 0, 2, 3} inside a k-tile)
    auto q8_inflight = [](int first, int last) {
        int n = 0;
        for (int g = first + 1; g <= last; ++g) n += (g & 1) ? 2 : 1;
        return n;
    };

    floatx4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // ---- fragment reads: 16-byte piece k (0, 1) of row `row` = chunk k*4 + q (fp16: k-step k; e4m3: half of the one 128-deep step)
    const int wcol = SWIGLU ? wc * 32 : wc * 64;
    // every format reads its two 16-byte pieces of a row from chunks q and 4 + q: fp16 = the two 32-deep k-steps; e4m3 = 32 of the
    // 128 k of the one MFMA step (k = 16 q .. + 15 and 64 + 16 q ..: any k assignment is a valid contraction when both operands use
    // it, and this one is conflict-free under the row & 7 swizzle where chunks 2q, 2q + 1 -- the round-2 e4m3 assignment -- cost
    // every ds_read_b128 a 2-way bank conflict)
    const unsigned sw0 = static_cast<unsigned>((q ^ (r & 7)) << 4), sw1 = static_cast<unsigned>(((4 + q) ^ (r & 7)) << 4);
    const unsigned a_lane = wr * HALF_BYTES + r * 128;   // + stage, + (ih*64 + i*16) * 128, + sw
    // B piece jh, fragment jj: plain -> half 2 + (wc >> 1), row (wc & 1)*64 + jh*32 + jj*16 + r;  SwiGLU -> half 2 + jh, row wc*32 + jj*16 + r
    constexpr unsigned BROW = WQ ? 64 : 128;   // bytes of a weight row in the LDS image
    const unsigned b_lane = SWIGLU ? 2 * HALF_BYTES + (wc * 32 + r) * BROW : (2 + (wc >> 1)) * HALF_BYTES + ((wc & 1) * 64 + r) * BROW;
    const unsigned q8o0 = g8_q8_piece(r, q, 0), q8o1 = g8_q8_piece(r, q, 1);   // (WQ) this lane's two 8-byte weight pieces of a row
    constexpr unsigned B_PIECE = SWIGLU ? HALF_BYTES : 32 * BROW;
    G8Frag<FP8> fa[4], fb0[2][2], fb1[2];   // A piece (4 row tiles), B first piece of even / odd k-tiles, B second piece
    auto read_a = [&](unsigned stage, int ih) {
        const unsigned char *p = lds + stage * STAGE_BYTES + a_lane + ih * 64 * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) g8_read<FP8>(fa[i], p + i * 2048, sw0, sw1);
    };
    auto read_b = [&](unsigned stage, int jh, G8Frag<FP8> (&f)[2]) {
        const unsigned char *p = lds + stage * STAGE_BYTES + b_lane + jh * B_PIECE;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            if constexpr (WQ != 0) g8_read_q8(f[jj], p + jj * 16 * BROW, q8o0, q8o1);
            else g8_read<FP8>(f[jj], p + jj * 2048, sw0, sw1);
        }
    };
    auto quadrant = [&](auto ih_, auto jh_, G8Frag<FP8> (&f)[2]) {
        constexpr int ih = decltype(ih_)::value, jh = decltype(jh_)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) g8_pin(fa[i]);
        g8_pin(f[0]); g8_pin(f[1]);
#pragma unroll
        for (int k = 0; k < (FP8 ? 1 : 2); ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) g8_mma<FP8>(acc[ih * 4 + i][jh * 2 + jj], f[jj], fa[i], k);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) g8_pin(acc[ih * 4 + i][jh * 2 + jj]);
    };

    const int KT = K / BK, G = 4 * KT;   // k-tiles, DMA groups (group g: k-tile g / 4, kind {1, 0, 2, 3}[g % 4])
    // ---- prologue: groups 0..6 (k-tile 0 whole, k-tile 1 without its last piece)
    issue(K1{}, 0, 0); issue(K0{}, 0, 0); issue(K2{}, 0, 0); issue(K3{}, 0, 0);
    if (KT > 1) { issue(K1{}, 1, 1); issue(K0{}, 1, 1); issue(K2{}, 1, 1); }
    if constexpr (WQ != 0) g8_wait_vm(q8_inflight(1, min(7, G) - 1));
    else g8_wait_groups(min(7, G) - 2);   // groups 0 and 1 landed
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_b(0, 0, fb0[0]);
    if (wr == 1) __builtin_amdgcn_s_barrier();   // the second wave row runs one barrier behind from here on
    asm volatile("" ::: "memory");

    // phase S (0..7) of the two-k-tile loop body: k-tile u (parity S / 4), quadrant S % 4.  STEADY (a compile-time fact of the
    // call): at least four k-tiles are left, so every piece this phase reads or issues exists and five groups stay in flight -- no
    // tail tests and no wait selection in the instruction stream of the reading group, which is on the critical path of every
    // phase (ten extra scalar wait cycles per phase were measured at 2-13 % of the kernel).
    auto phase = [&](auto S_, int u, auto steady_) {
        constexpr int S = decltype(S_)::value, s = S & 3;
        constexpr unsigned par = S >> 2;
        constexpr bool STEADY = decltype(steady_)::value;
        // reads of this phase's new piece (group 4u + s + 1)
        if constexpr (s == 0) read_a(par, 0);
        else if constexpr (s == 1) read_b(par, 1, fb1);
        else if constexpr (s == 2) read_a(par, 1);
        else if (STEADY || u + 1 < KT) read_b(par ^ 1, 0, fb0[par ^ 1]);
        // DMA group 4u + s + 7
        if constexpr (s == 0) { if (STEADY || u + 1 < KT) issue(K3{}, u + 1, par ^ 1); }
        else if constexpr (s == 1) { if (STEADY || u + 2 < KT) issue(K1{}, u + 2, par); }
        else if constexpr (s == 2) { if (STEADY || u + 2 < KT) issue(K0{}, u + 2, par); }
        else { if (STEADY || u + 2 < KT) issue(K2{}, u + 2, par); }
        // group 4u + s + 2 landed: the next phase reads it
        if constexpr (WQ != 0) {
            if constexpr (STEADY) {
                if constexpr ((s & 1) == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            } else {
                g8_wait_vm(q8_inflight(4 * u + s + 2, min(4 * u + s + 7, G - 1)));
            }
        } else if constexpr (STEADY) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else g8_wait_groups(G - (4 * u + s) - 3);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (s == 0) quadrant(K0{}, K0{}, fb0[par]);
        else if constexpr (s == 1) quadrant(K0{}, K1{}, fb1);
        else if constexpr (s == 2) quadrant(K1{}, K1{}, fb1);
        else quadrant(K1{}, K0{}, fb0[par]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto body = [&](int u, auto steady_) {
        phase(std::integral_constant<int, 0>{}, u, steady_); phase(std::integral_constant<int, 1>{}, u, steady_);
        phase(std::integral_constant<int, 2>{}, u, steady_); phase(std::integral_constant<int, 3>{}, u, steady_);
        phase(std::integral_constant<int, 4>{}, u + 1, steady_); phase(std::integral_constant<int, 5>{}, u + 1, steady_);
        phase(std::integral_constant<int, 6>{}, u + 1, steady_); phase(std::integral_constant<int, 7>{}, u + 1, steady_);
    };
    int u = 0;
    // steady bodies: k-tiles u and u + 1 with KT - (u + 1) >= 3 (phase 4(u+1) + 3 still waits with five groups behind it: 4 KT -
    // (4(u+1) + 3) - 3 >= 5; its issue of k-tile u + 3 exists)
    if constexpr (!(FP8 && SWIGLU)) {   // (the e4m3 SwiGLU form with both bodies spills two registers: it keeps the generic body)
        for (; u + 3 < KT; u += 2) body(u, std::true_type{});
    }
    for (; u + 1 < KT; u += 2) body(u, std::false_type{});
    if (u < KT) {   // odd number of k-tiles: the last one has even parity
        phase(std::integral_constant<int, 0>{}, u, std::false_type{}); phase(std::integral_constant<int, 1>{}, u, std::false_type{});
        phase(std::integral_constant<int, 2>{}, u, std::false_type{}); phase(std::integral_constant<int, 3>{}, u, std::false_type{});
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();   // pairs with the second row's last barrier

    if constexpr (ROPE) {   // (C, xscale, wscale and the bias are whole-matrix pointers: the epilogue indexes them by absolute column)
        g256_store_qkv_rope<FP8, 4, WQ>(acc, C, M, ldc, m0, col0 + n0, wr, wcol, r, q, xscale, wscale, bias, rap, layer);
        return;
    }
    g256_store<FP8, HAS_EPI, 4, SWIGLU, WQ>(acc, C, M, N, ldc, m0, n0, wr, wcol, r, q, bias, residual, xscale, wscale);
}

// ---- 256 x 128 tile in the same schedule, for projections whose 256-wide grid would leave CUs idle (N = 4096 at 2048 tokens).
// Wave tile 128 x 32: a k-tile is TWO phases (A rows 0-63, A rows 64-127, each against the tile's 32 columns), and three DMA
// groups: B (128 weight rows), A0 (rows 0-63 of both 128-row halves), A1 (rows 64-127), 16 KiB each, in a ring of nine 16-KiB
// slots (group g -> slot g % 9; three k-tiles = six phases per loop body keep every LDS address static).
//   phase 2u:     reads A0(u) = group 3u+1;              issues A0(u+2) = group 3u+7;            waits until group 3u+3 has landed
//   phase 2u + 1: reads A1(u), B(u+1) = groups 3u+2, 3u+3; issues A1(u+2), B(u+3) = groups 3u+8, 3u+9; waits until group 3u+4 has landed
// Slot reuse: group g + 9 is issued exactly two phases after the last read of group g (A0(u-1): read 2u-2, refilled 2u;
// A1(u-1): 2u-1 -> 2u+1; B(u): 2u-1 -> 2u+1); 4-5 groups (64-80 KiB) in flight.
__device__ __forceinline__ void g8_wait_instrs(int n) {   // n = DMA wave instructions that may stay in flight (even, 0..10)
    if (n >= 10) { asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); return; }
    if (n >= 8) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); return; }
    if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// SWIGLU: W = fused gate_up [2I, K]; the tile's 128 weight rows are 64 gate rows n0 .. and the 64 up rows I + n0 ..; wave wc
// multiplies gate columns n0 + 16 wc .. + 16 and the matching up columns, C is [M, I] (N = 2I); col0 = first column of the launch.
// Used for the columns that a whole number of 256-CU rounds of the 128-column SwiGLU tiles leaves over.
// WQ = 8 (int8 weights, as in gemm8p_kernel): the B group is one wave instruction per wave (128 rows x 64 B in the first half of its
// slot); behind phase 2u the groups 3u+4 .. 3u+7 = A0 A1 B A0 stay in flight (7 instructions), behind phase 2u + 1 the groups
// 3u+5 .. 3u+9 = A1 B A0 A1 B (8).
template <bool FP8, bool HAS_EPI, bool SWIGLU = false, int WQ = 0, bool ROPE = false>
__global__ __launch_bounds__(512) void gemm8p_n128_kernel(const void *__restrict__ Xv, const void *__restrict__ Wv, half_t *C, int M, int N,
                                                          int K, const half_t *__restrict__ bias, const half_t *residual, int tiles_n,
                                                          const float *__restrict__ xscale, const float *__restrict__ wscale,
                                                          int ldc_arg = 0, int group_m = 0, int col0 = 0, const QkvRopeArgs *__restrict__ rap = nullptr, int layer = 0) {
    static_assert(!ROPE || (!HAS_EPI && !SWIGLU), "ROPE form: the plain projection");
    static_assert(!SWIGLU || !HAS_EPI, "SwiGLU form: no bias / residual");
    static_assert(WQ == 0 || (WQ == 8 && !FP8), "WQ: 0 (operands as they are) or 8 (int8 weights under fp16 activations)");
    const int half_n = N >> 1;
    const size_t ldc = ldc_arg ? ldc_arg : N;
    constexpr int ES = FP8 ? 1 : 2, BK = 128 / ES;
    constexpr unsigned SLOT_BYTES = 128 * 128;
    const unsigned char *X = static_cast<const unsigned char *>(Xv), *W = static_cast<const unsigned char *>(Wv);
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];  // [9 slots][128 rows x 128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r = lane & 15, q = lane >> 4;
    int tile_m, tile_n;
    if (group_m > 0) {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int qd = nwg >> 3, rem = nwg & 7;
        const int L = xcd * qd + min(xcd, rem) + idx;
        const int tm = (M + 255) / 256, per_group = group_m * tiles_n;
        const int grp = L / per_group, in = L - grp * per_group;
        const int first_m = grp * group_m, gsz = min(tm - first_m, group_m);
        tile_m = first_m + in % gsz;
        tile_n = in / gsz;
    } else {
        tile_m = blockIdx.x / tiles_n;
        tile_n = blockIdx.x - tile_m * tiles_n;
    }
    const int m0 = tile_m * 256, n0 = SWIGLU ? col0 + tile_n * 64 : tile_n * 128;

    // DMA plan: kind 0 = B, 1 = A0, 2 = A1 (the group order inside a k-tile); this wave moves pieces pi = i*8 + wave (8 slot rows each)
    const size_t row_bytes = static_cast<size_t>(K) * ES;
    const size_t wrow_bytes = WQ ? static_cast<size_t>(K) : row_bytes;
    constexpr unsigned WKT_BYTES = WQ ? 64 : 128;   // bytes of a weight row per k-tile
    const unsigned char *xbase = X + static_cast<size_t>(m0) * row_bytes, *wbase = W + static_cast<size_t>(n0) * wrow_bytes;
    asm volatile("" : "+s"(xbase), "+s"(wbase));   // (scalar-resident DMA bases, as in gemm8p_kernel)
    typedef __attribute__((address_space(3))) void *lptr_t;
    const unsigned lds_addr = static_cast<unsigned>(reinterpret_cast<size_t>((lptr_t)lds));
    unsigned voff[3][2], ldst[2];
    unsigned ldst_b = 0;   // WQ: LDS address of this wave's one weight piece (16 rows x 64 B)
    if constexpr (WQ != 0) {
        const int row = wave * 16 + (lane >> 2), chunk = static_cast<int>(g8_q8_slot(row, lane & 3));
        ldst_b = __builtin_amdgcn_readfirstlane(lds_addr + wave * 1024);
        long grow;
        if constexpr (SWIGLU) grow = static_cast<long>(row >> 6) * half_n + min(n0 + (row & 63), half_n - 1) - n0;
        else if constexpr (ROPE) grow = qkv_rope_col<2>(row);
        else grow = min(n0 + row, N - 1) - n0;
        voff[0][0] = static_cast<unsigned>(grow * static_cast<long>(wrow_bytes) + chunk * 16);
        voff[0][1] = 0;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pi = i * 8 + wave, row = pi * 8 + (lane >> 3), chunk = (lane & 7) ^ (row & 7);   // slot row of this lane
        ldst[i] = __builtin_amdgcn_readfirstlane(lds_addr + pi * 1024);
        if constexpr (WQ != 0) {
        } else if constexpr (SWIGLU)   // slot rows 0-63: gate rows n0 + row; 64-127: up rows I + n0 + row - 64 (clamped inside their half)
            voff[0][i] = static_cast<unsigned>((static_cast<long>(row >> 6) * half_n + min(n0 + (row & 63), half_n - 1) - n0) * static_cast<long>(row_bytes) + chunk * 16);
        else if constexpr (ROPE)
            voff[0][i] = static_cast<unsigned>(static_cast<long>(qkv_rope_col<2>(row)) * static_cast<long>(row_bytes) + chunk * 16);
        else
            voff[0][i] = static_cast<unsigned>(static_cast<long>(min(n0 + row, N - 1) - n0) * static_cast<long>(row_bytes) + chunk * 16);
#pragma unroll
        for (int a = 0; a < 2; ++a) {   // slot rows 0-63 -> A half 0, 64-127 -> A half 1; rows (row % 64) + 64 a of the half
            const int arow = (row >> 6) * 128 + (row & 63) + 64 * a;
            voff[1 + a][i] = static_cast<unsigned>(static_cast<long>(min(m0 + arow, M - 1) - m0) * static_cast<long>(row_bytes) + chunk * 16);
        }
    }
    auto issue = [&](auto kind_, int kt, unsigned slot) {
        constexpr int kind = decltype(kind_)::value;
        if constexpr (WQ != 0 && kind == 0) {
            g8_dma16(voff[0][0], wbase + static_cast<size_t>(kt) * WKT_BYTES, ldst_b + slot * SLOT_BYTES);
        } else {
            const unsigned char *base = (kind == 0 ? wbase : xbase) + static_cast<size_t>(kt) * 128;
#pragma unroll
            for (int i = 0; i < 2; ++i) g8_dma16(voff[kind][i], base, ldst[i] + slot * SLOT_BYTES);
        }
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    // WQ: DMA wave instructions of the groups (first, last] still in flight; group g is the weight piece (1) when g % 3 == 0
    auto q8_inflight = [](int first, int last) {
        int n = 0;
        for (int g = first + 1; g <= last; ++g) n += (g % 3 == 0) ? 1 : 2;
        return n;
    };

    floatx4 acc[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int wcol = SWIGLU ? wc * 16 : wc * 32;
    const unsigned sw0 = static_cast<unsigned>((q ^ (r & 7)) << 4), sw1 = static_cast<unsigned>(((4 + q) ^ (r & 7)) << 4);   // (as in gemm8p_kernel)
    // B fragment jj: plain -> slot row wc*32 + jj*16 + r; SwiGLU -> jj = 0 the gate row wc*16 + r, jj = 1 the up row 64 + wc*16 + r
    constexpr unsigned BROW = WQ ? 64 : 128;
    const unsigned a_lane = (wr * 64 + r) * 128, b_lane = ((SWIGLU ? wc * 16 : wc * 32) + r) * BROW;
    const unsigned q8o0 = g8_q8_piece(r, q, 0), q8o1 = g8_q8_piece(r, q, 1);
    constexpr unsigned B_STEP = SWIGLU ? 64 * BROW : 16 * BROW;
    G8Frag<FP8> fa[4], fb[3][2];   // A piece; B fragments of k-tiles u % 3 = 0, 1, 2
    auto read_a = [&](unsigned slot) {
        const unsigned char *p = lds + slot * SLOT_BYTES + a_lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) g8_read<FP8>(fa[i], p + i * 2048, sw0, sw1);
    };
    auto read_b = [&](unsigned slot, G8Frag<FP8> (&f)[2]) {
        const unsigned char *p = lds + slot * SLOT_BYTES + b_lane;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            if constexpr (WQ != 0) g8_read_q8(f[jj], p + jj * B_STEP, q8o0, q8o1);
            else g8_read<FP8>(f[jj], p + jj * B_STEP, sw0, sw1);
        }
    };
    auto half_tile = [&](auto ih_, G8Frag<FP8> (&f)[2]) {
        constexpr int ih = decltype(ih_)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) g8_pin(fa[i]);
        g8_pin(f[0]); g8_pin(f[1]);
#pragma unroll
        for (int k = 0; k < (FP8 ? 1 : 2); ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) g8_mma<FP8>(acc[ih * 4 + i][jj], f[jj], fa[i], k);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) g8_pin(acc[ih * 4 + i][jj]);
    };

    const int KT = K / BK, G = 3 * KT;   // DMA groups: group g = k-tile g / 3, kind g % 3, slot g % 9
    // prologue: groups 0..6
    issue(K0{}, 0, 0); issue(K1{}, 0, 1); issue(K2{}, 0, 2);
    if (KT > 1) { issue(K0{}, 1, 3); issue(K1{}, 1, 4); issue(K2{}, 1, 5); }
    if (KT > 2) issue(K0{}, 2, 6);
    if constexpr (WQ != 0) g8_wait_vm(q8_inflight(1, min(7, G) - 1));
    else g8_wait_instrs(2 * (min(7, G) - 2));   // groups 0 and 1 landed
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_b(0, fb[0]);
    if (wr == 1) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // phase S (0..5) of the three-k-tile loop body: k-tile u with u % 3 = S / 2; STEADY as in gemm8p_kernel (at least four k-tiles
    // left: every piece exists, 4 / 5 groups stay in flight)
    auto phase = [&](auto S_, int u, auto steady_) {
        constexpr int S = decltype(S_)::value, s = S & 1;
        constexpr unsigned c = S >> 1, c1 = (c + 1) % 3, c2 = (c + 2) % 3;
        constexpr bool STEADY = decltype(steady_)::value;
        if constexpr (s == 0) {
            read_a(3 * c + 1);
            if (STEADY || u + 2 < KT) issue(K1{}, u + 2, 3 * c2 + 1);
            if constexpr (WQ != 0) {
                if constexpr (STEADY) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                else g8_wait_vm(q8_inflight(3 * u + 3, min(3 * u + 7, G - 1)));
            } else if constexpr (STEADY) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else g8_wait_instrs(2 * min(4, G - 3 * u - 4));
        } else {
            read_a(3 * c + 2);
            if (STEADY || u + 1 < KT) read_b(3 * c1, fb[c1]);
            if (STEADY || u + 2 < KT) issue(K2{}, u + 2, 3 * c2 + 2);
            if (STEADY || u + 3 < KT) issue(K0{}, u + 3, 3 * c);
            if constexpr (WQ != 0) {
                if constexpr (STEADY) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else g8_wait_vm(q8_inflight(3 * u + 4, min(3 * u + 9, G - 1)));
            } else if constexpr (STEADY) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else g8_wait_instrs(2 * min(5, G - 3 * u - 5));
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (s == 0) half_tile(K0{}, fb[c]);
        else half_tile(K1{}, fb[c]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto body = [&](int u, auto steady_) {
        phase(std::integral_constant<int, 0>{}, u, steady_); phase(std::integral_constant<int, 1>{}, u, steady_);
        phase(std::integral_constant<int, 2>{}, u + 1, steady_); phase(std::integral_constant<int, 3>{}, u + 1, steady_);
        phase(std::integral_constant<int, 4>{}, u + 2, steady_); phase(std::integral_constant<int, 5>{}, u + 2, steady_);
    };
    int u = 0;
    for (; u + 5 < KT; u += 3) body(u, std::true_type{});    // KT - (u + 2) >= 4
    for (; u + 2 < KT; u += 3) body(u, std::false_type{});
    if (u < KT) { phase(std::integral_constant<int, 0>{}, u, std::false_type{}); phase(std::integral_constant<int, 1>{}, u, std::false_type{}); }
    if (u + 1 < KT) { phase(std::integral_constant<int, 2>{}, u + 1, std::false_type{}); phase(std::integral_constant<int, 3>{}, u + 1, std::false_type{}); }
    if (wr == 0) __builtin_amdgcn_s_barrier();

    if constexpr (ROPE) {
        g256_store_qkv_rope<FP8, 2, WQ>(acc, C, M, ldc, m0, col0 + n0, wr, wcol, r, q, xscale, wscale, bias, rap, layer);
        return;
    }
    g256_store<FP8, HAS_EPI, 2, SWIGLU, WQ>(acc, C, M, N, ldc, m0, n0, wr, wcol, r, q, bias, residual, xscale, wscale);
}

}  // namespace llmie

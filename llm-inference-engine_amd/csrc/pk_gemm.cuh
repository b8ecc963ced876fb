// Batch-decode projections on PRE-PACKED weights (4 < M <= 32 token rows): y = x . W^T for launchLinearGemm
// (src/kernels/linear.cu:10-87) at decode batch sizes, HBM-bound (bytes = N*K*w per launch).
//
// Why a packed copy: an MFMA A operand wants 16 DIFFERENT weight rows across the lanes of one load, a DRAM stream
// wants the lanes of one load on ONE contiguous KiB.  With row-major weights those two exclude each other (the round-1
// split-K kernel bought the contiguity back with a per-wave LDS transposer and still streamed at 2-3.6 TB/s).  The
// MI355X has 288 GB of HBM, so the engine keeps a second, tile-packed image of every matrix it streams at these batch
// sizes: tile = 16 weight rows, block = KB consecutive k of those 16 rows = exactly 1 KiB laid out so that lane l of a
// wave reads its 16 bytes at offset 16*l and owns row (l & 15), k-group (l >> 4) -- the MFMA fragment itself.  A tile is
// nblk contiguous KiB: every wave instruction is one contiguous KiB and consecutive instructions continue the stream.
//
//   block image [q = 0..3][r = 0..15][16 bytes]  (lane l = 16 q + r), per format:
//     fp16        KB = 32 : bytes = W[r][32 j + 8 q + (0..7)]                                  1 MFMA step / block
//     int8 / e4m3 KB = 64 : bytes 8 s + e = W[r][64 j + 32 s + 8 q + e], s = 0,1              2 steps / block
//     int4        KB = 128: word s = W[r][128 j + 32 s + 8 q + (0..7)], s = 0..3, nibbles placed so that the
//                           0x6400-magic unpack yields natural k order                        4 steps / block
//
// Kernel (the M = 1 K-split GEMV's structure, with MFMA in place of the dot products): ONE 512-thread workgroup per CU,
// persistent over 16-row tiles; its 8 waves split K, so the activation slice a wave multiplies is fixed for the whole
// launch and lives in REGISTERS as ready-made B fragments (loaded once, the fused RMSNorm applied in place) -- no
// activation staging, no barrier in front of the stream.  Per tile a wave loads XBLK contiguous KiB (ring of 16 loads in
// flight per lane, refilled as each register is consumed), de-quantises in registers, issues 2 * MT MFMAs per block,
// and the 8 partial tiles meet in LDS once per tile (double-buffered slot, one barrier per tile, like the GEMV).  The
// epilogue (row scale, bias, residual, SwiGLU over a gate tile and its up tile, or fp32 split-K slabs) runs in registers
// of the first MT waves: no fp32 slab round trip except where K itself is split over workgroups (down projection).
#pragma once
#include "gemm_kernels.cuh"

#include <type_traits>
#include <utility>

namespace llmie {

enum : int { PK_F16 = 16, PK_I8 = 8, PK_I4 = 4, PK_FP8 = 108 };
enum : int { PK_EPI_PLAIN = 0, PK_EPI_SWIGLU = 1, PK_EPI_SLAB = 2 };
enum : int { PK_X32_X = 1, PK_X32_Y = 2, PK_X32_RES = 4 };   // which operands of a call are in the x32 activation layout
constexpr int PK_NORM_MAX_K = 8192;   // fused-norm prologue: gamma / pre_bias staged in LDS by 2 chunks per thread

template <int WF> struct PkFmt;
template <> struct PkFmt<PK_F16> {
    static constexpr int KB = 32, SPB = 1, XBLK = 16;
};
template <> struct PkFmt<PK_I8> {
    static constexpr int KB = 64, SPB = 2, XBLK = 8;
};
template <> struct PkFmt<PK_FP8> {
    static constexpr int KB = 64, SPB = 2, XBLK = 8;
};
template <> struct PkFmt<PK_I4> {
    static constexpr int KB = 128, SPB = 4, XBLK = 4;
};

struct PkArgs {
    const half_t *x;             // [M, K] fp16
    const unsigned char *Wp;     // packed image: [tiles][nblk][1 KiB]
    int M, K, N;                 // N = logical output features (PLAIN / SLAB) or 2 * inter (SWIGLU)
    int units;                   // work units along N: tiles (PLAIN / SLAB) or gate/up tile pairs (SWIGLU)
    int nblk;                    // K / KB
    int bps;                     // blocks per K slice (gridDim.y slices)
    half_t *y;                   // PLAIN: [M, N]; SWIGLU: [M, N/2]
    float *slab;                 // SLAB: [gridDim.y][M][N] fp32 partial sums (unscaled)
    const void *scale;           // int8: fp16 [N] per row; fp8: fp32 [N]; int4: packed group scales [tiles][nblk][16] fp16
    const half_t *residual;      // [M, N] or null (PLAIN; may alias y)
    int x_x32, y_x32, res_x32;   // x / y / residual are in the fragment-ordered "x32" activation layout (below) instead of row-major
    const half_t *gamma;         // [K]: rmsnorm(x + pre_bias) * gamma fused in front (gridDim.y == 1 only), or null
    const half_t *pre_bias;      // [K] or null
    float eps;
};

// "x32" activation layout of a [<= 32 rows, K] fp16 matrix, K % 32 == 0: [K / 32][2 row tiles][64 lanes][8 halves], lane =
// 16 * ((k % 32) / 8) + (m % 16) -- each (32 k, 16 rows) piece is the 1 KiB MFMA B fragment of a 16x16x32 step, so a consumer
// wave loads its operand registers with contiguous KiB reads and no transposition (measured: fragment-shaped loads of a
// row-major matrix are TA-bound, row-contiguous loads + an LDS transposition cost ~1 us of a 32-row prologue).  The engine
// keeps its internal activations (residual stream, attention output, SwiGLU output) in this layout between the packed
// kernels; rows >= M of a tile are never read as data (the consumer zeroes them).  Always 32 rows of storage: K * 64 bytes.
__host__ __device__ inline size_t x32_offset(int m, int k) {
    return (static_cast<size_t>(k >> 5) * 2 + (m >> 4)) * 512 + ((k & 31) >> 3) * 128 + (m & 15) * 8 + (k & 7);
}

// A fragment of MFMA step s from one 16-byte packed chunk
template <int WF> __device__ __forceinline__ half8_t pk_afrag(const uint4_t &w, const int s) {
    if constexpr (WF == PK_F16) {
        return __builtin_bit_cast(half8_t, w);
    } else if constexpr (WF == PK_I8) {
        return dequant_i8x8(w[2 * s], w[2 * s + 1]);
    } else {
        // int4 word: nibble i at bits 4i.  (w & 0x000F000F)|0x6400.. = (1024 + n0, 1024 + n4); bits 4-7 land 16x higher and
        // come back with one packed fma (x/16 - 72 = n - 8); a shift by 8 exposes n2,n6 / n3,n7 to the same masks.
        // The packer stores k = 0..7 of the step as nibbles (n0,n4,n1,n5,n2,n6,n3,n7): the result is in natural k order.
        const half2_t off8 = {static_cast<half_t>(1032.f), static_cast<half_t>(1032.f)};
        const half2_t sixteenth = {static_cast<half_t>(0.0625f), static_cast<half_t>(0.0625f)};
        const half2_t off72 = {static_cast<half_t>(72.f), static_cast<half_t>(72.f)};
        const unsigned int w0 = w[s], w8 = w0 >> 8;
        const half2_t h0 = as_half2((w0 & 0x000F000Fu) | 0x64006400u) - off8;
        const half2_t h1 = as_half2((w0 & 0x00F000F0u) | 0x64006400u) * sixteenth - off72;
        const half2_t h2 = as_half2((w8 & 0x000F000Fu) | 0x64006400u) - off8;
        const half2_t h3 = as_half2((w8 & 0x00F000F0u) | 0x64006400u) * sixteenth - off72;
        return half8_t{h0[0], h0[1], h1[0], h1[1], h2[0], h2[1], h3[0], h3[1]};
    }
}

// ---- loads of the kernel ----
// (Every asm statement with an SGPR base opens with s_nop 4: hipcc pads no hazard inside an asm string, and under register
// pressure it restores a spilled SGPR with v_readlane right in front of the statement -- "VALU writes SGPR -> VMEM reads it"
// needs 5 wait states (cdna_hip_programming.md 5.7 item 2).  Without the pad a load used a stale base: GPU memory fault.)
// Weights: LDS-DMA (global_load_lds_dwordx4, one KiB per wave instruction, no VGPR destination) into a per-wave ring of D
// one-KiB slots, consumed D blocks later with a hand-counted s_waitcnt vmcnt(D - 1) and one ds_read_b128.  Two earlier
// forms of this kernel kept the ring in registers: (1) with hipcc-visible loads the compiler's vmcnt bookkeeping merges
// every path through the loop (refill / no-refill tails, the epilogue branch) to the most conservative count and drained
// the whole ring at the top of every tile; (2) with the loads hidden in asm the counts were exact, but a 64-register ring
// beside the 128 registers of B fragments left hipcc no slack at 32 rows: spills of registers whose asm load was still in
// flight, and loop-carried moves of them, store / copy garbage (NaN at M = 32).  A ring in LDS has neither problem, costs
// one ds_read per KiB, and its depth is set by LDS capacity (96 KiB in flight per CU), not by registers.
// Activations / gamma / epilogue operands: asm loads in the PROLOGUE only (issued before the first DMA, one exact wait), so
// the loop contains no VGPR-returning global load at all; the only VMEM operations hipcc sees there are the epilogue
// stores, which nobody waits for (an op issued behind a DMA makes a counted wait retire one more DMA than needed, never
// one less).
__device__ __forceinline__ void pk_gload16_s(uint4_t &dst, const unsigned voff, const void *sbase) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
// one KiB per wave: 16 bytes per lane from sbase + voff (sbase wave-uniform: scalar pointer arithmetic only, no VALU) -> LDS
// lds_dst + 16 lane (M0 = wave-uniform LDS byte address, written in the same statement that uses it: hipcc does not preserve M0
// for asm and keeps nothing of its own there in this kernel -- no DMA builtin, no indirect register indexing).  In asm, not the
// builtin: hipcc orders every LDS write of its own behind a pending builtin DMA with s_waitcnt vmcnt(0) (measured here: each
// table store of the prologue drained the ring fill).
__device__ __forceinline__ void pk_dma16_nt(const unsigned voff, const void *sbase, const unsigned lds_dst) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// 256 bytes per wave (4 per lane): the int4 group-scale records of one tile slice
__device__ __forceinline__ void pk_dma4(const unsigned voff, const void *sbase, const unsigned lds_dst) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void pk_gload16(uint4_t &dst, const void *p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void pk_gload8(uint2 &dst, const void *p) {
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N> __device__ __forceinline__ void pk_vmwait() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// the register was written by an asm load that the wait above has retired: nothing that reads it may move above this point
__device__ __forceinline__ void pk_landed(uint4_t &r) { asm volatile("" : "+v"(r)::"memory"); }
__device__ __forceinline__ void pk_landed(uint2 &r) { asm volatile("" : "+v"(r)::"memory"); }

// compile-time loop: f(std::integral_constant<int, 0>{}), ... (register arrays indexed by the loop variable stay registers)
template <typename Fn, int... Is> __device__ __forceinline__ void pk_static_for_impl(Fn &&f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename Fn> __device__ __forceinline__ void pk_static_for(Fn &&f) {
    pk_static_for_impl(static_cast<Fn &&>(f), std::make_integer_sequence<int, N>{});
}

// Workgroup barrier that leaves the LDS-DMA ring in flight: __syncthreads() would make hipcc drain vmcnt to 0 while a
// global_load_lds is pending (cdna_hip_programming.md 5, "Pipelining across barriers"); LDS traffic of this wave is complete
// (lgkmcnt 0) before it arrives, the asm memory clobbers keep the compiler from moving LDS accesses across.
__device__ __forceinline__ void pk_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

#ifdef PK_STAMPS   // diagnostic build only (tools/pkstamps.py): where a workgroup's time goes; never defined in the product build
__device__ unsigned long long pk_stamp_buf[256 * 8 * 16];
#define PK_STAMP(i) do { if (lane == 0) pk_stamp_buf[(blockIdx.x * 8 + wave) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PK_STAMP(i) do { } while (0)
#endif

constexpr int PK_MAX_UNITS = 16;   // work units per workgroup whose epilogue operands are pre-staged in LDS
__host__ __device__ constexpr int pk_ring_depth(int) { return 12; }
// LDS carve (bytes), shared by the kernel and the host launcher
struct PkLds {
    int red, ring, etab, rtab, total;
};
__host__ __device__ constexpr PkLds pk_lds(int mt, int epi, int norm_halves /* staged gamma (+ pre_bias) halves: K or 2 K, 0 = no norm */, bool resid) {
    const int tpi = epi == PK_EPI_SWIGLU ? 2 : 1;
    const int npar = tpi == 1 ? 2 : 1;
    int red = npar * 8 * tpi * mt * 1024;                  // [parities][8 waves][tpi * mt tiles][64 lanes] floatx4
    if (red < 8 * 4096) red = 8 * 4096;                    // prologue only, aliases `red`: one 4 KiB transposition patch per wave (row-major x)
    const int ring = 8 * pk_ring_depth(epi) * 1024;
    const int etab = epi == PK_EPI_SLAB ? 0 : PK_MAX_UNITS * tpi * 16 * 4;   // row scales (fp32) of every unit of the workgroup
    const int rtab = resid ? PK_MAX_UNITS * mt * 16 * 16 * 2 : 0;            // residual pieces [unit][mt * 16 rows][16] fp16
    const int stat = 2 * 8 * mt * 16 * 4 + 2 * norm_halves;   // norm / amax statistics, gamma (+ pre_bias) staging
    return PkLds{red, ring, etab, rtab, red + ring + etab + rtab + stat};
}

// A value every lane holds identically, moved into scalar registers (the chain kernel reads its phase descriptors through a
// run-time index: hipcc then keeps them in VGPRs, and the "s" operands of the asm loads need SGPRs)
__device__ __forceinline__ int pk_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T> __device__ __forceinline__ T *pk_uni(T *p) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(u)));
    const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(u >> 32)));
    return reinterpret_cast<T *>((static_cast<unsigned long long>(hi) << 32) | lo);
}
__device__ __forceinline__ PkArgs pk_uni(const PkArgs &a) {
    PkArgs u;
    u.x = pk_uni(a.x); u.Wp = pk_uni(a.Wp);
    u.M = pk_uni(a.M); u.K = pk_uni(a.K); u.N = pk_uni(a.N); u.units = pk_uni(a.units); u.nblk = pk_uni(a.nblk); u.bps = pk_uni(a.bps);
    u.y = pk_uni(a.y); u.slab = pk_uni(a.slab); u.scale = pk_uni(a.scale); u.residual = pk_uni(a.residual);
    u.x_x32 = pk_uni(a.x_x32); u.y_x32 = pk_uni(a.y_x32); u.res_x32 = pk_uni(a.res_x32);
    u.gamma = pk_uni(a.gamma); u.pre_bias = pk_uni(a.pre_bias);
    u.eps = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.eps)));
    return u;
}

// This wave's weight stream of one projection: the running DMA source pointer over (unit, tile, block) and the ring fill.  A
// struct of its own because the persistent chain kernel (pk_chain_kernel below) starts the NEXT projection's stream -- its first
// D ring elements -- before the grid barrier that ends the current one: the addresses of a weight stream depend on nothing a
// previous phase computes.
template <int WF> struct PkStream {
    static constexpr bool I4 = WF == PK_I4;
    const unsigned char *src, *ssrc;
    long step_in, step_unit, sstep_in, sstep_unit;
    int ri, ru, ept, T, tpi;
    // bx / by / gx: the workgroup's coordinates in the launch plan of this projection (gx workgroups along N, slice by of K)
    __device__ __forceinline__ void init(const PkArgs &a, const int tpi_, const int bx, const int by, const int gx, const int wave) {
        constexpr int NW = 8;
        tpi = tpi_;
        const int nb0 = by * a.bps, nb1 = min(a.nblk, nb0 + a.bps), nbs = nb1 - nb0;
        const int per = nbs / NW, rem = nbs - per * NW;
        const int cnt = per + (wave < rem ? 1 : 0);
        const int blk0 = nb0 + wave * per + min(wave, rem);
        const int iters = bx < a.units ? (a.units - bx + gx - 1) / gx : 0;
        ept = cnt > 0 ? cnt + (I4 ? 1 : 0) : 0;
        T = iters * tpi * ept;
        const size_t tile_bytes = static_cast<size_t>(a.nblk) * 1024;
        step_in = static_cast<long>(tile_bytes) - static_cast<long>(cnt) * 1024;
        step_unit = static_cast<long>(static_cast<size_t>(gx) * tpi - (tpi - 1)) * static_cast<long>(tile_bytes) - static_cast<long>(cnt) * 1024;
        src = a.Wp + (static_cast<size_t>(bx) * tpi * a.nblk + blk0) * 1024;
        ssrc = I4 ? reinterpret_cast<const unsigned char *>(a.scale) + (static_cast<size_t>(bx) * tpi * a.nblk + blk0) * 32 : nullptr;
        sstep_in = static_cast<long>(a.nblk) * 32;
        sstep_unit = static_cast<long>(static_cast<size_t>(gx) * tpi - (tpi - 1)) * static_cast<long>(a.nblk) * 32;
        ri = 0;
        ru = 0;
    }
    // the next stream element -> ring slot `slot_bytes` (ISSUE = false: only step the pointers, for elements a previous phase fetched)
    template <bool ISSUE = true> __device__ __forceinline__ void next(const unsigned ring_lds, const unsigned slot_bytes, const int lane) {
        if (I4 && ru == 0) {
            if constexpr (ISSUE) pk_dma4(lane * 4, pk_uni(ssrc), pk_uni(static_cast<int>(ring_lds + slot_bytes)));
        } else {
            if constexpr (ISSUE) pk_dma16_nt(lane * 16, pk_uni(src), pk_uni(static_cast<int>(ring_lds + slot_bytes)));
            src += 1024;
        }
        if (++ru == ept) {
            ru = 0;
            ++ri;
            src += (tpi == 2 && (ri & 1)) ? step_in : step_unit;
            if constexpr (I4) ssrc += (tpi == 2 && (ri & 1)) ? sstep_in : sstep_unit;
        }
    }
};

// One projection on the workgroup (bx, by) of a plan with gx workgroups along N.  `prefetched`: the first D elements of this
// wave's stream are already in (or on their way into) its ring -- issued by the previous phase of a chain launch.
// XM (activation layout, a compile-time fact of the launch): 0 = row-major x; 1 = x32 image
// `next_fill(ring_lds, wave, lane)`: called once per wave when its own stream has left the ring (the chain kernel issues the next
// projection's first D elements there; the single-projection kernel passes a no-op)
template <int MT, int WF, int EPI, int XM, typename NextFill>
__device__ __forceinline__ void pk_phase(const PkArgs &a, const int bx, const int by, const int gx, unsigned char *pk_smem, const bool prefetched,
                                         NextFill &&next_fill) {
    constexpr bool XL = XM != 0;
    using F = PkFmt<WF>;
    constexpr int NW = 8, KB = F::KB, SPB = F::SPB, XBLK = F::XBLK;
    constexpr bool FP8 = WF == PK_FP8;
    constexpr bool SCALED = WF == PK_I8 || FP8;          // per-row weight scales applied in the epilogue
    constexpr int TPI = EPI == PK_EPI_SWIGLU ? 2 : 1;    // tiles per work unit
    constexpr int NPAR = TPI == 1 ? 2 : 1;               // reduction-slot parities (SwiGLU: one slot set + a second barrier per unit)
    constexpr int D = pk_ring_depth(EPI);
    constexpr int XPB = SPB * MT;                        // activation fragments (= loads) per block
    typedef __attribute__((address_space(3))) void *lptr_t;
    const bool has_res = EPI == PK_EPI_PLAIN && a.residual != nullptr;
    const PkLds lay = pk_lds(MT, EPI, a.gamma ? (a.pre_bias ? 2 * a.K : a.K) : 0, has_res);
    floatx4 *red = reinterpret_cast<floatx4 *>(pk_smem);                                  // [NPAR][NW][TPI * MT][64]
    float *etab = reinterpret_cast<float *>(pk_smem + lay.red + lay.ring);               // [units][TPI][16] row scales
    half_t *rtab = reinterpret_cast<half_t *>(pk_smem + lay.red + lay.ring + lay.etab);  // [units][MT * 16][16] residual pieces
    float *stat = reinterpret_cast<float *>(pk_smem + lay.red + lay.ring + lay.etab + lay.rtab);   // [2][NW][MT][16] sum of squares | amax
    half_t *gam = reinterpret_cast<half_t *>(stat + 2 * NW * MT * 16);                   // gamma [K], pre_bias [K] (fused norm only)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: K range, block addresses stay scalar
    const int r = lane & 15, q = lane >> 4;
    // ---- this wave's K range: blocks [blk0, blk0 + cnt) of slice blockIdx.y ----
    const int nb0 = by * a.bps, nb1 = min(a.nblk, nb0 + a.bps), nbs = nb1 - nb0;
    const int per = nbs / NW, rem = nbs - per * NW;
    const int cnt = per + (wave < rem ? 1 : 0);                  // 0 <= cnt <= XBLK
    const int blk0 = nb0 + wave * per + min(wave, rem);
    auto blkc = [&](int u) { return min(blk0 + u, nb1 - 1); };   // in-bounds address for a slot the wave does not own (zero x)

    // ---- work units of this workgroup: blockIdx.x, + gridDim.x, ... ; its tile stream has L = iters * TPI elements ----
    unsigned char *ring_all = pk_smem + lay.red;
    unsigned char *ring = ring_all + wave * (pk_ring_depth(EPI) * 1024);
    const unsigned ring_lds = static_cast<unsigned>(reinterpret_cast<size_t>((lptr_t)ring));   // LDS byte address, wave-uniform
    // the next projection's ring fill (chain launch): issued when this wave's own stream has left the ring
    auto prefetch_next = [&]() { next_fill(ring_lds, wave, lane); };
    if (bx >= a.units) {   // no work unit of this projection for this workgroup (it still takes part in what follows the phase)
        prefetch_next();
        return;
    }
    const int iters = (a.units - bx + gx - 1) / gx;
    const int L = iters * TPI;
    auto unit_at = [&](int it) { return it * gx + bx; };
    // this wave's block stream: element k = (tile i = k / ept, block u = k % ept), T = L * ept elements.
    // int4 (group-128 scales, KB = 128 = one group per block): every tile's blocks are preceded by ONE more stream element, the
    // [blocks][16 rows] fp16 scale records of the wave's slice (<= 4 x 32 bytes, fetched as a 256-byte DMA into a ring slot of
    // its own) -- a ring element like any other, so every counted wait of the loop stays what it is.
    constexpr bool I4 = WF == PK_I4;
    constexpr int SE = I4 ? 1 : 0;
    const int ept = cnt > 0 ? cnt + SE : 0;
    const int T = L * ept;
    const unsigned woff = lane * 16;
    // DMA source of the NEXT stream element to fetch: a running scalar pointer (no multiplications in the loop).  Inside a
    // tile the wave's blocks are contiguous KiB; from its last block the pointer steps to the wave's first block of the next
    // tile of the unit, or of the workgroup's next unit.  (int4: the scale image [tiles][nblk][16] fp16 runs beside the weight
    // image, a.scale, 32 bytes per block.)
    PkStream<WF> strm;
    strm.init(a, TPI, bx, by, gx, wave);
    auto dma_next = [&](const unsigned slot_bytes) { strm.next(ring_lds, slot_bytes, lane); };

    PK_STAMP(0);
    // ================= issue phase: every asm load of the kernel except the ring refills, oldest first =================
    // Loads RETURN in issue order (one in-order path per CU; vmcnt counts them that way), and every 1 KiB wave-load takes the
    // CU's address path 16 cycles (64 B/clk): what a launch pays before its first MFMA is the bytes it requests up front --
    // activation slice 256 KiB per CU at 32 rows, ring fill 96 KiB, small operands -- plus one cold HBM latency.  Order:
    // (1) the small operands (needed first; only the ones this launch has), (2) the ring fill, so the HBM stream starts at
    // once, (3) the activation slice (L2 / MALL), which then arrives right behind the ring.  [Measured alternatives, batch 32,
    // in the decode step: activations in front of the ring fill: equal or 0.5-1.5 us/launch slower (HBM idles until they are
    // through); the first tiles streamed block-major so that the activations spread over 2-4 tiles of HBM time: equal (what
    // the spread gains, the back-to-back reductions of those tiles at the end of the pass cost).]
    // (1) small operands: gamma / pre_bias chunks for the LDS staging, and the epilogue operands of every unit of this
    //     workgroup (row scales, residual pieces) -> LDS tables, so that the loop contains no VGPR-returning global load at all.
    //     The wait for them counts the YOUNGER loads only, so their own number may differ from launch to launch.
    constexpr int GCH = PK_NORM_MAX_K / 4096;   // gamma / pre_bias: 16-byte chunks per thread
    constexpr bool HAS_ST = SCALED && EPI != PK_EPI_SLAB, HAS_RT = EPI == PK_EPI_PLAIN;
    uint4_t graw[GCH], praw[GCH];
    const int kchunks = a.K >> 3;
    if (a.gamma) {   // kernel-uniform
#pragma unroll
        for (int c = 0; c < GCH; ++c) {
            if (c * 512 < kchunks) {   // K <= 4096: one chunk per thread
                const unsigned goff = static_cast<unsigned>(min(c * 512 + tid, kchunks - 1)) * 16;
                pk_gload16_s(graw[c], goff, a.gamma);
                if (a.pre_bias) pk_gload16_s(praw[c], goff, a.pre_bias);
            }
        }
    }
    const int n_sc = HAS_ST ? iters * TPI * 4 : 0;                 // items of 4 row scales
    const int n_rs = has_res ? iters * MT * 16 * 4 : 0;            // items of 4 residual halves
    uint2 escale{0u, 0u};
    uint4_t escale4{0u, 0u, 0u, 0u};
    if (HAS_ST && tid < n_sc) {
        const int e = tid;
        const int it = e / (TPI * 4), j = (e / 4) % TPI, c = e & 3;
        int n = 16 * unit_at(it) + 4 * c;                          // PLAIN: feature; SWIGLU: inter index (+ inter for the up tile)
        if constexpr (EPI == PK_EPI_SWIGLU) n = min(n, (a.N >> 1) - 4) + j * (a.N >> 1);
        else n = min(n, a.N - 4);
        if constexpr (FP8) pk_gload16(escale4, reinterpret_cast<const float *>(a.scale) + n);
        else pk_gload8(escale, reinterpret_cast<const half_t *>(a.scale) + n);
    }
    uint2 eres2[HAS_RT ? 4 : 1];   // PK_MAX_UNITS * MT * 64 <= 2048 items = 4 rounds of the 512 threads
    if constexpr (HAS_RT) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int e = rr * 512 + tid;
            if (e < n_rs) {
                const int it = e / (MT * 64), m = min((e / 4) % (MT * 16), a.M - 1), c = e & 3;
                const int n = min(16 * unit_at(it) + 4 * c, a.N - 4);
                pk_gload8(eres2[rr], a.residual + (a.res_x32 ? x32_offset(m, n) : static_cast<size_t>(m) * a.N + n));
            }
        }
    }
    PK_STAMP(8);
    // (2) the ring fill: the first D blocks of the wave's weight stream
    if (prefetched) {
        for (int k = 0; k < D && k < T; ++k) strm.template next<false>(ring_lds, k * 1024, lane);
    } else {
        for (int k = 0; k < D && k < T; ++k) dma_next(k * 1024);
    }
    // (3) the activation slice of this wave = rows [0, 16 MT) x its XBLK blocks, XBLK * XPB loads in block order.
    // B fragments: lane (c = r, q) holds x[16 t + c][k .. k + 8), k = blk * KB + 32 s + 8 q; one register array from the load to
    // the MFMA operand (zeroing, bias, gamma in place).
    //   x32 image: fragment (u, s, t) is one contiguous KiB.
    //   row-major: fragment-shaped loads (16 rows x 64 bytes per instruction) are TA-bound (~90 cycles each: ~10 us of prologue
    //   that way), so the slice comes in row-contiguous pieces -- one instruction = 4 rows x 256 bytes -- into the SAME registers
    //   and is turned into fragments 4 KiB at a time through a private LDS patch after the wait:
    //     piece (t, c, i): rows 16 t + 4 i + (lane >> 4), bytes [256 c + 16 (lane & 15), +16) of the slice row
    //     fragment (u, s, t): lane (r, q) <- row 16 t + r, bytes [XB u + 64 s + 16 q, +16)
    constexpr int XB = KB * 2;                 // activation bytes per block and row (fp16 x): 64 / 128 / 256
    constexpr int PAIRB = 256;                 // bytes of one row piece
    constexpr int NPAIR = XBLK * XB / PAIRB;   // row pieces per row (4 for every format)
    static_assert(NPAIR * 4 == XBLK * SPB, "register count of pieces == fragments");
    uint4_t xw[XBLK][SPB][MT];
    auto piece = [&](int t, int c, int i) -> uint4_t & {   // the register that receives piece (t, c, i): a bijection onto xw
        const int f = c * 4 + i;
        return xw[f / SPB][f % SPB][t];
    };
    auto xload_block = [&](auto u_tag) {
        constexpr int u = decltype(u_tag)::value;
#pragma unroll
        for (int s2 = 0; s2 < SPB; ++s2)
#pragma unroll
            for (int t = 0; t < MT; ++t)
                pk_gload16_s(xw[u][s2][t], woff, reinterpret_cast<const unsigned char *>(a.x) + (static_cast<size_t>(blkc(u)) * SPB + s2) * 2048 + t * 1024);
    };
    if constexpr (XL) {
        pk_static_for<XBLK>([&](auto u_tag) { xload_block(u_tag); });
    } else {
        const size_t slice_bytes = static_cast<size_t>(a.K) * 2;
        // first byte of the window inside a row, clamped so that every piece stays inside the row
        const size_t sb0 = min(static_cast<size_t>(blk0) * XB, slice_bytes - static_cast<size_t>(XBLK) * XB);
#pragma unroll
        for (int c = 0; c < NPAIR; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const unsigned voff = static_cast<unsigned>(min(16 * t + 4 * i + (lane >> 4), a.M - 1)) * static_cast<unsigned>(slice_bytes) + 16 * (lane & 15);
                    pk_gload16_s(piece(t, c, i), voff, reinterpret_cast<const unsigned char *>(a.x) + sb0 + c * PAIRB);
                }
    }
    PK_STAMP(1);
    // ================= first wait: the small operands (a wave with a short stream waits for its ring fill as well) ==========
    if (T >= D) pk_vmwait<XBLK * XPB + D>();
    else pk_vmwait<XBLK * XPB>();
    PK_STAMP(2);
    if constexpr (HAS_ST) {
        if (tid < n_sc) {
            if constexpr (FP8) {
                pk_landed(escale4);
                *reinterpret_cast<uint4_t *>(etab + tid * 4) = escale4;
            } else {
                pk_landed(escale);
                const half4_t h = __builtin_bit_cast(half4_t, escale);
                *reinterpret_cast<floatx4 *>(etab + tid * 4) = floatx4{to_f32(h[0]), to_f32(h[1]), to_f32(h[2]), to_f32(h[3])};
            }
        }
    }
    if constexpr (HAS_RT) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            pk_landed(eres2[rr]);
            const int e = rr * 512 + tid;
            if (e < n_rs) *reinterpret_cast<uint2 *>(rtab + static_cast<size_t>(e) * 4) = eres2[rr];
        }
    }
#pragma unroll
    for (int c = 0; c < GCH; ++c) {
        pk_landed(graw[c]);
        pk_landed(praw[c]);
        const int ch = c * 512 + tid;
        if (a.gamma && ch < kchunks) {
            *reinterpret_cast<uint4_t *>(gam + static_cast<size_t>(ch) * 8) = graw[c];
            if (a.pre_bias) *reinterpret_cast<uint4_t *>(gam + a.K + static_cast<size_t>(ch) * 8) = praw[c];
        }
    }
    pk_barrier();   // gamma / pre_bias staged, tables visible to the epilogue waves
    PK_STAMP(3);

    // Fused RMSNorm (rmsnorm.cu / add_residual_and_rmsnorm.cu semantics): h = (x + pre_bias) * gamma * rsqrt(mean((x + pre_bias)^2) + eps).
    // The register slice is multiplied by gamma only (packed fp16: 4 instructions per fragment); the per-token factor
    // rsqrt(...) is a ROW scale of the product and is applied to the fp32 sums in the epilogue -- per element that is one
    // fp16 rounding (of x * gamma) where the unfused sequence has one (of the normalised value), and a 32-row prologue drops
    // from ~1100 to ~400 VALU instructions.
    const bool need_mask = cnt < XBLK || a.M < 16 * MT;   // wave-uniform; the common full case skips the selects
    float ss[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) ss[t] = 0.f;
    auto xfrag = [&](const uint4_t &v) { return __builtin_bit_cast(half8_t, v); };
    // block u's fragments have landed: zero what the wave does not own / rows past M, pre_bias, sum of squares, gamma
    auto prep = [&](auto u_tag) {
        constexpr int u = decltype(u_tag)::value;
#pragma unroll
        for (int s2 = 0; s2 < SPB; ++s2)
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                pk_landed(xw[u][s2][t]);
                if (need_mask && !(u < cnt && 16 * t + r < a.M)) xw[u][s2][t] = uint4_t{0u, 0u, 0u, 0u};
            }
        if (a.gamma) {
#pragma unroll
            for (int s2 = 0; s2 < SPB; ++s2) {
                const size_t k = static_cast<size_t>(blkc(u)) * KB + s2 * 32 + 8 * q;
                if (a.pre_bias) {
                    const half8_t b = *reinterpret_cast<const half8_t *>(gam + a.K + k);
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        if (u < cnt && 16 * t + r < a.M) xw[u][s2][t] = __builtin_bit_cast(uint4_t, xfrag(xw[u][s2][t]) + b);   // fp16 sum, as the unfused kernel stores it
                }
                const half8_t g = *reinterpret_cast<const half8_t *>(gam + k);
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    ss[t] = dot8(xfrag(xw[u][s2][t]), xfrag(xw[u][s2][t]), ss[t]);
                    xw[u][s2][t] = __builtin_bit_cast(uint4_t, xfrag(xw[u][s2][t]) * g);
                }
            }
        }
    };
    float inv_rms[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) inv_rms[t] = 1.f;
    auto norm_finalize = [&]() {   // after every block has been prepared: the per-token factor, for the epilogue
        if (a.gamma) {
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                ss[t] = lane_xor_sum<16>(ss[t]);   // (row swaps: no LDS round trip; device_utils.cuh)
                ss[t] = lane_xor_sum<32>(ss[t]);
                if (q == 0) stat[(wave * MT + t) * 16 + r] = ss[t];
            }
            pk_barrier();
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) tot += stat[(w * MT + t) * 16 + r];
                inv_rms[t] = rsqrtf(tot / static_cast<float>(a.K) + a.eps);   // of token 16 t + r: this lane's B column AND its D column
            }
        }
    };

    float xscale[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) xscale[t] = 1.f;
    {
        pk_vmwait<0>();   // the activation slice (the ring fill is older: landed as well)
        if constexpr (!XL) {
            // pieces -> fragments through this wave's 4 KiB patch of the (still unused) reduction region: [16 rows][16 chunks] of
            // 16 bytes, chunk index XOR row (conflict-free 16-row column reads); same-wave LDS operations execute in order
            unsigned char *patch = pk_smem + wave * 4096;
            const int own0 = blk0 - static_cast<int>(min(static_cast<size_t>(blk0) * XB, static_cast<size_t>(a.K) * 2 - static_cast<size_t>(XBLK) * XB) / XB);
            // own0 = index, inside the fetched window, of the wave's first own block (0 unless the window was clamped at the row end)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int c = 0; c < NPAIR; ++c) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = 4 * i + (lane >> 4);
                        pk_landed(piece(t, c, i));
                        *reinterpret_cast<uint4_t *>(patch + row * 256 + (((lane & 15) ^ row) << 4)) = piece(t, c, i);
                    }
                    asm volatile("" ::: "memory");
                    // the 4 fragments this 256-byte column range holds, back into the same 4 registers: fragment f = 4 c + j is
                    // (window block f / SPB, step f % SPB); lane (r, q) takes chunk 4 j + q of row r
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        piece(t, c, j) = *reinterpret_cast<const uint4_t *>(patch + r * 256 + (((j * 4 + q) ^ r) << 4));
                    asm volatile("" ::: "memory");
                }
            // window block wb = own block u + own0: shift down (own0 > 0 only for the last wave(s) of a clamped window)
            if (own0 > 0) {
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int sft = 0; sft < XBLK; ++sft) {   // at most XBLK - 1 single-block shifts
                        if (sft < own0) {
#pragma unroll
                            for (int u = 0; u + 1 < XBLK; ++u)
#pragma unroll
                                for (int s2 = 0; s2 < SPB; ++s2) xw[u][s2][t] = xw[u + 1][s2][t];
                        }
                    }
            }
        }
        pk_static_for<XBLK>([&](auto u_tag) { prep(u_tag); });
    }
    if constexpr (!XL) pk_barrier();   // the patches alias the reduction slots of the first publish
    if constexpr (FP8) {
        // fp8: the activation rows are quantised per token to the e4m3 grid (scale amax / 448: quantize_rows_fp8's arithmetic;
        // the per-token norm factor cancels in value / amax, so the codes are those of the normalised row) and packed 8 bytes
        // per fragment; weights and activations go to v_mfma_f32_16x16x32_fp8_fp8 unconverted
        float *amx = stat + NW * MT * 16;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float amax = 0.f;
#pragma unroll
            for (int u = 0; u < XBLK; ++u)
#pragma unroll
                for (int s2 = 0; s2 < SPB; ++s2)
#pragma unroll
                    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(to_f32(xfrag(xw[u][s2][t])[e])));
            amax = lane_xor_max<16>(amax);
            amax = lane_xor_max<32>(amax);
            if (q == 0) amx[(wave * MT + t) * 16 + r] = amax;
        }
        pk_barrier();
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float amax = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) amax = fmaxf(amax, amx[(w * MT + t) * 16 + r]);
            const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
            xscale[t] = sc;   // of token 16 t + r: this lane's B column AND its D column
            const float rsc = 1.0f / sc;
#pragma unroll
            for (int u = 0; u < XBLK; ++u)
#pragma unroll
                for (int s2 = 0; s2 < SPB; ++s2) {
                    // the e4m3 fragment (8 bytes) replaces the fp16 one in the low half of its register quad
                    const half8_t v = xfrag(xw[u][s2][t]);
                    // (x * (1 / sc), not x / sc: 128 IEEE divisions per lane were 15 us of a 35 us launch at 32 rows; the product
                    // can differ from the quotient by one ulp before the e4m3 rounding, i.e. flip a code in a rare tie)
                    xw[u][s2][t][0] = pack4_e4m3(to_f32(v[0]) * rsc, to_f32(v[1]) * rsc, to_f32(v[2]) * rsc, to_f32(v[3]) * rsc);
                    xw[u][s2][t][1] = pack4_e4m3(to_f32(v[4]) * rsc, to_f32(v[5]) * rsc, to_f32(v[6]) * rsc, to_f32(v[7]) * rsc);
                }
        }
    }
    PK_STAMP(4);

    // ---- reduction + epilogue of a finished unit ----
    floatx4 acc[MT];
    auto publish = [&](const floatx4 (&av)[MT], const int j, const int parity) {
        floatx4 *slot = red + static_cast<size_t>(parity) * NW * TPI * MT * 64;
#pragma unroll
        for (int t = 0; t < MT; ++t) slot[(wave * TPI * MT + j * MT + t) * 64 + lane] = av[t];
    };
    auto finish = [&](const int it, const int parity) {
        const floatx4 *slot = red + static_cast<size_t>(parity) * NW * TPI * MT * 64;
        pk_barrier();  // every wave's partial tile is in the slot (two parities: the other one is free for the next unit)
        if (wave < MT) {
            const int unit = unit_at(it);
            const int t = wave, m = 16 * t + r;
            floatx4 v[TPI];
#pragma unroll
            for (int j = 0; j < TPI; ++j) {
                v[j] = slot[(j * MT + t) * 64 + lane];
#pragma unroll
                for (int w = 1; w < NW; ++w) v[j] += slot[(w * TPI * MT + j * MT + t) * 64 + lane];
            }
            const int n0 = 16 * unit + 4 * q;  // PLAIN / SLAB: output features n0..n0+3; SWIGLU: inter index
            float xs = t == 0 ? inv_rms[0] : inv_rms[MT - 1];               // MT <= 2: no runtime-indexed register array
            if constexpr (FP8) xs *= t == 0 ? xscale[0] : xscale[MT - 1];
            if constexpr (EPI == PK_EPI_SLAB) {
                // partial sums without the weight-row scale (applied by the reduce launch).  fp8: times THIS slice's activation scale
                // -- a K-split launch quantises every token's activations per slice (amax over the slice's k range), a finer grid
                // than the one-scale-per-token of the unsplit launch
                if constexpr (FP8) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[0][e] *= xs;
                }
                if (m < a.M && n0 < a.N)
                    *reinterpret_cast<floatx4 *>(a.slab + (static_cast<size_t>(by) * a.M + m) * a.N + n0) = v[0];
            } else if constexpr (EPI == PK_EPI_SWIGLU) {
                const int inter = a.N >> 1;
                floatx4 sg{1.f, 1.f, 1.f, 1.f}, su{1.f, 1.f, 1.f, 1.f};
                if constexpr (SCALED) {
                    sg = *reinterpret_cast<const floatx4 *>(etab + ((it * 2 + 0) * 4 + q) * 4);
                    su = *reinterpret_cast<const floatx4 *>(etab + ((it * 2 + 1) * 4 + q) * 4);
                }
                half4_t y4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gt = v[0][e] * sg[e] * xs, up = v[1][e] * su[e] * xs;
                    y4[e] = from_f32<half_t>((gt / (1.0f + expf(-gt))) * up);
                }
                if (m < a.M && n0 < inter)
                    *reinterpret_cast<half4_t *>(a.y + (a.y_x32 ? x32_offset(m, n0) : static_cast<size_t>(m) * inter + n0)) = y4;
            } else {
                floatx4 sc{1.f, 1.f, 1.f, 1.f};
                if constexpr (SCALED) sc = *reinterpret_cast<const floatx4 *>(etab + (it * 4 + q) * 4);
                half4_t r4{0, 0, 0, 0};
                if (a.residual) r4 = *reinterpret_cast<const half4_t *>(rtab + ((static_cast<size_t>(it) * MT * 16 + m) * 4 + q) * 4);
                half4_t y4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float f = v[0][e] * sc[e] * xs;
                    if (a.residual) f += to_f32(r4[e]);
                    y4[e] = from_f32<half_t>(f);
                }
                if (m < a.M && n0 < a.N)
                    *reinterpret_cast<half4_t *>(a.y + (a.y_x32 ? x32_offset(m, n0) : static_cast<size_t>(m) * a.N + n0)) = y4;
            }
        }
        if constexpr (NPAR == 1) pk_barrier();   // one slot set: the reducers are done with it before the next unit publishes
    };

    // ---- the block stream, software-pipelined by one block: while block k is multiplied the LDS read of block k + 1 is in
    //      flight and the DMA of block k + D has been issued into the slot block k has just left ----
    // The kernel is instruction-issue bound before it is HBM bound (PMC: the first form of this loop, with its per-block
    // index arithmetic and tail tests, spent ~120 instructions per KiB and kept the SIMDs 78 % busy at 3 TB/s), so a tile whose
    // blocks are all in the steady state (the common case) runs a branch-free body: no tail tests, no per-block ownership
    // test, two alternating block registers instead of a copy.
    unsigned sb = 0;         // ring slot (byte offset) of the stream element being multiplied
    int kk = 0;              // stream index of the first block of the current tile
    const unsigned lds_rd = static_cast<unsigned>(wave * (D * 1024) + lane * 16);
    auto lds_block = [&](const unsigned slot_bytes) { return *reinterpret_cast<const uint4_t *>(ring_all + lds_rd + slot_bytes); };
    auto next_slot = [&](const unsigned b) { return b + 1024 == D * 1024 ? 0u : b + 1024; };
    uint4_t wq[2];           // block kk in wq[0] at every tile start
    wq[0] = uint4_t{0u, 0u, 0u, 0u};
    wq[1] = wq[0];
    if (T > 0) wq[0] = lds_block(0);
    floatx4 scf[I4 ? XBLK : 1];   // int4: group scales of the current tile's blocks, rows 4 q + e of the tile
    // the scale element at ring slot `slot` (landed): records [block][16 rows] fp16, this lane takes rows 4 q .. 4 q + 3
    auto read_scales = [&](const unsigned slot_bytes) {
        if constexpr (I4) {
            half4_t raw[XBLK];
#pragma unroll
            for (int u = 0; u < XBLK; ++u)
                raw[u] = *reinterpret_cast<const half4_t *>(ring_all + wave * (D * 1024) + slot_bytes + u * 32 + q * 8);
#pragma unroll
            for (int u = 0; u < XBLK; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) scf[u][e] = to_f32(raw[u][e]);   // (the conversions make the reads complete before the refill)
        }
    };
    auto mma_into = [&](floatx4 (&acc)[MT], auto u_tag, const uint4_t &w, const half8_t (&af)[(FP8 || WF == PK_F16) ? 1 : SPB]) {
        constexpr int u = decltype(u_tag)::value;
        if constexpr (FP8) {
#pragma unroll
            for (int s = 0; s < SPB; ++s) {
                const long wa = static_cast<long>((static_cast<unsigned long>(w[2 * s + 1]) << 32) | w[2 * s]);
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const long xa = static_cast<long>((static_cast<unsigned long>(xw[u][s][t][1]) << 32) | xw[u][s][t][0]);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wa, xa, acc[t], 0, 0, 0);
                }
            }
        } else if constexpr (WF == PK_F16) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, w), xfrag(xw[u][0][t]), acc[t], 0, 0, 0);
        } else if constexpr (I4) {
            // one group per block: the block's 4 k-steps accumulate from zero and enter the running sums through the group scales
            // of the lane's 4 weight rows (D rows 4 q + e): 4 FMAs per row tile instead of 16 packed multiplies of the A fragments
            floatx4 tmp[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) tmp[t] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < SPB; ++s)
#pragma unroll
                for (int t = 0; t < MT; ++t) tmp[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[s], xfrag(xw[u][s][t]), tmp[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][e] = fmaf(tmp[t][e], scf[u][e], acc[t][e]);
        } else {
#pragma unroll
            for (int s = 0; s < SPB; ++s)
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[s], xfrag(xw[u][s][t]), acc[t], 0, 0, 0);
        }
    };
    auto mma_block = [&](auto u_tag, const uint4_t &w, const half8_t (&af)[(FP8 || WF == PK_F16) ? 1 : SPB]) { mma_into(acc, u_tag, w, af); };
    auto dequant = [&](const uint4_t &w, half8_t (&af)[(FP8 || WF == PK_F16) ? 1 : SPB]) {
        if constexpr (!FP8 && WF != PK_F16) {
#pragma unroll
            for (int s = 0; s < SPB; ++s) af[s] = pk_afrag<WF>(w, s);
        }
    };
    // One tile = ept stream elements: (int4: the scale element, then) the wave's blocks.  Step e of a tile works on the element in
    // wq[e & 1] and fetches its successor into the other register.
    auto tile_blocks = [&]() {
        constexpr int STEPS = XBLK + SE;
        if (cnt == XBLK && kk + STEPS + D <= T) {
            // steady tile: every element's refill (element + D) and successor (element + 1) exist and are D - 1 deep
            pk_static_for<STEPS>([&](auto e_tag) {
                constexpr int e = decltype(e_tag)::value;
                uint4_t &w = wq[e & 1];
                pk_landed(w);   // its LDS read (issued a step ago) has returned: the slot may be overwritten by the refill below
                // successor first: its LDS latency hides behind this block's de-quantisation and MFMAs (element + D is not
                // issued yet: D - 2 younger DMAs)
                const unsigned nsb = next_slot(sb);
                pk_vmwait<D - 2>();
                wq[(e + 1) & 1] = lds_block(nsb);
                if constexpr (I4 && e == 0) {
                    read_scales(sb);
                    dma_next(sb);
                    sb = nsb;
                } else {
                    half8_t af[(FP8 || WF == PK_F16) ? 1 : SPB];
                    dequant(w, af);
                    dma_next(sb);
                    sb = nsb;
                    mma_block(std::integral_constant<int, (e - SE < 0 ? 0 : e - SE)>{}, w, af);
                }
            });
            if constexpr (STEPS & 1) wq[0] = wq[1];   // the next tile's first element is expected in wq[0]
            kk += STEPS;
        } else if (kk + ept + D <= T) {
            // the same steady body for a wave that owns fewer blocks than its register slice holds (K split over workgroups:
            // the 7B down projection gives its waves 5 or 6 of 8)
            pk_static_for<STEPS>([&](auto e_tag) {
                constexpr int e = decltype(e_tag)::value;
                if (e < ept) {   // wave-uniform
                    uint4_t &w = wq[e & 1];
                    pk_landed(w);
                    const unsigned nsb = next_slot(sb);
                    pk_vmwait<D - 2>();
                    wq[(e + 1) & 1] = lds_block(nsb);
                    if constexpr (I4 && e == 0) {
                        read_scales(sb);
                        dma_next(sb);
                        sb = nsb;
                    } else {
                        half8_t af[(FP8 || WF == PK_F16) ? 1 : SPB];
                        dequant(w, af);
                        dma_next(sb);
                        sb = nsb;
                        mma_block(std::integral_constant<int, (e - SE < 0 ? 0 : e - SE)>{}, w, af);
                    }
                }
            });
            if (ept & 1) wq[0] = wq[1];
            kk += ept;
        } else {
            pk_static_for<STEPS>([&](auto e_tag) {
                constexpr int e = decltype(e_tag)::value;
                if (e < ept) {   // wave-uniform
                    uint4_t w = wq[0];
                    pk_landed(w);
                    half8_t af[(FP8 || WF == PK_F16) ? 1 : SPB];
                    if constexpr (I4 && e == 0) read_scales(sb);
                    else dequant(w, af);
                    const int k = kk + e;
                    if (k + D < T) dma_next(sb);
                    sb = next_slot(sb);
                    if (k + 1 < T) {
                        if (k + 1 + D <= T) pk_vmwait<D - 1>();
                        else pk_vmwait<0>();
                        wq[0] = lds_block(sb);
                    }
                    if constexpr (!(I4 && e == 0)) mma_block(std::integral_constant<int, (e - SE < 0 ? 0 : e - SE)>{}, w, af);
                }
            });
            kk += ept;
        }
    };
    int parity = 0;
    for (int i = 0; i < L; ++i) {
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
        tile_blocks();
        // every element of this wave's stream has been read out of the ring: the next projection's first D elements go in, and
        // stream from HBM under the last reduction, the epilogue stores, the grid barrier and the next prologue
        if (i == L - 1) prefetch_next();
        if (i == 0) {
            PK_STAMP(5);
            norm_finalize();
        }
        publish(acc, i % TPI, parity);
        if (i % TPI == TPI - 1) {
            finish(i / TPI, parity);
            if constexpr (NPAR == 2) parity ^= 1;
        }
        if (i == 0) PK_STAMP(6);
    }
    PK_STAMP(7);
}

template <int MT, int WF, int EPI, int XM>
__global__ __launch_bounds__(512, 2) void pk_mfma_kernel(const PkArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pk_smem_dyn[];
    pk_phase<MT, WF, EPI, XM>(a, static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.y), static_cast<int>(gridDim.x), pk_smem_dyn, false,
                              [](unsigned, int, int) {});
}

// ================= persistent chain: several dependent projections of a decoder layer in ONE launch =================
// Batch decode (4 < batch <= 32) is a chain of weight-streaming projections, each of which needs ALL of its predecessor's output
// (self_decoder.cpp:69-119: attention -> O -> norm -> gate/up -> SwiGLU -> down -> next layer's norm -> QKV).  As separate launches
// every link pays ~7 us that are not streaming (launch ramp, first-byte latency, the activation slice through the CU's address
// path, drain) against 3-15 us of stream; here the links O -> gate/up -> down (-> slab reduce) -> next QKV run as phases of one
// launch on one workgroup per CU, separated by grid barriers, and -- what makes the barrier cheaper than the launch it replaces --
// every wave issues the NEXT phase's first D ring elements (12 KiB per wave, 96 KiB per CU, 24 MiB chip-wide) as soon as its own
// stream has left the ring: the HBM stream runs through the last reduction, the barrier and the next prologue's activation loads.
// Phases use pk_phase unchanged (same arithmetic, same order: results are bit-identical to the launch sequence).
//
// Grid barrier (MI355X guide, "barrier-xcd" form without the placement assumption): workgroups arrive in groups of 32 on a counter
// of their group (one 128-byte line each), the last of a group on the top counter, the last of all stores the epoch into every
// group's generation word, which the group's workgroups poll (relaxed agent-scope loads, s_sleep between polls).  Producer side:
// every wave drains its stores, workgroup barrier, one lane's agent-scope release + drained wait, then the arrive; consumer
// side: one agent-scope acquire + wait behind the poll, workgroup barrier, then plain loads.  Counters are monotonic inside a
// launch and start from zero: the LAST workgroup to leave the kernel (a `done` counter every workgroup adds to behind its last
// barrier) zeroes the block again, so a replayed hipGraph needs no memset node (measured: a captured hipMemsetAsync in front of the
// step did not re-zero the counters on the second replay -- barriers fell through, wrong results, no timeout); the owner zeroes
// the block once at create and after a reported error.  Every spin is bounded: on expiry
// the lane sets the decoder's error word and the workgroup leaves the kernel (later barriers see the word and leave at once), so
// a residency mistake fails llmie_decoder_status() in a test instead of hanging the GPU.
constexpr int PK_SYNC_WORDS = (8 + 1 + 8 + 1) * 32;   // arrive[8] | top | gen[8] | done, one 128-byte line each
constexpr int PK_CHAIN_MAX_PHASES = 5;
enum : int { PK_PH_NONE = -1, PK_PH_PLAIN = 0, PK_PH_SWIGLU = 1, PK_PH_SLAB = 2, PK_PH_REDUCE = 3 };
struct PkChainPhase {
    PkArgs a;        // REDUCE: M, N, slab, scale (row scales or null), residual, y, res_x32, y_x32 are used
    int kind;        // PK_PH_*
    int gx, ks;      // launch plan of the projection: gx workgroups along N x ks slices of K (workgroup w -> (w % gx, w / gx))
    int scale_f32;   // REDUCE: a.scale holds fp32 row scales (e4m3 weights) instead of fp16 ones
};
struct PkChainArgs {
    PkChainPhase ph[PK_CHAIN_MAX_PHASES];   // fixed slots, see pk_chain_kernel
    int nph;
    unsigned *sync;   // PK_SYNC_WORDS zeroed words of this launch
    unsigned *err;    // the decoder's device error word (0 = fine)
    unsigned long long *stamps;   // diagnostic (null in the product path): [256 workgroups][16] s_memrealtime values at the phase edges
    int flag_off;     // byte offset, inside the dynamic LDS, of the barrier's broadcast word (behind every phase's carve; no static
                      // __shared__ object: one would shift the 16-byte alignment of the dynamic region)
};

typedef __attribute__((address_space(1))) unsigned pk_gu32;
__device__ __forceinline__ bool pk_grid_barrier(unsigned *sync, unsigned *err, const unsigned epoch, const int wg, const int nwg, volatile int *bar_ok) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its stores (and ring prefetch) are out
    __syncthreads();
    if (threadIdx.x == 0) {
        pk_gu32 *arrive = (pk_gu32 *)sync, *top = (pk_gu32 *)(sync + 8 * 32), *gen = (pk_gu32 *)(sync + 9 * 32), *e = (pk_gu32 *)err;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the compiler may drop the fence's own wait: keep this one)
        const int g = wg >> 5, ng = (nwg + 31) >> 5, gsize = min(32, nwg - (g << 5));
        bool ok = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
        if (ok) {
            const unsigned old = __hip_atomic_fetch_add(arrive + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == static_cast<unsigned>(gsize) * epoch - 1u) {
                const unsigned t = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t == static_cast<unsigned>(ng) * epoch - 1u)
                    for (int i = 0; i < ng; ++i) __hip_atomic_store(gen + i * 32, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
            while (__hip_atomic_load(gen + g * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) {   // 1 s: some workgroup is not resident / never arrived
                    __hip_atomic_store(e, 0xBA221E20u + epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *bar_ok = ok ? 1 : 0;
    }
    __syncthreads();
    return *bar_ok != 0;
}

// Fixed slot layout (compile-time slot indices keep every descriptor field a plain kernel-argument load, re-readable anywhere,
// instead of ~30 scalar registers held for a whole phase -- with a run-time slot index the 32-row int8 / int4 forms spilled):
//   slot 0 PLAIN (O)  |  1 SWIGLU (gate/up)  |  2 SLAB or PLAIN (down)  |  3 REDUCE (behind a SLAB)  |  4 PLAIN (next layer's QKV)
// kind PK_PH_NONE = slot not used (the chain then ends / starts elsewhere); a barrier follows a slot when a later one is used.
template <int MT, int WF, int SLOT, int EPI, int NEXT>
__device__ __forceinline__ void pk_chain_slot(const PkChainArgs &c, unsigned char *smem, const int wg, const bool prefetched) {
    const int gx = c.ph[SLOT].gx;
    const int by = wg / gx;
    const int bx = by < c.ph[SLOT].ks ? wg - by * gx : c.ph[SLOT].a.units;   // workgroups beyond the plan: no unit
    // the next streaming slot's ring fill: its descriptor is read (from the kernel arguments) only when a wave's own stream has
    // left the ring -- nothing of it is live during the phase
    auto fill = [&c, wg](const unsigned ring_lds, const int wave, const int lane) {
        if constexpr (NEXT >= 0) {
            if (c.ph[NEXT].kind != PK_PH_NONE) {
                const int ngx = c.ph[NEXT].gx, nby = wg / ngx;
                const int nbx = nby < c.ph[NEXT].ks ? wg - nby * ngx : c.ph[NEXT].a.units;
                PkStream<WF> ns;
                ns.init(c.ph[NEXT].a, c.ph[NEXT].kind == PK_PH_SWIGLU ? 2 : 1, nbx, nby, ngx, wave);
                for (int k = 0; k < pk_ring_depth(0) && k < ns.T; ++k) ns.next(ring_lds, k * 1024, lane);
            }
        }
    };
    pk_phase<MT, WF, EPI, 1>(c.ph[SLOT].a, bx, by, gx, smem, prefetched, fill);
}

template <int MT, int WF>
__global__ __launch_bounds__(512, 2) void pk_chain_kernel(const PkChainArgs c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pk_smem_dyn[];
    const int wg = static_cast<int>(blockIdx.x), nwg = static_cast<int>(gridDim.x);
    volatile int *bar_ok = reinterpret_cast<volatile int *>(pk_smem_dyn + c.flag_off);
    unsigned epoch = 0;
    int nstamp = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if (c.stamps && threadIdx.x == 0 && nstamp < 16) c.stamps[wg * 16 + nstamp] = __builtin_amdgcn_s_memrealtime();
        ++nstamp;
    };
    // behind the last barrier of the launch: the last workgroup to get here puts the barrier words back to zero
    auto leave = [&]() __attribute__((always_inline)) {
        stamp();
        if (epoch > 0 && threadIdx.x == 0) {
            pk_gu32 *w = (pk_gu32 *)c.sync;
            if (__hip_atomic_fetch_add(w + 17 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == static_cast<unsigned>(nwg) - 1u)
                for (int i = 0; i < 18; ++i) __hip_atomic_store(w + i * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    stamp();
    bool pre = false;   // the running slot's first ring elements were issued by its predecessor
    const bool u0 = c.ph[0].kind != PK_PH_NONE, u1 = c.ph[1].kind != PK_PH_NONE, u2 = c.ph[2].kind != PK_PH_NONE,
               u3 = c.ph[3].kind != PK_PH_NONE, u4 = c.ph[4].kind != PK_PH_NONE;
    if (u0) {
        pk_chain_slot<MT, WF, 0, PK_EPI_PLAIN, 1>(c, pk_smem_dyn, wg, false);
        pre = u1;
        stamp();
        if ((u1 || u2 || u3 || u4) && !pk_grid_barrier(c.sync, c.err, ++epoch, wg, nwg, bar_ok)) return;
        stamp();
    }
    if (u1) {
        pk_chain_slot<MT, WF, 1, PK_EPI_SWIGLU, 2>(c, pk_smem_dyn, wg, pre);
        pre = u2;
        stamp();
        if ((u2 || u3 || u4) && !pk_grid_barrier(c.sync, c.err, ++epoch, wg, nwg, bar_ok)) return;
        stamp();
    }
    if (u2) {
        if (c.ph[2].kind == PK_PH_SLAB) pk_chain_slot<MT, WF, 2, PK_EPI_SLAB, 4>(c, pk_smem_dyn, wg, pre);
        else pk_chain_slot<MT, WF, 2, PK_EPI_PLAIN, 4>(c, pk_smem_dyn, wg, pre);
        pre = u4;
        stamp();
        if ((u3 || u4) && !pk_grid_barrier(c.sync, c.err, ++epoch, wg, nwg, bar_ok)) return;
        stamp();
    }
    if (u3) {
        // y = scale * sum of the K-slice slabs (+ residual): pk_slab_reduce_kernel's arithmetic, 4 columns per thread
        const PkArgs &a = c.ph[3].a;
        const size_t total4 = static_cast<size_t>(a.M) * a.N / 4, slab_sz = static_cast<size_t>(a.M) * a.N;
        for (size_t i = static_cast<size_t>(wg) * 512 + threadIdx.x; i < total4; i += static_cast<size_t>(nwg) * 512) {
            const size_t e0 = i * 4;
            const int n = static_cast<int>(e0 % a.N), m = static_cast<int>(e0 / a.N);
            floatx4 v = *reinterpret_cast<const floatx4 *>(a.slab + e0);
            for (int k = 1; k < c.ph[3].ks; ++k) v += *reinterpret_cast<const floatx4 *>(a.slab + k * slab_sz + e0);
            half4_t o, res{0, 0, 0, 0};
            if (a.residual) res = *reinterpret_cast<const half4_t *>(a.residual + (a.res_x32 ? x32_offset(m, n) : e0));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float f = v[e];
                if (a.scale) f *= c.ph[3].scale_f32 ? reinterpret_cast<const float *>(a.scale)[n + e] : to_f32(reinterpret_cast<const half_t *>(a.scale)[n + e]);
                if (a.residual) f += to_f32(res[e]);
                o[e] = from_f32<half_t>(f);
            }
            *reinterpret_cast<half4_t *>(a.y + (a.y_x32 ? x32_offset(m, n) : e0)) = o;
        }
        stamp();
        if (u4 && !pk_grid_barrier(c.sync, c.err, ++epoch, wg, nwg, bar_ok)) return;
        stamp();
    }
    if (u4) pk_chain_slot<MT, WF, 4, PK_EPI_PLAIN, -1>(c, pk_smem_dyn, wg, pre);
    leave();
}

// ---- packers: row-major weights of the reference layout -> tile-packed image ----
// One thread per 16-byte chunk of the image.  swiglu != 0: W is a fused gate_up matrix [2I, K]; packed tile 2p holds gate
// rows [16p, 16p + 16), tile 2p + 1 the matching up rows (I + 16p ...), so a work unit streams one contiguous 2-tile run.
// Rows past N are zero.
__device__ __forceinline__ int pk_src_row(int tile, int r, int N, int swiglu) {
    if (!swiglu) return tile * 16 + r;
    const int inter = N >> 1, p = tile >> 1;
    const int i = p * 16 + r;
    return i < inter ? ((tile & 1) ? inter + i : i) : N;  // N = "no such row"
}

template <int WF>
__global__ __launch_bounds__(256) void pk_pack_kernel(const unsigned char *__restrict__ src, uint4_t *__restrict__ dst, int N,
                                                      int K, int tiles, int swiglu) {
    using F = PkFmt<WF>;
    const int nblk = K / F::KB;
    const size_t total = static_cast<size_t>(tiles) * nblk * 64;
    for (size_t c = blockIdx.x * 256ull + threadIdx.x; c < total; c += static_cast<size_t>(gridDim.x) * 256) {
        const int l = static_cast<int>(c & 63), r = l & 15, q = l >> 4;
        const size_t tb = c >> 6;
        const int j = static_cast<int>(tb % nblk), tile = static_cast<int>(tb / nblk);
        const int row = pk_src_row(tile, r, N, swiglu);
        uint4_t out{0u, 0u, 0u, 0u};
        if (row < N) {
            if constexpr (WF == PK_F16) {
                out = *reinterpret_cast<const uint4_t *>(src + (static_cast<size_t>(row) * K + 32 * j + 8 * q) * 2);
            } else if constexpr (WF == PK_I8 || WF == PK_FP8) {
                const unsigned char *p = src + static_cast<size_t>(row) * K + 64 * j + 8 * q;
                const uint2 a = *reinterpret_cast<const uint2 *>(p), b = *reinterpret_cast<const uint2 *>(p + 32);
                out = uint4_t{a.x, a.y, b.x, b.y};
            } else {
                // source: two nibbles per byte, low nibble = even k.  word s: k = 128 j + 32 s + 8 q + (0..7) = 4 source bytes
                const unsigned char *p = src + (static_cast<size_t>(row) * K + 128 * j + 8 * q) / 2;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const unsigned int v = *reinterpret_cast<const unsigned int *>(p + 16 * s);  // nibble i of v = k + i
                    unsigned int o = 0;
                    // image nibble order (n0,n4,n1,n5,n2,n6,n3,n7) = k (0,1,2,3,4,5,6,7): image nibble pos[i] holds k = i
                    constexpr int pos[8] = {0, 4, 1, 5, 2, 6, 3, 7};
#pragma unroll
                    for (int i = 0; i < 8; ++i) o |= ((v >> (4 * i)) & 0xFu) << (4 * pos[i]);
                    out[s] = o;
                }
            }
        }
        dst[c] = out;
    }
}

// tile-packed image -> fp16 row-major [N, K] with the format's scales applied (the inverse of pk_pack_kernel followed by the
// de-quantisation of quant_linear.hip's dequant_f16_kernel: fp16(code * scale), one rounding).  One thread per 16-byte image chunk.
// Used by the prefill of LLMIE_DEC_PACKED_ONLY engines (the row-major matrices are gone): one matrix at a time into the prefill
// workspace, then the fp16 GEMM.  scale: int8 fp16 [N]; int4 fp16 [N, K / 128] (the caller's arrays, which stay referenced).
template <int WF>
__global__ __launch_bounds__(256) void pk_unpack_f16_kernel(const uint4_t *__restrict__ src, const half_t *__restrict__ scale,
                                                            half_t *__restrict__ w16, int N, int K, int tiles, int swiglu) {
    using F = PkFmt<WF>;
    const int nblk = K / F::KB;
    const size_t total = static_cast<size_t>(tiles) * nblk * 64;
    for (size_t c = blockIdx.x * 256ull + threadIdx.x; c < total; c += static_cast<size_t>(gridDim.x) * 256) {
        const int l = static_cast<int>(c & 63), r = l & 15, q = l >> 4;
        const size_t tb = c >> 6;
        const int j = static_cast<int>(tb % nblk), tile = static_cast<int>(tb / nblk);
        const int row = pk_src_row(tile, r, N, swiglu);
        if (row >= N) continue;
        const uint4_t v = src[c];
        half_t *dst = w16 + static_cast<size_t>(row) * K;
        if constexpr (WF == PK_F16) {
            *reinterpret_cast<uint4_t *>(dst + 32 * j + 8 * q) = v;
        } else if constexpr (WF == PK_I8) {
            const float sc = to_f32(scale[row]);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                half8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned w = v[2 * s2 + (e >> 2)];
                    o[e] = from_f32<half_t>(static_cast<float>(static_cast<int8_t>((w >> (8 * (e & 3))) & 0xffu)) * sc);
                }
                *reinterpret_cast<half8_t *>(dst + 64 * j + 32 * s2 + 8 * q) = o;
            }
        } else {
            static_assert(WF == PK_I4, "unpack: fp16 / int8 / int4 images");
            const float sc = to_f32(scale[static_cast<size_t>(row) * (K / 128) + j]);
            constexpr int pos[8] = {0, 4, 1, 5, 2, 6, 3, 7};   // image nibble pos[i] holds k = i (pk_pack_kernel)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                half8_t o;
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = from_f32<half_t>(static_cast<float>(static_cast<int>((v[s2] >> (4 * pos[i])) & 0xfu) - 8) * sc);
                *reinterpret_cast<half8_t *>(dst + 128 * j + 32 * s2 + 8 * q) = o;
            }
        }
    }
}

// int4 group-128 scales [N, K/128] fp16 -> [tiles][nblk][16 rows] (one 32-byte record per block; lane group q reads 8 bytes)
static __global__ __launch_bounds__(256) void pk_pack_scale4_kernel(const half_t *__restrict__ src, half_t *__restrict__ dst, int N,
                                                                  int K, int tiles, int swiglu) {
    const int nblk = K / 128;
    const size_t total = static_cast<size_t>(tiles) * nblk * 16;
    for (size_t c = blockIdx.x * 256ull + threadIdx.x; c < total; c += static_cast<size_t>(gridDim.x) * 256) {
        const int r = static_cast<int>(c & 15);
        const size_t tb = c >> 4;
        const int j = static_cast<int>(tb % nblk), tile = static_cast<int>(tb / nblk);
        const int row = pk_src_row(tile, r, N, swiglu);
        dst[c] = row < N ? src[static_cast<size_t>(row) * nblk + j] : static_cast<half_t>(0.f);
    }
}

// y[m, n] = scale(n, m) * sum_ks slab[ks][m][n] (+ bias[n]) (+ residual[m, n]); 4 columns per thread, every slab load in flight
static __global__ __launch_bounds__(256) void pk_slab_reduce_kernel(const float *__restrict__ slab, int KS, int M, int N,
                                                                  const half_t *__restrict__ wscale_h, const float *__restrict__ wscale_f,
                                                                  const half_t *__restrict__ bias, const half_t *residual, half_t *y,
                                                                  int res_x32, int y_x32) {
    const size_t total4 = static_cast<size_t>(M) * N / 4, slab_sz = static_cast<size_t>(M) * N;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total4; i += static_cast<size_t>(gridDim.x) * 256) {
        const size_t e0 = i * 4;
        const int n = static_cast<int>(e0 % N);
        floatx4 v = *reinterpret_cast<const floatx4 *>(slab + e0);
        for (int k = 1; k < KS; ++k) v += *reinterpret_cast<const floatx4 *>(slab + k * slab_sz + e0);
        half4_t o, res{0, 0, 0, 0};
        const int m = static_cast<int>(e0 / N);
        if (residual) res = *reinterpret_cast<const half4_t *>(residual + (res_x32 ? x32_offset(m, n) : e0));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float f = v[e];
            if (wscale_h) f *= to_f32(wscale_h[n + e]);
            if (wscale_f) f *= wscale_f[n + e];
            if (bias) f += to_f32(bias[n + e]);
            if (residual) f += to_f32(res[e]);
            o[e] = from_f32<half_t>(f);
        }
        *reinterpret_cast<half4_t *>(y + (y_x32 ? x32_offset(m, n) : e0)) = o;
    }
}

// row-major [M, K] fp16 <-> x32 (to_x32 != 0: rows >= M of the image are zero-filled); one thread per 8 halves
static __global__ __launch_bounds__(256) void x32_convert_kernel(const half_t *__restrict__ src, half_t *__restrict__ dst, int M, int K,
                                                              int to_x32) {
    const size_t total = static_cast<size_t>(32) * (K >> 3);
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * 256) {
        const int m = static_cast<int>(i / (K >> 3)), k = static_cast<int>(i % (K >> 3)) * 8;
        if (to_x32) {
            uint4_t v{0u, 0u, 0u, 0u};
            if (m < M) v = *reinterpret_cast<const uint4_t *>(src + static_cast<size_t>(m) * K + k);
            *reinterpret_cast<uint4_t *>(dst + x32_offset(m, k)) = v;
        } else if (m < M) {
            *reinterpret_cast<uint4_t *>(dst + static_cast<size_t>(m) * K + k) = *reinterpret_cast<const uint4_t *>(src + x32_offset(m, k));
        }
    }
}

}  // namespace llmie

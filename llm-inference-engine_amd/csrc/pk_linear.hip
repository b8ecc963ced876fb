// Host side of the packed-weight batch-decode projections (pk_gemm.cuh): packers, launch planning, C ABI.
#include "pk_gemm.cuh"
#include "llmie_internal.h"

namespace llmie {

static int pk_code(llmie_weight_format fmt) {
    switch (fmt) {
        case LLMIE_W_F16: return PK_F16;
        case LLMIE_W_INT8: return PK_I8;
        case LLMIE_W_INT4: return PK_I4;
        case LLMIE_W_FP8: return PK_FP8;
        default: return 0;
    }
}
static int pk_kb(int wf) { return wf == PK_F16 ? 32 : (wf == PK_I4 ? 128 : 64); }
static int pk_xblk(int wf) { return wf == PK_F16 ? 16 : (wf == PK_I4 ? 4 : 8); }
static int pk_bits(int wf) { return wf == PK_F16 ? 16 : (wf == PK_I4 ? 4 : 8); }

int pk_tiles(int N, int swiglu) { return swiglu ? 2 * ((N / 2 + 15) / 16) : (N + 15) / 16; }

size_t pk_packed_bytes(int wf, int N, int K, int swiglu) {
    if (N <= 0 || K <= 0 || K % pk_kb(wf)) return 0;
    return static_cast<size_t>(pk_tiles(N, swiglu)) * (K / pk_kb(wf)) * 1024;
}
// int4: packed group-128 scales, one 32-byte record per (tile, block)
size_t pk_packed_scale_bytes(int wf, int N, int K, int swiglu) {
    if (wf != PK_I4 || K % 128) return 0;
    // [tiles][K / 128][16 rows] fp16 + 256 bytes: the kernel fetches a wave's records as one 256-byte DMA, which may run past the
    // last record of the image
    return static_cast<size_t>(pk_tiles(N, swiglu)) * (K / 128) * 32 + 256;
}

int pk_pack(int wf, const void *src, const void *src_scale, void *dst, void *dst_scale, int N, int K, int swiglu, hipStream_t st) {
    if (!pk_packed_bytes(wf, N, K, swiglu) || (swiglu && (N % 2)) || (reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) % 16) {
        set_error("pack_weight: needs K %% %d == 0 and 16-byte aligned buffers (N=%d K=%d)", pk_kb(wf), N, K);
        return LLMIE_ERR_UNSUPPORTED;
    }
    const int tiles = pk_tiles(N, swiglu);
    const size_t chunks = static_cast<size_t>(tiles) * (K / pk_kb(wf)) * 64;
    const int grid = static_cast<int>(chunks / 256 > 4096 ? 4096 : (chunks + 255) / 256);
    const unsigned char *s = static_cast<const unsigned char *>(src);
    uint4_t *d = static_cast<uint4_t *>(dst);
    switch (wf) {
        case PK_F16: pk_pack_kernel<PK_F16><<<grid, 256, 0, st>>>(s, d, N, K, tiles, swiglu); break;
        case PK_I8: pk_pack_kernel<PK_I8><<<grid, 256, 0, st>>>(s, d, N, K, tiles, swiglu); break;
        case PK_FP8: pk_pack_kernel<PK_FP8><<<grid, 256, 0, st>>>(s, d, N, K, tiles, swiglu); break;
        case PK_I4:
            if (!src_scale || !dst_scale) {
                set_error("pack_weight(int4): group scales missing");
                return LLMIE_ERR_INVALID_ARG;
            }
            pk_pack_kernel<PK_I4><<<grid, 256, 0, st>>>(s, d, N, K, tiles, swiglu);
            pk_pack_scale4_kernel<<<256, 256, 0, st>>>(static_cast<const half_t *>(src_scale), static_cast<half_t *>(dst_scale), N, K, tiles, swiglu);
            break;
        default: set_error("pack_weight: unknown format"); return LLMIE_ERR_UNSUPPORTED;
    }
    return launch_status("pack_weight");
}

// Launch plan: K slices (each wave of the 8 holds at most XBLK blocks of its slice in registers) x workgroups along N.
// One 512-thread workgroup per CU (its register budget admits no second one), so gx * KS <= CUs and the makespan is
// iters * blocks-per-slice; more slices than needed only add slab traffic.
struct PkPlan {
    int KS, bps, gx;
};
static bool pk_plan(int wf, int K, int units, bool may_split, PkPlan *p) {
    constexpr int CUS = 256;
    const int nblk = K / pk_kb(wf), cap = 8 * pk_xblk(wf);
    const int ks_min = (nblk + cap - 1) / cap;
    if (!may_split && ks_min > 1) return false;
    int best_cost = 1 << 30;
    // K is split over workgroups only where the register-resident slice forces it (then the best split >= the minimum is
    // taken); a shape that fits unsplit never needs a slab workspace
    for (int ks = ks_min; ks <= (may_split && ks_min > 1 ? 8 : 1) && ks <= nblk; ++ks) {
        const int bps = (nblk + ks - 1) / ks;
        if ((nblk + bps - 1) / bps != ks) continue;  // every slice non-empty
        const int gx = units < CUS / ks ? units : CUS / ks;
        if (gx < 1) continue;
        const int iters = (units + gx - 1) / gx;
        if (iters > PK_MAX_UNITS) continue;  // epilogue operand tables hold PK_MAX_UNITS units per workgroup
        const int cost = iters * bps * 8 + (ks - 1);  // stream time, then fewer slabs
        if (cost < best_cost) {
            best_cost = cost;
            *p = PkPlan{ks, bps, gx};
        }
    }
    return best_cost != (1 << 30);
}

int pk_unpack_f16(int wf, const void *packed, const half_t *scale, half_t *w16, int N, int K, int swiglu, hipStream_t st) {
    if (!pk_packed_bytes(wf, N, K, swiglu) || wf == PK_FP8 || (wf != PK_F16 && !scale) || (wf == PK_I4 && K % 128) ||
        (reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(w16)) % 16) {
        set_error("unpack_weight: fp16 / int8 / int4 images with their scales, 16-byte aligned buffers (N=%d K=%d)", N, K);
        return LLMIE_ERR_UNSUPPORTED;
    }
    const int tiles = pk_tiles(N, swiglu);
    const size_t chunks = static_cast<size_t>(tiles) * (K / pk_kb(wf)) * 64;
    const int grid = static_cast<int>(chunks / 256 > 8192 ? 8192 : (chunks + 255) / 256);
    const uint4_t *s = static_cast<const uint4_t *>(packed);
    if (wf == PK_F16) pk_unpack_f16_kernel<PK_F16><<<grid, 256, 0, st>>>(s, scale, w16, N, K, tiles, swiglu);
    else if (wf == PK_I8) pk_unpack_f16_kernel<PK_I8><<<grid, 256, 0, st>>>(s, scale, w16, N, K, tiles, swiglu);
    else pk_unpack_f16_kernel<PK_I4><<<grid, 256, 0, st>>>(s, scale, w16, N, K, tiles, swiglu);
    return launch_status("unpack_weight");
}

int x32_convert(const half_t *src, half_t *dst, int M, int K, int to_x32, hipStream_t st) {
    if (M < 1 || M > 32 || K <= 0 || K % 32 || (reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) % 16) {
        set_error("x32_convert: needs 1 <= M <= 32, K %% 32 == 0, 16-byte aligned buffers");
        return LLMIE_ERR_UNSUPPORTED;
    }
    const int grid = (32 * (K / 8) + 255) / 256;
    x32_convert_kernel<<<grid > 1024 ? 1024 : grid, 256, 0, st>>>(src, dst, M, K, to_x32);
    return launch_status("x32_convert");
}

bool pk_eligible(int wf, int M, int K, int N, int epi) {
    if (M < 1 || M > 32 || K % pk_kb(wf) || K < 512) return false;  // a wave's activation window is XBLK blocks = 512 k wide
    if (epi == PK_EPI_SWIGLU && (N % 2 || (N / 2) % 4)) return false;
    if (epi != PK_EPI_SWIGLU && N % 4) return false;
    PkPlan p;
    return pk_plan(wf, K, 1, epi != PK_EPI_SWIGLU, &p);
}

// K slices this shape will use (1 = epilogue in registers, no slabs); 0 if not eligible
int pk_slices(int wf, int M, int K, int N, int epi, bool norm) {
    if (!pk_eligible(wf, M, K, N, epi)) return 0;
    PkPlan p;
    const int units = epi == PK_EPI_SWIGLU ? pk_tiles(N, 1) / 2 : pk_tiles(N, 0);
    if (!pk_plan(wf, K, units, epi != PK_EPI_SWIGLU && !norm, &p)) return 0;
    return p.KS;
}
size_t pk_slab_floats(int wf, int M, int K, int N) {
    const int ks = pk_slices(wf, M, K, N, PK_EPI_PLAIN, false);
    return ks > 1 ? static_cast<size_t>(ks) * M * N : 0;
}

template <int MT, int WF, int EPI, int XM> static void pk_launch_x(const PkArgs &a, const PkPlan &p, hipStream_t st) {
    // reduction slots + per-wave DMA ring + epilogue tables + norm staging (pk_lds); the opt-in limit is registered once
    constexpr int lds_max = 160 * 1024;
    static const bool attr_set = [] {   // once per process, thread-safe (function-local static initialisation)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pk_mfma_kernel<MT, WF, EPI, XM>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        return true;
    }();
    (void)attr_set;
    const int lds = pk_lds(MT, EPI, a.gamma ? (a.pre_bias ? 2 * a.K : a.K) : 0, EPI == PK_EPI_PLAIN && a.residual != nullptr).total;
    pk_mfma_kernel<MT, WF, EPI, XM><<<dim3(p.gx, p.KS), 512, lds, st>>>(a);
}
template <int MT, int WF, int EPI> static void pk_launch_t(const PkArgs &a, const PkPlan &p, hipStream_t st) {
    if (!a.x_x32) {
        pk_launch_x<MT, WF, EPI, 0>(a, p, st);
        return;
    }
    pk_launch_x<MT, WF, EPI, 1>(a, p, st);
}
template <int WF> static void pk_launch_f(int mt, int epi, const PkArgs &a, const PkPlan &p, hipStream_t st) {
    if (mt == 1) {
        if (epi == PK_EPI_SWIGLU) pk_launch_t<1, WF, PK_EPI_SWIGLU>(a, p, st);
        else if (epi == PK_EPI_SLAB) pk_launch_t<1, WF, PK_EPI_SLAB>(a, p, st);
        else pk_launch_t<1, WF, PK_EPI_PLAIN>(a, p, st);
    } else {
        if (epi == PK_EPI_SWIGLU) pk_launch_t<2, WF, PK_EPI_SWIGLU>(a, p, st);
        else if (epi == PK_EPI_SLAB) pk_launch_t<2, WF, PK_EPI_SLAB>(a, p, st);
        else pk_launch_t<2, WF, PK_EPI_PLAIN>(a, p, st);
    }
}

// y = [swiglu]( rmsnorm(x + pre_bias) * gamma . W^T ) (+ residual) on a packed image.
// epi: PK_EPI_PLAIN or PK_EPI_SWIGLU.  K longer than the 8 waves' register budget is split over workgroups: fp32 slabs in
// `slab_ws` (>= pk_slab_floats floats, caller-owned) and a reduce launch (no fused norm on that route).
int pk_linear(int wf, const half_t *x, const void *Wp, const void *scale, half_t *y, int M, int K, int N, int epi, int x32_flags,
              const half_t *residual, const half_t *gamma, const half_t *pre_bias, float eps, float *slab_ws, size_t slab_ws_floats,
              hipStream_t st) {
    if (!pk_eligible(wf, M, K, N, epi) || (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(Wp) | reinterpret_cast<uintptr_t>(gamma) |
                                            reinterpret_cast<uintptr_t>(pre_bias)) % 16 ||
        reinterpret_cast<uintptr_t>(y) % 8 || reinterpret_cast<uintptr_t>(residual) % 8 ||
        (wf != PK_F16 && !scale) || (gamma && K > PK_NORM_MAX_K) || (gamma && residual) /* LDS: norm staging + residual table */) {
        set_error("linear(packed): unsupported shape / alignment M=%d K=%d N=%d", M, K, N);
        return LLMIE_ERR_UNSUPPORTED;
    }
    const int units = epi == PK_EPI_SWIGLU ? pk_tiles(N, 1) / 2 : pk_tiles(N, 0);
    PkPlan p;
    if (!pk_plan(wf, K, units, epi != PK_EPI_SWIGLU && !gamma, &p)) {
        set_error("linear(packed): K=%d does not fit the register-resident activation slice%s", K, gamma ? " (fused norm: no K split)" : "");
        return LLMIE_ERR_UNSUPPORTED;
    }
    if (pk_lds((M + 15) / 16, epi, gamma ? (pre_bias ? 2 * K : K) : 0, epi == PK_EPI_PLAIN && residual).total > 160 * 1024) {
        set_error("linear(packed): K=%d with this fused norm / residual does not fit the LDS carve", K);
        return LLMIE_ERR_UNSUPPORTED;
    }
    const int x_x32 = x32_flags & PK_X32_X, y_x32 = (x32_flags & PK_X32_Y) ? 1 : 0, res_x32 = (x32_flags & PK_X32_RES) ? 1 : 0;
    if ((x_x32 && K % 32) || (y_x32 && (epi == PK_EPI_SWIGLU ? N / 2 : N) % 32) || (res_x32 && N % 32)) {
        set_error("linear(packed): the x32 activation layout needs a multiple of 32 columns");
        return LLMIE_ERR_UNSUPPORTED;
    }
    PkArgs a{x, static_cast<const unsigned char *>(Wp), M, K, N, units, K / pk_kb(wf), p.bps, y, nullptr, scale, residual, x_x32, y_x32, res_x32,
             gamma, pre_bias, eps};
    int kepi = epi;
    if (p.KS > 1) {
        if (!slab_ws || slab_ws_floats < static_cast<size_t>(p.KS) * M * N || reinterpret_cast<uintptr_t>(slab_ws) % 16) {
            set_error("linear(packed): split-K workspace too small (%zu < %zu floats)", slab_ws_floats, static_cast<size_t>(p.KS) * M * N);
            return LLMIE_ERR_WORKSPACE;
        }
        a.slab = slab_ws;
        kepi = PK_EPI_SLAB;
    }
    const int mt = (M + 15) / 16;
    switch (wf) {
        case PK_F16: pk_launch_f<PK_F16>(mt, kepi, a, p, st); break;
        case PK_I8: pk_launch_f<PK_I8>(mt, kepi, a, p, st); break;
        case PK_FP8: pk_launch_f<PK_FP8>(mt, kepi, a, p, st); break;
        case PK_I4: pk_launch_f<PK_I4>(mt, kepi, a, p, st); break;   // `scale` = the packed group-scale image of llmie_pack_weight
        default: set_error("linear(packed): unknown format"); return LLMIE_ERR_UNSUPPORTED;
    }
    if (p.KS > 1) {
        const size_t total4 = static_cast<size_t>(M) * N / 4;
        const int grid = static_cast<int>((total4 + 255) / 256 > 2048 ? 2048 : (total4 + 255) / 256);
        pk_slab_reduce_kernel<<<grid, 256, 0, st>>>(slab_ws, p.KS, M, N, wf == PK_I8 ? static_cast<const half_t *>(scale) : nullptr,
                                                    wf == PK_FP8 ? static_cast<const float *>(scale) : nullptr, nullptr, residual, y, res_x32, y_x32);
    }
    return launch_status("linear(packed)");
}

// ---- persistent chain launch (pk_chain_kernel): host side ----
// The phases are planned exactly as pk_linear plans the separate launches (same units, K slices, LDS carve), so a phase computes
// what its launch computes; the chain needs every workgroup resident (one 512-thread workgroup per CU on a 256-CU device).
static bool pk_chain_device_ok() {
    static const bool ok = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
        return cus >= 256;
    }();
    return ok;
}
void pk_chain_begin(PkChain *ch, int wf, int M) {
    ch->wf = wf;
    ch->M = M;
    ch->nph = 0;
    ch->lds = 0;
    // (int4 at 17-32 rows: that instantiation's scalar spills reach scratch memory -- VMEM operations the hand-counted waits do not
    // know about: no chain there; tests/test_abi_cpu.py holds every other instantiation to zero scratch)
    ch->ok = pk_chain_device_ok() && M >= 1 && M <= 32 && !(wf == PK_I4 && M > 16);
    ch->stamps = nullptr;
    PkChainArgs &c = *reinterpret_cast<PkChainArgs *>(ch->args);
    for (PkChainPhase &ph : c.ph) {
        ph = PkChainPhase{};
        ph.kind = PK_PH_NONE;
        ph.gx = 1;
        ph.ks = 1;
    }
}
// slot: 0 = O projection (PLAIN), 1 = gate/up (SWIGLU), 2 = down (PLAIN; a K split adds the reduce slot 3), 4 = next QKV (PLAIN)
int pk_chain_add(PkChain *ch, int slot, const half_t *x, const void *Wp, const void *scale, half_t *y, int K, int N, int epi, int x32_flags,
                 const half_t *residual, const half_t *gamma, const half_t *pre_bias, float eps, float *slab_ws, size_t slab_ws_floats) {
    const int wf = ch->wf, M = ch->M;
    PkChainArgs &c = *reinterpret_cast<PkChainArgs *>(ch->args);
    static_assert(sizeof(PkChainArgs) <= sizeof(ch->args), "PkChain::args too small");
    auto fail = [&](const char *why) {
        ch->ok = false;
        set_error("linear(packed chain): %s (M=%d K=%d N=%d)", why, M, K, N);
        return LLMIE_ERR_UNSUPPORTED;
    };
    if (!ch->ok) return fail("chain not available");
    if (!(x32_flags & PK_X32_X)) return fail("every phase reads the x32 activation image");
    if (!pk_eligible(wf, M, K, N, epi) || (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(Wp) | reinterpret_cast<uintptr_t>(gamma) |
                                            reinterpret_cast<uintptr_t>(pre_bias)) % 16 ||
        reinterpret_cast<uintptr_t>(y) % 8 || reinterpret_cast<uintptr_t>(residual) % 8 || (wf != PK_F16 && !scale) ||
        (gamma && K > PK_NORM_MAX_K) || (gamma && residual))
        return fail("unsupported shape / alignment");
    const int units = epi == PK_EPI_SWIGLU ? pk_tiles(N, 1) / 2 : pk_tiles(N, 0);
    PkPlan p;
    if (!pk_plan(wf, K, units, epi != PK_EPI_SWIGLU && !gamma, &p)) return fail("K does not fit the register-resident activation slice");
    const int y_x32 = (x32_flags & PK_X32_Y) ? 1 : 0, res_x32 = (x32_flags & PK_X32_RES) ? 1 : 0;
    if (K % 32 || (y_x32 && (epi == PK_EPI_SWIGLU ? N / 2 : N) % 32) || (res_x32 && N % 32)) return fail("x32 layout needs multiples of 32 columns");
    const bool slot_ok = (slot == 0 && epi == PK_EPI_PLAIN && p.KS == 1) || (slot == 1 && epi == PK_EPI_SWIGLU) || (slot == 2 && epi == PK_EPI_PLAIN) ||
                         (slot == 4 && epi == PK_EPI_PLAIN && p.KS == 1);
    if (!slot_ok || c.ph[slot].kind != PK_PH_NONE) return fail("projection does not fit its chain slot");
    const int mt = (M + 15) / 16;
    ++ch->nph;
    PkChainPhase &ph = c.ph[slot];
    ph.a = PkArgs{x, static_cast<const unsigned char *>(Wp), M, K, N, units, K / pk_kb(wf), p.bps, y, nullptr, scale, residual, 1, y_x32, res_x32,
                  gamma, pre_bias, eps};
    ph.kind = epi == PK_EPI_SWIGLU ? PK_PH_SWIGLU : PK_PH_PLAIN;
    ph.gx = p.gx;
    ph.ks = p.KS;
    ph.scale_f32 = 0;
    int kepi = epi;
    if (p.KS > 1) {
        if (!slab_ws || slab_ws_floats < static_cast<size_t>(p.KS) * M * N || reinterpret_cast<uintptr_t>(slab_ws) % 16) return fail("split-K workspace too small");
        ph.a.slab = slab_ws;
        ph.a.residual = nullptr;
        ph.kind = PK_PH_SLAB;
        kepi = PK_EPI_SLAB;
        ++ch->nph;
        PkChainPhase &rd = c.ph[3];
        rd.a = ph.a;
        rd.a.scale = wf == PK_I8 || wf == PK_FP8 ? scale : nullptr;
        rd.a.residual = residual;
        rd.kind = PK_PH_REDUCE;
        rd.gx = 1;
        rd.ks = p.KS;
        rd.scale_f32 = wf == PK_FP8 ? 1 : 0;
    }
    const int lds = pk_lds(mt, kepi, gamma ? (pre_bias ? 2 * K : K) : 0, kepi == PK_EPI_PLAIN && residual).total;
    if (lds > 160 * 1024 - 16) return fail("LDS carve too large");
    ch->lds = lds > ch->lds ? lds : ch->lds;
    return LLMIE_OK;
}
template <int MT, int WF> static void pk_chain_launch_t(const PkChainArgs &c, int lds, hipStream_t st) {
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pk_chain_kernel<MT, WF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)attr_set;
    pk_chain_kernel<MT, WF><<<256, 512, lds, st>>>(c);
}
int pk_chain_launch(PkChain *ch, unsigned *sync, unsigned *err, hipStream_t st) {
    PkChainArgs &c = *reinterpret_cast<PkChainArgs *>(ch->args);
    if (!ch->ok || ch->nph < 1 || !sync || !err) {
        set_error("linear(packed chain): nothing to launch");
        return LLMIE_ERR_UNSUPPORTED;
    }
    c.nph = ch->nph;
    c.sync = sync;
    c.err = err;
    c.stamps = ch->stamps;
    c.flag_off = (ch->lds + 15) & ~15;
    const int lds = c.flag_off + 16;
    const int mt = (ch->M + 15) / 16;
#define LLMIE_CHAIN(WF_) (mt == 1 ? pk_chain_launch_t<1, WF_>(c, lds, st) : pk_chain_launch_t<2, WF_>(c, lds, st))
    switch (ch->wf) {
        case PK_F16: LLMIE_CHAIN(PK_F16); break;
        case PK_I8: LLMIE_CHAIN(PK_I8); break;
        case PK_FP8: LLMIE_CHAIN(PK_FP8); break;
        case PK_I4: LLMIE_CHAIN(PK_I4); break;
        default: set_error("linear(packed chain): unknown format"); return LLMIE_ERR_UNSUPPORTED;
    }
#undef LLMIE_CHAIN
    return launch_status("linear(packed chain)");
}
size_t pk_chain_sync_bytes() { return PK_SYNC_WORDS * sizeof(unsigned); }

}  // namespace llmie

using namespace llmie;

extern "C" size_t llmie_packed_weight_bytes(llmie_weight_format fmt, int N, int K, int swiglu_pairs) {
    const int wf = pk_code(fmt);
    return wf ? pk_packed_bytes(wf, N, K, swiglu_pairs) : 0;
}
extern "C" size_t llmie_packed_scale_bytes(llmie_weight_format fmt, int N, int K, int swiglu_pairs) {
    const int wf = pk_code(fmt);
    return wf ? pk_packed_scale_bytes(wf, N, K, swiglu_pairs) : 0;
}

extern "C" int llmie_pack_weight(llmie_weight_format fmt, const void *w, const void *scale, void *packed, void *packed_scale, int N, int K,
                                 int swiglu_pairs, llmie_stream stream) {
    LLMIE_REQUIRE(w && packed && N > 0 && K > 0, "pack_weight: bad arguments");
    const int wf = pk_code(fmt);
    if (!wf) LLMIE_UNSUPPORTED("pack_weight: format %d", (int)fmt);
    return pk_pack(wf, w, scale, packed, packed_scale, N, K, swiglu_pairs, as_stream(stream));
}

extern "C" size_t llmie_linear_packed_workspace_bytes(llmie_weight_format fmt, int M, int K, int N) {
    const int wf = pk_code(fmt);
    return wf ? pk_slab_floats(wf, M, K, N) * sizeof(float) : 0;
}

extern "C" int llmie_linear_packed(llmie_weight_format fmt, const void *x, const void *packed, const void *scale, void *y, int M, int K,
                                   int N, int swiglu, int x32_flags, const void *residual, const void *gamma, const void *pre_bias,
                                   float eps, void *workspace, size_t workspace_bytes, llmie_stream stream) {
    LLMIE_REQUIRE(x && packed && y, "linear_packed: NULL pointer");
    LLMIE_REQUIRE(M > 0 && K > 0 && N > 0, "linear_packed: bad shape");
    LLMIE_REQUIRE(!swiglu || !residual, "linear_packed: the SwiGLU form takes no residual");
    const int wf = pk_code(fmt);
    if (!wf) LLMIE_UNSUPPORTED("linear_packed: format %d", (int)fmt);
    return pk_linear(wf, (const half_t *)x, packed, scale, (half_t *)y, M, K, N, swiglu ? PK_EPI_SWIGLU : PK_EPI_PLAIN, x32_flags,
                     (const half_t *)residual, (const half_t *)gamma, (const half_t *)pre_bias, eps, static_cast<float *>(workspace),
                     workspace_bytes / sizeof(float), as_stream(stream));
}

extern "C" size_t llmie_x32_bytes(int K) { return K > 0 && K % 32 == 0 ? static_cast<size_t>(K) * 64 : 0; }
extern "C" int llmie_x32_convert(const void *src, void *dst, int M, int K, int to_x32, llmie_stream stream) {
    LLMIE_REQUIRE(src && dst, "x32_convert: NULL pointer");
    return x32_convert((const half_t *)src, (half_t *)dst, M, K, to_x32, as_stream(stream));
}

#ifdef PK_STAMPS
extern "C" int llmie_debug_stamps(void *dst_host, size_t bytes) {
    return hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(llmie::pk_stamp_buf), bytes) == hipSuccess ? 0 : -1;
}
#endif

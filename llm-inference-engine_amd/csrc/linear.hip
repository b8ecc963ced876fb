// llmie_linear / llmie_batched_gemm: shape dispatch over the kernels in gemm_kernels.cuh.
#include "gemm_kernels.cuh"
#include "gemm256.cuh"
#include "gemm8p.cuh"
#include "gemm_mid.cuh"
#include "llmie_internal.h"

#include <cstdlib>

namespace llmie {

// ---- decode GEMV dispatch ----
// K-split kernel: XC = 16-byte chunks per thread = ceil(K*WBITS/128/256) rounded up to {1,2,3,4,6,8}; RPW rows per
// iteration so that RPW*XC ~ 16 loads are in flight per lane; register budget M*XC*XE <= 16 half8 of activations.
template <int M, int RPW, int XC, int WBITS, bool DB = false, bool FP8 = false, int RI = 1> static void launch_ksplit(const GemvArgs &a, hipStream_t st) {
    const bool swiglu = a.epi == EPI_SWIGLU;
    const int groups = swiglu ? (a.N / 2 + (RPW / 2) * RI - 1) / ((RPW / 2) * RI) : (a.N + RPW * RI - 1) / (RPW * RI);
    // long-lived workgroups (the prologue -- activation slice + norm -- is paid once per workgroup), evenly loaded
    // (measured per format, interleaved A/B of whole decode steps: fp16 and int4 +0.4 .. +2 % with 512 against 768, int8 -0.6 %, fp8 -2 .. -7 %;
    // 384 and 1024 lose everywhere)
    constexpr int target = (WBITS == 16 || WBITS == 4) ? 512 : 768;
    const int iters = (groups + target - 1) / target;
    const int grid = (groups + iters - 1) / iters;
    gemv_ksplit_kernel<M, RPW, XC, WBITS, DB, FP8, RI><<<grid, 256, 0, st>>>(a);
}

static int ksplit_xc(int K, int wbits) { return (K * wbits / 128 + 255) / 256; }

template <int M, int WBITS, bool FP8 = false> static bool dispatch_ksplit(const GemvArgs &a, hipStream_t st) {
    constexpr int XE = WFmt<WBITS>::XE;
    const int xc = ksplit_xc(a.K, WBITS);
    if constexpr (M * 1 * XE <= 16) {
        if (xc <= 1) {  // quantised rows are short: 8 rows per group, double buffered (16 loads in flight per lane)
            if constexpr (WBITS == 4) {
                // rows of at most 2 KiB (K <= 4096): two consecutive rows per workgroup instruction, or half the threads idle
                if (a.K * WBITS / 128 <= 128 && (a.epi != EPI_SWIGLU || (a.N / 2) % 2 == 0)) {
                    launch_ksplit<M, 8, 1, WBITS, true, FP8, 2>(a, st);
                    return true;
                }
            }
            if constexpr (WBITS == 16) launch_ksplit<M, 8, 1, 16>(a, st);
            else launch_ksplit<M, (WBITS == 8 && !FP8) ? 4 : 8, 1, WBITS, true, FP8>(a, st);   // int8: 4 rows per iteration (+2.8 % in A/B); fp8, int4: 8
            return true;
        }
    }
    if constexpr (M * 2 * XE <= 16) {
        if (xc <= 2) {
            if constexpr (WBITS == 16) launch_ksplit<M, 4, 2, 16>(a, st);   // 4 rows x 2 chunks per iteration: A/B against 8 x 2 (+1.6 % batch 1, +4.4 % batch 3) and 2 x 2, 16 x 2
            else launch_ksplit<M, 4, 2, WBITS, true, FP8>(a, st);
            return true;
        }
    }
    if constexpr (M * 3 * XE <= 16 && WBITS != 16) {
        if (xc <= 3) { launch_ksplit<M, 2, 3, WBITS, true, FP8>(a, st); return true; }
    }
    if constexpr (M * 4 * XE <= 16 && WBITS == 16) {
        if (xc <= 4) { launch_ksplit<M, 4, 4, WBITS>(a, st); return true; }
    }
    if constexpr (M * 6 * XE <= 16 && WBITS == 16) {
        if (xc <= 6) { launch_ksplit<M, 4, 6, WBITS>(a, st); return true; }
    }
    if constexpr (M * 8 * XE <= 16 && WBITS == 16) {
        if (xc <= 8) { launch_ksplit<M, 2, 8, WBITS>(a, st); return true; }
    }
    return false;
}

bool ksplit_eligible(int M, int K, int wbits) {
    const int xe = wbits == 16 ? 1 : (wbits == 8 ? 2 : 4);
    const int xc = ksplit_xc(K, wbits);
    if (K % (128 / wbits) != 0) return false;  // whole 16-byte chunks
    int xcr;  // the XC the dispatcher would round to
    if (xc <= 1) xcr = 1;
    else if (xc <= 2) xcr = 2;
    else if (wbits != 16) xcr = xc <= 3 ? 3 : 99;
    else xcr = xc <= 4 ? 4 : (xc <= 6 ? 6 : (xc <= 8 ? 8 : 99));
    return M >= 1 && M <= 8 && M * xcr * xe <= 16;
}

template <int M> static bool dispatch_gemv_m(const GemvArgs &a, hipStream_t st) {
    if (dispatch_ksplit<M, 16>(a, st)) return true;
    if (static_cast<size_t>(M) * a.K * 2 > 64 * 1024) return false;
    const bool swiglu = a.epi == EPI_SWIGLU;
    const int npairs = swiglu ? a.N / 2 : (a.N + 1) / 2;
    int wgs = (npairs + 3) / 4;
    if (wgs > 2048) wgs = 2048;
    gemv_lds_kernel<M><<<wgs, 256, static_cast<size_t>(M) * a.K * sizeof(half_t), st>>>(a);
    return true;
}

static bool dispatch_gemv(int M, const GemvArgs &a, hipStream_t st) {
    switch (M) {
        case 1: return dispatch_gemv_m<1>(a, st);
        case 2: return dispatch_gemv_m<2>(a, st);
        case 3: return dispatch_gemv_m<3>(a, st);
        case 4: return dispatch_gemv_m<4>(a, st);
        case 5: return dispatch_gemv_m<5>(a, st);
        case 6: return dispatch_gemv_m<6>(a, st);
        case 7: return dispatch_gemv_m<7>(a, st);
        case 8: return dispatch_gemv_m<8>(a, st);
        default: return false;
    }
}

// quantised-weight GEMV (M <= 8): true when launched
template <int WBITS> static bool dispatch_gemv_q(int M, const GemvArgs &a, hipStream_t st) {
    switch (M) {
        case 1: return dispatch_ksplit<1, WBITS>(a, st);
        case 2: return dispatch_ksplit<2, WBITS>(a, st);
        case 3: return dispatch_ksplit<3, WBITS>(a, st);
        case 4: return dispatch_ksplit<4, WBITS>(a, st);
        case 5: return dispatch_ksplit<5, WBITS>(a, st);
        case 6: return dispatch_ksplit<6, WBITS>(a, st);
        case 7: return dispatch_ksplit<7, WBITS>(a, st);
        case 8: return dispatch_ksplit<8, WBITS>(a, st);
        default: return false;
    }
}
bool gemv_q_launch(int wbits, int M, const GemvArgs &a, hipStream_t st) {
    return wbits == 8 ? dispatch_gemv_q<8>(M, a, st) : dispatch_gemv_q<4>(M, a, st);
}
// fp8 (e4m3 weights, fp32 row scales in a.scale, activations quantised per token in the prologue)
bool gemv_fp8_launch(int M, const GemvArgs &a, hipStream_t st) {
    switch (M) {
        case 1: return dispatch_ksplit<1, 8, true>(a, st);
        case 2: return dispatch_ksplit<2, 8, true>(a, st);
        case 3: return dispatch_ksplit<3, 8, true>(a, st);
        case 4: return dispatch_ksplit<4, 8, true>(a, st);
        case 5: return dispatch_ksplit<5, 8, true>(a, st);
        case 6: return dispatch_ksplit<6, 8, true>(a, st);
        case 7: return dispatch_ksplit<7, 8, true>(a, st);
        case 8: return dispatch_ksplit<8, 8, true>(a, st);
        default: return false;
    }
}

template <int EPI>
static bool dispatch_skinny(int M, const half_t *x, const half_t *W, half_t *y, int K, int N,
                            const half_t *bias, const half_t *residual, hipStream_t st) {
    constexpr int NW = 8;
    const int mt = (M + 15) / 16;
    if constexpr (EPI == EPI_SWIGLU) {
        const int wgs = (N / 2 + 15) / 16;
        switch (mt) {
            case 1: skinny_mfma_f16_kernel<1, 2, NW, EPI><<<wgs, NW * 64, 0, st>>>(x, W, y, M, K, N, bias, residual); return true;
            case 2: skinny_mfma_f16_kernel<2, 2, NW, EPI><<<wgs, NW * 64, 0, st>>>(x, W, y, M, K, N, bias, residual); return true;
            case 3: skinny_mfma_f16_kernel<3, 2, NW, EPI><<<wgs, NW * 64, 0, st>>>(x, W, y, M, K, N, bias, residual); return true;
            case 4: skinny_mfma_f16_kernel<4, 2, NW, EPI><<<wgs, NW * 64, 0, st>>>(x, W, y, M, K, N, bias, residual); return true;
            default: return false;
        }
    } else {
        const int tiles = (N + 15) / 16;
        switch (mt) {
            case 1: skinny_mfma_f16_kernel<1, 1, NW, EPI><<<tiles, NW * 64, 0, st>>>(x, W, y, M, K, N, bias, residual); return true;
            case 2: skinny_mfma_f16_kernel<2, 2, NW, EPI><<<(tiles + 1) / 2, NW * 64, 0, st>>>(x, W, y, M, K, N, bias, residual); return true;
            case 3: skinny_mfma_f16_kernel<3, 2, NW, EPI><<<(tiles + 1) / 2, NW * 64, 0, st>>>(x, W, y, M, K, N, bias, residual); return true;
            case 4: skinny_mfma_f16_kernel<4, 2, NW, EPI><<<(tiles + 1) / 2, NW * 64, 0, st>>>(x, W, y, M, K, N, bias, residual); return true;
            default: return false;
        }
    }
}

template <typename T>
static void launch_generic(const T *a, const T *b, T *c, int batch, int M, int N, int K, bool trans_b,
                           const T *bias, const T *residual, hipStream_t st) {
    dim3 grid((N + 63) / 64, (M + 63) / 64, batch);
    const size_t sa = static_cast<size_t>(M) * K, sb = static_cast<size_t>(N) * K, sc = static_cast<size_t>(M) * N;
    if (trans_b)
        generic_gemm_kernel<T, true><<<grid, 256, 0, st>>>(a, b, c, M, N, K, sa, sb, sc, bias, residual);
    else
        generic_gemm_kernel<T, false><<<grid, 256, 0, st>>>(a, b, c, M, N, K, sa, sb, sc, bias, residual);
}

// fp16, W[N,K]: the decode / prefill projection path.  epi selects the fused epilogue; a non-null
// `norm` fuses rmsnorm(x + pre_bias)*gamma in front of the projection (GEMV path only: returns
// LLMIE_ERR_UNSUPPORTED otherwise so the caller can run the norm as its own kernel).
// y = sum over the KS slabs (* scales of the weight format) (+bias)(+residual) | SwiGLU over (n, N/2+n)
static __global__ __launch_bounds__(256) void skinny_finalize_kernel(const float *__restrict__ slab, half_t *y, int M, int N, int KS,
                                                              const SlabScale scale, const half_t *__restrict__ bias,
                                                              const half_t *residual, int epi) {
    const int out_n = epi == EPI_SWIGLU ? N / 2 : N;
    const size_t total = static_cast<size_t>(M) * out_n;
    const size_t slab_sz = static_cast<size_t>(M) * N;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * 256) {
        const int m = static_cast<int>(i / out_n), n = static_cast<int>(i - static_cast<size_t>(m) * out_n);
        auto gather = [&](int col) {
            float v = 0.f;
            for (int k = 0; k < KS; ++k) v += slab[k * slab_sz + static_cast<size_t>(m) * N + col];
            return scale.apply(v, m, col);
        };
        float v;
        if (epi == EPI_SWIGLU) {
            const float gt = gather(n), up = gather(n + out_n);
            v = (gt / (1.0f + expf(-gt))) * up;
        } else {
            v = gather(n);
            if (bias) v += to_f32(bias[n]);
            if (residual) v += to_f32(residual[static_cast<size_t>(m) * N + n]);
        }
        y[static_cast<size_t>(m) * out_n + n] = from_f32<half_t>(v);
    }
}

// The same consumer, four columns per thread and every slab load of them in flight at once (round 3).  skinny_finalize_kernel
// above walks the KS slabs of ONE element with dependent loads -- the slabs were just written by other XCDs, so each is a full
// memory round trip: 7.4 us per launch at 128 x 12288 x 4 slabs, which is mostly 4 serial latencies.  Same summation order
// (s0 + s1 + ...), same arithmetic per element: bit-identical results.  N % 4 == 0 (SwiGLU: N % 8 == 0).
template <bool SWIGLU>
static __global__ __launch_bounds__(256) void splitk_finalize4_kernel(const float *__restrict__ slab, half_t *y, int M, int N, int KS,
                                                                      const SlabScale scale, const half_t *__restrict__ bias,
                                                                      const half_t *residual) {
    const int out_n = SWIGLU ? N / 2 : N, n4 = out_n / 4;
    const size_t total = static_cast<size_t>(M) * n4, slab_sz = static_cast<size_t>(M) * N;
    constexpr int NV = SWIGLU ? 2 : 1;   // column groups per item: the gate columns and the matching up columns
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * 256) {
        const int m = static_cast<int>(i / n4), n = static_cast<int>(i - static_cast<size_t>(m) * n4) * 4;
        floatx4 v[NV];
        for (int k0 = 0; k0 < KS; k0 += 8) {
            floatx4 t[8][NV];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int g = 0; g < NV; ++g)
                    t[kk][g] = *reinterpret_cast<const floatx4 *>(slab + static_cast<size_t>(min(k0 + kk, KS - 1)) * slab_sz +
                                                                  static_cast<size_t>(m) * N + n + g * out_n);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                if (k0 + kk < KS) {
#pragma unroll
                    for (int g = 0; g < NV; ++g) {
                        if (k0 + kk == 0) v[g] = floatx4{0.f, 0.f, 0.f, 0.f} + t[kk][g];
                        else v[g] += t[kk][g];
                    }
                }
        }
        half4_t o4;
        if constexpr (SWIGLU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gt = scale.apply(v[0][e], m, n + e), up = scale.apply(v[1][e], m, n + e + out_n);
                o4[e] = from_f32<half_t>((gt / (1.0f + expf(-gt))) * up);
            }
        } else {
            half4_t b4{0, 0, 0, 0}, r4{0, 0, 0, 0};
            if (bias) b4 = *reinterpret_cast<const half4_t *>(bias + n);
            if (residual) r4 = *reinterpret_cast<const half4_t *>(residual + static_cast<size_t>(m) * N + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = scale.apply(v[0][e], m, n + e);
                if (bias) t += to_f32(b4[e]);
                if (residual) t += to_f32(r4[e]);
                o4[e] = from_f32<half_t>(t);
            }
        }
        *reinterpret_cast<half4_t *>(y + static_cast<size_t>(m) * out_n + n) = o4;
    }
}
static void launch_finalize(const float *slab, half_t *y, int M, int N, int KS, const SlabScale &sc, const half_t *bias, const half_t *residual,
                            int epi, hipStream_t st) {
    const int out_n = epi == EPI_SWIGLU ? N / 2 : N;
    const bool vec = out_n % 4 == 0 && (reinterpret_cast<uintptr_t>(slab) % 16) == 0 &&
                     ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual)) % 8) == 0;
    if (vec) {
        const size_t items = static_cast<size_t>(M) * (out_n / 4);
        int fgrid = static_cast<int>((items + 255) / 256);
        if (fgrid > 4096) fgrid = 4096;
        if (epi == EPI_SWIGLU) splitk_finalize4_kernel<true><<<fgrid, 256, 0, st>>>(slab, y, M, N, KS, sc, bias, residual);
        else splitk_finalize4_kernel<false><<<fgrid, 256, 0, st>>>(slab, y, M, N, KS, sc, bias, residual);
        return;
    }
    const size_t total = static_cast<size_t>(M) * out_n;
    int fgrid = static_cast<int>((total + 255) / 256);
    if (fgrid > 2048) fgrid = 2048;
    skinny_finalize_kernel<<<fgrid, 256, 0, st>>>(slab, y, M, N, KS, sc, bias, residual, epi);
}

// Split-K planning, shared by the launcher and by the workspace-size query (the slabs are CALLER-owned: nothing on the
// compute path allocates, so a first call may already run under hipGraph capture).
constexpr int kSplitKPassRows = 128;   // activation rows per pass (16 per MFMA tile)
struct SplitKPlan {
    int form;   // 0 = 64-weight-row skinny kernel, 1 = 128-row LDS-DMA kernel (gemm_mid.cuh), 2 = its 64-row form
    int ks;     // K slices = slabs
    int per;    // form 1 / 2: K tiles per slice; form 0: sub-blocks per slice
    int wn;     // form 1 / 2: 64-row groups per workgroup (4 = 256 weight rows)
    bool ok;
};
static SplitKPlan splitk_plan(int wbits, int M, int K, int N) {
    SplitKPlan p{0, 1, 0, 2, false};
    const int bk = wbits == 16 ? 128 : (wbits == 4 ? 512 : 256);  // k per sub-block (4 weight loads per lane)
    if (wbits == 4 && (M > 64 || K % 256)) return p;
    if (M < 1 || M > kSplitKPassRows || (wbits != 4 && K % bk) || K < 512 || (wbits != 16 && wbits != 8 && wbits != 4 && wbits != WF_FP8))
        return p;
    p.ok = true;
    // 64 < M <= 128, fp16 or e4m3 operands: 128-row LDS-DMA kernel; 32 < M <= 64: its 64-row form (measured fp16 M=64:
    // gate/up 55.6 -> 44.7 us, qkv 33.0 -> 29.8, down 29.2 -> 25.8 against the skinny kernel; about equal at M = 32)
    const bool mid64 = M <= 64 && M >= 33;
    // (int8 weights, round 3: the same kernel with raw int8 weight tiles; 128 or 256 weight rows per workgroup, from 65 rows)
    if ((wbits == 16 || wbits == WF_FP8 || (wbits == 8 && M >= 65)) && (M >= 65 || mid64) && K % 128 == 0 && N >= 128) {
        const bool fp8 = wbits == WF_FP8;
        p.form = mid64 ? 2 : 1;
        const int KT = K / (fp8 ? 128 : 64);
        auto slices = [&](int wn) {
            const int mtiles = (N + 64 * wn - 1) / (64 * wn);
            int ks = 256 / mtiles;
            ks = ks < 1 ? 1 : (ks > 8 ? 8 : ks);
            if (ks > KT / 4) ks = KT / 4 > 0 ? KT / 4 : 1;
            return ks;
        };
        p.wn = N >= 8192 ? 4 : 2;   // wide N: 256 weight rows per workgroup
        if (N >= 8192) {
            // one workgroup per CU: tiles x slices should come close to 256 -- 192-row tiles where they fill the chip better than
            // 256-row ones (N = 22016: 115 x 2 = 230 against 86 x 2 = 172; N = 12288: 64 x 4 = 256, one slab less than 48 x 5)
            const int t4 = (N + 255) / 256, t3 = (N + 191) / 192;
            if (t3 * slices(3) > t4 * slices(4)) p.wn = 3;
        }
        const int ks = slices(p.wn);
        p.per = (KT + ks - 1) / ks;
        p.ks = (KT + p.per - 1) / p.per;  // every slice non-empty
        return p;
    }
    const int tiles = (N + 63) / 64, total_blocks = (K + bk - 1) / bk;
    // K slices: enough workgroups to fill the chip (512), but >= 4 sub-blocks per slice (the weight ring depth) and as few
    // slabs as possible (slab traffic = 2 * KS * M * N * 4 bytes)
    int KS = 1;
    const int min_blocks = wbits == 4 ? 1 : 4;  // int4 sub-blocks are 512 k wide: K = 4096 has only 8 of them
    while (KS < 16 && tiles * KS < 512 && total_blocks / (KS * 2) >= min_blocks) KS *= 2;
    p.ks = KS;
    p.per = (total_blocks + KS - 1) / KS;
    return p;
}
// fp32 floats of slab workspace a split-K projection of M rows needs (M > 128 runs in passes of 128 rows that reuse it);
// 0 = this shape has no split-K form
size_t linear_splitk_ws_floats(int wbits, int M, int K, int N) {
    const int rows = wbits == 4 ? (M < 64 ? M : 64) : (M < kSplitKPassRows ? M : kSplitKPassRows);
    if (rows < 1) return 0;
    const SplitKPlan p = splitk_plan(wbits, rows, K, N);
    return p.ok ? static_cast<size_t>(p.ks) * rows * N : 0;
}

// split-K skinny MFMA path, first half: partial products of one pass (M <= 128) into the caller's fp32 slabs [KS][M][N]
// (`ws`, >= linear_splitk_ws_floats floats, 16-byte aligned).  The consumer (finalize kernel, splitk_rownorm, or the
// decode attention reading q/k/v straight from the slabs) must be enqueued before the next launch that writes `ws`.
int linear_splitk_partial(int wbits, const void *x, const void *W, int M, int K, int N, hipStream_t st, SplitKSlabs *out,
                          SlabWs ws, const half_t *gscale) {
    // wbits: 16 = fp16 weights, 8 = int8 weights, 4 = int4 weights with group-128 scales `gscale` applied in the kernel (all
    // with fp16 activations), WF_FP8 = e4m3 weights and e4m3 activations
    const SplitKPlan p = splitk_plan(wbits, M, K, N);
    if (wbits == 4 && (!gscale || reinterpret_cast<uintptr_t>(gscale) % 4 || !p.ok)) {
        set_error("linear(split-K int4): needs M <= 64 per pass, K %% 256 == 0, group-128 scales");
        return LLMIE_ERR_UNSUPPORTED;
    }
    if (!p.ok || (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W)) % 16) {
        set_error("linear(split-K): unsupported shape M=%d K=%d (bits=%d)", M, K, wbits);
        return LLMIE_ERR_UNSUPPORTED;
    }
    const size_t need = static_cast<size_t>(p.ks) * M * N;
    if (!ws.p || ws.floats < need || reinterpret_cast<uintptr_t>(ws.p) % 16) {
        set_error("linear(split-K): slab workspace too small or misaligned (%zu < %zu bytes); size it with "
                  "llmie_linear_workspace_bytes()", ws.p ? ws.floats * sizeof(float) : static_cast<size_t>(0), need * sizeof(float));
        return LLMIE_ERR_WORKSPACE;
    }
    if (p.form != 0) {
        const bool fp8 = wbits == WF_FP8, mid64 = p.form == 2;
        const int wn = p.wn, ks = p.ks, per = p.per;
        const int mtiles = (N + 64 * wn - 1) / (64 * wn);
        float *mslab = ws.p;
    static const bool attr_set = [] {   // once per process, thread-safe (function-local static initialisation)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mid_splitk_kernel<false, 4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 16384);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mid_splitk_kernel<true, 4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 16384);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mid_splitk_kernel<false, 2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 16384);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mid_splitk_kernel<true, 2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 16384);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mid_splitk_kernel<false, 3, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 40960);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mid_splitk_kernel<true, 3, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 40960);
            return true;
        }();
        (void)attr_set;
        const dim3 mgrid(mtiles * ks);
        if (wbits == 8) {   // int8 weights: stage = 16 KiB of activations + 16 KiB (192 / 256 rows) or 8 KiB (128 rows) of weights
            static const bool attrq = [] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mid_splitk_kernel<false, 4, 5, 128, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 32768);
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mid_splitk_kernel<false, 3, 5, 128, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 32768);
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mid_splitk_kernel<false, 2, 6, 128, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 24576);
                return true;
            }();
            (void)attrq;
            if (wn == 4) mid_splitk_kernel<false, 4, 5, 128, 8><<<mgrid, 512, 5 * 32768, st>>>(x, W, mslab, M, N, K, ks, per);
            else if (wn == 3) mid_splitk_kernel<false, 3, 5, 128, 8><<<mgrid, 512, 5 * 32768, st>>>(x, W, mslab, M, N, K, ks, per);
            else mid_splitk_kernel<false, 2, 6, 128, 8><<<mgrid, 512, 6 * 24576, st>>>(x, W, mslab, M, N, K, ks, per);
        } else if (mid64) {
            static const bool attr64 = [] {   // once per process, thread-safe (function-local static initialisation)
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mid_splitk_kernel<false, 4, 4, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 40960);
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mid_splitk_kernel<true, 4, 4, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 40960);
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mid_splitk_kernel<false, 2, 6, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 24576);
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mid_splitk_kernel<true, 2, 6, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 24576);
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mid_splitk_kernel<false, 3, 5, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 32768);
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mid_splitk_kernel<true, 3, 5, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 32768);
                return true;
            }();
            (void)attr64;
            if (wn == 4) {
                if (fp8) mid_splitk_kernel<true, 4, 4, 64><<<mgrid, 512, 4 * 40960, st>>>(x, W, mslab, M, N, K, ks, per);
                else mid_splitk_kernel<false, 4, 4, 64><<<mgrid, 512, 4 * 40960, st>>>(x, W, mslab, M, N, K, ks, per);
            } else if (wn == 3) {   // stage = 8 KiB of activations + 24 KiB of weights: five stages fill the 160 KiB exactly
                if (fp8) mid_splitk_kernel<true, 3, 5, 64><<<mgrid, 512, 5 * 32768, st>>>(x, W, mslab, M, N, K, ks, per);
                else mid_splitk_kernel<false, 3, 5, 64><<<mgrid, 512, 5 * 32768, st>>>(x, W, mslab, M, N, K, ks, per);
            } else {
                if (fp8) mid_splitk_kernel<true, 2, 6, 64><<<mgrid, 512, 6 * 24576, st>>>(x, W, mslab, M, N, K, ks, per);
                else mid_splitk_kernel<false, 2, 6, 64><<<mgrid, 512, 6 * 24576, st>>>(x, W, mslab, M, N, K, ks, per);
            }
        } else if (wn == 4) {
            if (fp8) mid_splitk_kernel<true, 4, 3><<<mgrid, 512, 3 * 3 * 16384, st>>>(x, W, mslab, M, N, K, ks, per);
            else mid_splitk_kernel<false, 4, 3><<<mgrid, 512, 3 * 3 * 16384, st>>>(x, W, mslab, M, N, K, ks, per);
        } else if (wn == 3) {
            if (fp8) mid_splitk_kernel<true, 3, 3><<<mgrid, 512, 3 * 40960, st>>>(x, W, mslab, M, N, K, ks, per);
            else mid_splitk_kernel<false, 3, 3><<<mgrid, 512, 3 * 40960, st>>>(x, W, mslab, M, N, K, ks, per);
        } else {
            if (fp8) mid_splitk_kernel<true, 2, 4><<<mgrid, 512, 4 * 2 * 16384, st>>>(x, W, mslab, M, N, K, ks, per);
            else mid_splitk_kernel<false, 2, 4><<<mgrid, 512, 4 * 2 * 16384, st>>>(x, W, mslab, M, N, K, ks, per);
        }
        out->slab = mslab;
        out->KS = ks;
        out->M = M;
        out->N = N;
        return launch_status("linear(split-K 128-row)");
    }
    const int tiles = (N + 63) / 64, KS = p.ks, spp = p.per;
    float *slab = ws.p;
    const int mt = (M + 15) / 16;
    const dim3 grid(tiles * KS);
#define LLMIE_SK(MT_)                                                                                          \
    (wbits == 16 ? skinny_splitk_kernel<MT_, 16><<<grid, 256, 0, st>>>(x, W, slab, M, K, N, KS, spp, nullptr)     \
     : wbits == 8 ? skinny_splitk_kernel<MT_, 8><<<grid, 256, 0, st>>>(x, W, slab, M, K, N, KS, spp, nullptr)     \
                  : skinny_splitk_kernel<MT_, 8, true><<<grid, 256, 0, st>>>(x, W, slab, M, K, N, KS, spp, nullptr))
    if (wbits == 4) {
        switch (mt) {
            case 1: skinny_splitk_kernel<1, 4><<<grid, 256, 0, st>>>(x, W, slab, M, K, N, KS, spp, gscale); break;
            case 2: skinny_splitk_kernel<2, 4><<<grid, 256, 0, st>>>(x, W, slab, M, K, N, KS, spp, gscale); break;
            case 3: skinny_splitk_kernel<3, 4><<<grid, 256, 0, st>>>(x, W, slab, M, K, N, KS, spp, gscale); break;
            default: skinny_splitk_kernel<4, 4><<<grid, 256, 0, st>>>(x, W, slab, M, K, N, KS, spp, gscale); break;
        }
    } else
    switch (mt) {
        case 1: LLMIE_SK(1); break;
        case 2: LLMIE_SK(2); break;
        case 3: LLMIE_SK(3); break;
        case 4: LLMIE_SK(4); break;
        case 5: LLMIE_SK(5); break;
        case 6: LLMIE_SK(6); break;
        case 7: LLMIE_SK(7); break;
        default: LLMIE_SK(8); break;
    }
#undef LLMIE_SK
    out->slab = slab;
    out->KS = KS;
    out->M = M;
    out->N = N;
    return launch_status("linear(split-K)");
}

// split-K skinny MFMA path: 8 < M (any M, 128 rows of x per pass); wbits 16 or 8
int linear_splitk(int wbits, const half_t *x, const void *W, const half_t *scale, half_t *y, int M, int K, int N, int epi,
                  const half_t *bias, const half_t *residual, SlabWs ws, hipStream_t st) {
    // wbits 8: `scale` = per-row fp16 scales applied by the finalize; wbits 4: `scale` = group-128 scales applied in the kernel
    const int mpass = wbits == 4 ? 64 : kSplitKPassRows;
    const int out_n = epi == EPI_SWIGLU ? N / 2 : N;
    for (int m0 = 0; m0 < M; m0 += mpass) {
        const int mc = M - m0 < mpass ? M - m0 : mpass;
        SplitKSlabs sk;
        const int rc = linear_splitk_partial(wbits, x + static_cast<size_t>(m0) * K, W, mc, K, N, st, &sk, ws, wbits == 4 ? scale : nullptr);
        if (rc) return rc;
        launch_finalize(sk.slab, y + static_cast<size_t>(m0) * out_n, mc, N, sk.KS, SlabScale{wbits == 4 ? nullptr : scale, nullptr, nullptr}, bias,
                        residual ? residual + static_cast<size_t>(m0) * N : nullptr, epi, st);
    }
    return launch_status("linear(split-K)");
}

// elementwise consumer of the slabs: y = scale(sum_ks slab) (+bias) (+residual) | SwiGLU over (n, N/2 + n)
int splitk_finalize(const SplitKSlabs &sk, const SlabScale &sc, half_t *y, int epi, const half_t *bias, const half_t *residual,
                    hipStream_t st) {
    launch_finalize(sk.slab, y, sk.M, sk.N, sk.KS, sc, bias, residual, epi, st);
    return launch_status("linear(split-K finalize)");
}

// Row epilogue of a split-K projection fused with the residual stream and the next RMSNorm (one launch instead of
// finalize + launchFusedAddBiasResidualAndRMSNorm, add_residual_and_rmsnorm.cu:43-121 semantics):
//   t = sum_ks slab[ks][m][:] (* wscale[:]) + resid[m][:];   resid[m][:] = t;   t += bias;
//   y[m][:] = gamma ? t * rsqrt(mean(t^2) + eps) * gamma : t
// One 1024-thread workgroup per row (a thread owns 4 columns per 4096; every slab load of the row is in flight at
// once: the slabs were just written by other XCDs, so each load is a full memory round trip).  N <= 8192, N % 4 == 0.
// With xq != null the normalised row is ALSO quantised per token to e4m3 (scale amax/448 -> xscale[m]; the arithmetic of
// quantize_rows_fp8_kernel on the fp16-rounded values), the input format of the next fp8 projection.
static __global__ __launch_bounds__(1024) void splitk_rownorm_kernel(const float *__restrict__ slab, int KS, int M, int N,
                                                                     const SlabScale wscale,
                                                                     const half_t *__restrict__ bias, half_t *resid,
                                                                     const half_t *__restrict__ gamma, float eps, half_t *y,
                                                                     uint8_t *xq, float *xscale) {
    __shared__ float red[16];
    const int m = blockIdx.x, tid = threadIdx.x;
    const size_t slab_sz = static_cast<size_t>(M) * N;
    constexpr int NC = 2;
    floatx4 v[NC];
    half4_t r4[NC];
    int col[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        col[i] = (i * 1024 + tid) * 4;
        v[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        if (col[i] < N) r4[i] = *reinterpret_cast<const half4_t *>(resid + static_cast<size_t>(m) * N + col[i]);
    }
    for (int k0 = 0; k0 < KS; k0 += 8) {
        floatx4 t[8][NC];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int i = 0; i < NC; ++i)
                if (col[i] < N)
                    t[kk][i] = *reinterpret_cast<const floatx4 *>(slab + static_cast<size_t>(min(k0 + kk, KS - 1)) * slab_sz +
                                                                  static_cast<size_t>(m) * N + col[i]);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
            if (k0 + kk < KS) {
#pragma unroll
                for (int i = 0; i < NC; ++i) v[i] += t[kk][i];
            }
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        if (col[i] < N) {
            const int c = col[i];
            half4_t o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = wscale.apply(v[i][e], m, c + e);
                // same roundings as the unfused sequence: projection output -> fp16, residual sum -> fp16
                t = to_f32(from_f32<half_t>(t)) + to_f32(r4[i][e]);
                o4[e] = from_f32<half_t>(t);
                t = to_f32(o4[e]);
                if (bias) t += to_f32(bias[c + e]);
                v[i][e] = t;
                ss = fmaf(t, t, ss);
            }
            *reinterpret_cast<half4_t *>(resid + static_cast<size_t>(m) * N + c) = o4;
        }
    }
    float inv = 1.f;
    if (gamma) inv = rsqrtf(block_sum<16>(ss, red) / static_cast<float>(N) + eps);
    half4_t o4[NC];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        if (col[i] < N) {
            const int c = col[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o4[i][e] = from_f32<half_t>(gamma ? v[i][e] * inv * to_f32(gamma[c + e]) : v[i][e]);
                amax = fmaxf(amax, fabsf(to_f32(o4[i][e])));
            }
            if (y) *reinterpret_cast<half4_t *>(y + static_cast<size_t>(m) * N + c) = o4[i];
        }
    }
    if (xq) {
        amax = block_max<16>(amax, red);
        const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
        if (tid == 0) xscale[m] = sc;
#pragma unroll
        for (int i = 0; i < NC; ++i)
            if (col[i] < N)
                *reinterpret_cast<unsigned int *>(xq + static_cast<size_t>(m) * N + col[i]) =
                    pack4_e4m3(to_f32(o4[i][0]) / sc, to_f32(o4[i][1]) / sc, to_f32(o4[i][2]) / sc, to_f32(o4[i][3]) / sc);
    }
}

bool splitk_rownorm_eligible(int N) { return N % 4 == 0 && N <= 8192; }

int splitk_rownorm(const SplitKSlabs &sk, const SlabScale &wscale, const half_t *bias, half_t *resid, const half_t *gamma,
                   float eps, half_t *y, uint8_t *xq, float *xscale, hipStream_t st) {
    if (!splitk_rownorm_eligible(sk.N)) {
        set_error("splitk_rownorm: N=%d not supported", sk.N);
        return LLMIE_ERR_UNSUPPORTED;
    }
    splitk_rownorm_kernel<<<sk.M, 1024, 0, st>>>(sk.slab, sk.KS, sk.M, sk.N, wscale, bias, resid, gamma, eps, y, xq, xscale);
    return launch_status("splitk_rownorm");
}

// does the GEMV family take (M, K)?  (K-split register budget, else the LDS fallback's 64 KB)
bool gemv_f16_eligible(int M, int K, const void *x, const void *W) {
    if (K % 8 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W)) % 16) || M < 1 || M > 8) return false;
    return ksplit_eligible(M, K, 16) || static_cast<size_t>(M) * K * 2 <= 64 * 1024;
}

int linear_f16_nk_norm(const half_t *x, const half_t *W, half_t *y, int M, int K, int N, int epi, const half_t *bias,
                       const half_t *residual, const half_t *gamma, const half_t *pre_bias, float eps, hipStream_t st) {
    if (!gemv_f16_eligible(M, K, x, W) || !gamma || reinterpret_cast<uintptr_t>(gamma) % 16 ||
        reinterpret_cast<uintptr_t>(pre_bias) % 16) {
        set_error("linear(norm-fused): shape M=%d K=%d not on the GEMV path", M, K);
        return LLMIE_ERR_UNSUPPORTED;
    }
    const GemvArgs a{x, W, y, K, N, bias, residual, gamma, pre_bias, eps, epi, 1, nullptr, 0};
    dispatch_gemv(M, a, st);
    return launch_status("linear(norm-fused)");
}

// 256 x 256 (or 256 x 128) LDS-DMA GEMM (gemm256.cuh): fp16 operands, or e4m3 operands with per-token / per-row scales.
// Tile choice by grid fill: 256-wide column tiles when they give >= min_tiles workgroups (one per CU), else 128-wide.
static int g256_group_m() {
    constexpr int v = 4;
    return v;
}
static int gemm256_wn(int M, int N) {
    constexpr int min_tiles = 192, min_tiles_narrow = 128;
    const int tm = (M + 255) / 256;
    if (tm * ((N + 255) / 256) >= min_tiles) return 4;
    // 256 x 128 tiles already from HALF the chip (round 3): at 1024 tokens the O / down projections (N = 4096: 128 tiles) used to
    // fall to the 128 x 128 kernel -- 80 / 181 us against 58 / 140 us on the eight-phase kernel; 1 x 1024 prefill 57.7k -> 65.5k tok/s
    if (tm * ((N + 127) / 128) >= min_tiles_narrow) return 2;
    return 0;
}
bool gemm256_fills(int M, int N) { return gemm256_wn(M, N) != 0; }

// eight-phase form of the 256 x 256 tile (gemm8p.cuh): per-lane DMA offsets are 32-bit
static bool g8p_fits(int N, int K, bool fp8) {
    return (static_cast<size_t>(N) + 512) * K * (fp8 ? 1 : 2) < (size_t{1} << 32);
}

template <bool FP8, bool EPI, int WN, int WQ = 0>
static void gemm256_launch_t(const void *x, const void *W, half_t *y, int M, int N, int K, const half_t *bias, const half_t *residual,
                             const float *xscale, const float *wscale, hipStream_t st, int ldc = 0) {
    constexpr int lds_bytes = 2 * (2 + WN / 2) * 128 * 128;
    if constexpr (WQ != 0) {   // int8 weights: the eight-phase kernels only (callers check g8p_w8_eligible)
        const int tmq = (M + 255) / 256, tnq = (N + 64 * WN - 1) / (64 * WN);
        if constexpr (WN == 4) {
            static const bool a8 = [] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm8p_kernel<false, EPI, false, WQ>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
                return true;
            }();
            (void)a8;
            gemm8p_kernel<false, EPI, false, WQ><<<tmq * tnq, 512, lds_bytes, st>>>(x, W, y, M, N, K, bias, residual, tnq, xscale, wscale, ldc, g256_group_m());
        } else {
            constexpr int ring_bytes = 9 * 128 * 128;
            static const bool a8 = [] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm8p_n128_kernel<false, EPI, false, WQ>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, ring_bytes);
                return true;
            }();
            (void)a8;
            gemm8p_n128_kernel<false, EPI, false, WQ><<<tmq * tnq, 512, ring_bytes, st>>>(x, W, y, M, N, K, bias, residual, tnq, xscale, wscale, ldc, g256_group_m());
        }
        return;
    }
    static const bool attr_set = [] {   // once per process, thread-safe (function-local static initialisation)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm256_kernel<FP8, EPI, WN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        return true;
    }();
    (void)attr_set;
    const int tm = (M + 255) / 256, tn = (N + 64 * WN - 1) / (64 * WN);
    if constexpr (WN == 4) {
        if (g8p_fits(N, K, FP8)) {
            static const bool attr8 = [] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm8p_kernel<FP8, EPI, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
                return true;
            }();
            (void)attr8;
            gemm8p_kernel<FP8, EPI, false><<<tm * tn, 512, lds_bytes, st>>>(x, W, y, M, N, K, bias, residual, tn, xscale, wscale, ldc, g256_group_m());
            return;
        }
    }
    if constexpr (WN == 2) {
        if (g8p_fits(N, K, FP8)) {
            constexpr int ring_bytes = 9 * 128 * 128;
            static const bool attr8 = [] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm8p_n128_kernel<FP8, EPI>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, ring_bytes);
                return true;
            }();
            (void)attr8;
            gemm8p_n128_kernel<FP8, EPI><<<tm * tn, 512, ring_bytes, st>>>(x, W, y, M, N, K, bias, residual, tn, xscale, wscale, ldc, g256_group_m());
            return;
        }
    }
    gemm256_kernel<FP8, EPI, WN><<<tm * tn, 512, lds_bytes, st>>>(x, W, y, M, N, K, bias, residual, tn, xscale, wscale, ldc, g256_group_m());
}

// SwiGLU form: W = fused gate_up [2I, K], y = silu(x.Wg^T) * (x.Wu^T) [M, I]
bool gemm256_swiglu_fills(int M, int two_inter) {
    constexpr int min_tiles = 192;
    constexpr bool off = false;
    return !off && two_inter % 8 == 0 && ((M + 255) / 256) * ((two_inter / 2 + 127) / 128) >= min_tiles;
}
void gemm256_swiglu_launch(bool fp8, const void *x, const void *W, half_t *y, int M, int two_inter, int K, const float *xscale,
                           const float *wscale, hipStream_t st, int wq) {
    constexpr int lds_bytes = 2 * 4 * 128 * 128;
    static const bool attr_set = [] {   // once per process, thread-safe (function-local static initialisation)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm256_kernel<false, false, 4, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm256_kernel<true, false, 4, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        return true;
    }();
    (void)attr_set;
    const int tm = (M + 255) / 256, tn = (two_inter / 2 + 127) / 128;
    if (g8p_fits(two_inter, K, fp8)) {
        static const bool attr8 = [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm8p_kernel<false, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm8p_kernel<true, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            return true;
        }();
        (void)attr8;
        // Rounds of the 256 CUs, as gemm256_launch plans them: a partly filled last round of 128-column tiles (4096 tokens: 1376 tiles =
        // 5.4 rounds) runs instead as 64-column tiles (gemm8p_n128_kernel's SwiGLU form, ~0.6 of a wide tile each) over the
        // left-over columns, in a second launch -- when that is cheaper (not at 2048 tokens: 2.7 rounds become 2 + 2 x 0.6)
        constexpr int cus = 256, ring_bytes = 9 * 128 * 128;
        auto rounds = [&](int t) { return (t + cus - 1) / cus; };
        const int inter = two_inter / 2, tiles = tm * tn, full = tiles / cus;
        int a_tn = tn, b_tn = 0;
        if (full >= 1 && tiles % cus != 0 && inter % 4 == 0) {
            const int a = full * cus / tm;   // 128-column tiles that make whole rounds
            if (a > 0 && a < tn) {
                const int tnb = (inter - a * 128 + 63) / 64;
                if (rounds(a * tm) + rounds(tm * tnb) * 0.6f < static_cast<float>(rounds(tiles))) {
                    a_tn = a;
                    b_tn = tnb;
                }
            }
        }
        if (wq) {   // int8 weights (wscale = their fp16 row scales)
            static const bool attrq = [] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm8p_kernel<false, false, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm8p_n128_kernel<false, false, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, ring_bytes);
                return true;
            }();
            (void)attrq;
            gemm8p_kernel<false, false, true, 8><<<tm * a_tn, 512, lds_bytes, st>>>(x, W, y, M, two_inter, K, nullptr, nullptr, a_tn, nullptr, wscale, 0, g256_group_m(), 0);
            if (b_tn)
                gemm8p_n128_kernel<false, false, true, 8><<<tm * b_tn, 512, ring_bytes, st>>>(x, W, y, M, two_inter, K, nullptr, nullptr, b_tn, nullptr, wscale, 0, g256_group_m(), a_tn * 128);
            return;
        }
        if (fp8)
            gemm8p_kernel<true, false, true><<<tm * a_tn, 512, lds_bytes, st>>>(x, W, y, M, two_inter, K, nullptr, nullptr, a_tn, xscale, wscale, 0, g256_group_m(), 0);
        else
            gemm8p_kernel<false, false, true><<<tm * a_tn, 512, lds_bytes, st>>>(x, W, y, M, two_inter, K, nullptr, nullptr, a_tn, nullptr, nullptr, 0, g256_group_m(), 0);
        if (b_tn) {
            static const bool attrn = [] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm8p_n128_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ring_bytes);
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm8p_n128_kernel<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ring_bytes);
                return true;
            }();
            (void)attrn;
            if (fp8)
                gemm8p_n128_kernel<true, false, true><<<tm * b_tn, 512, ring_bytes, st>>>(x, W, y, M, two_inter, K, nullptr, nullptr, b_tn, xscale, wscale, 0, g256_group_m(), a_tn * 128);
            else
                gemm8p_n128_kernel<false, false, true><<<tm * b_tn, 512, ring_bytes, st>>>(x, W, y, M, two_inter, K, nullptr, nullptr, b_tn, nullptr, nullptr, 0, g256_group_m(), a_tn * 128);
        }
        return;
    }
    if (fp8)
        gemm256_kernel<true, false, 4, true><<<tm * tn, 512, lds_bytes, st>>>(x, W, y, M, two_inter, K, nullptr, nullptr, tn, xscale, wscale, 0, g256_group_m());
    else
        gemm256_kernel<false, false, 4, true><<<tm * tn, 512, lds_bytes, st>>>(x, W, y, M, two_inter, K, nullptr, nullptr, tn, nullptr, nullptr, 0, g256_group_m());
}

// One launch over the column range [nb, nb + n) of the [M, N] output with 64*wn-column tiles.
static void gemm256_range(bool fp8, int wn, const void *x, const void *W, half_t *y, int M, int N, int K, const half_t *bias,
                          const half_t *residual, const float *xscale, const float *wscale, hipStream_t st, int nb, int n, int wq = 0) {
    const bool epi = bias || residual;
    const void *Wr = static_cast<const unsigned char *>(W) + static_cast<size_t>(nb) * K * ((fp8 || wq) ? 1 : 2);
    half_t *yr = y + nb;
    const half_t *br = bias ? bias + nb : nullptr, *rr = residual ? residual + nb : nullptr;
    // (int8: the scales are fp16, carried through the float pointer of the shared signature)
    const float *wsr = !wscale ? nullptr : (wq ? reinterpret_cast<const float *>(reinterpret_cast<const half_t *>(wscale) + nb) : wscale + nb);
    if (wq) {
        if (wn == 4) {
            if (epi) gemm256_launch_t<false, true, 4, 8>(x, Wr, yr, M, n, K, br, rr, nullptr, wsr, st, N);
            else gemm256_launch_t<false, false, 4, 8>(x, Wr, yr, M, n, K, br, rr, nullptr, wsr, st, N);
        } else {
            if (epi) gemm256_launch_t<false, true, 2, 8>(x, Wr, yr, M, n, K, br, rr, nullptr, wsr, st, N);
            else gemm256_launch_t<false, false, 2, 8>(x, Wr, yr, M, n, K, br, rr, nullptr, wsr, st, N);
        }
        return;
    }
#define LLMIE_G256(F8_, EPI_)                                                                                          \
    (wn == 4 ? gemm256_launch_t<F8_, EPI_, 4>(x, Wr, yr, M, n, K, br, rr, xscale, wsr, st, N)                          \
             : gemm256_launch_t<F8_, EPI_, 2>(x, Wr, yr, M, n, K, br, rr, xscale, wsr, st, N))
    if (fp8) {
        if (epi) LLMIE_G256(true, true);
        else LLMIE_G256(true, false);
    } else {
        if (epi) LLMIE_G256(false, true);
        else LLMIE_G256(false, false);
    }
#undef LLMIE_G256
}

// Tile plan by rounds of the 256 CUs (one 512-thread workgroup per CU): a 256 x 128 tile costs ~0.6 of a 256 x 256 tile
// (measured: 0.96 vs 1.12 PFLOP/s on full rounds).  A half-empty last round of 256-wide tiles (qkv at 2048 tokens: 384
// tiles = 1.5 rounds) is avoided by running the full rounds 256-wide and the remaining columns 128-wide in a second launch.
void gemm256_launch(bool fp8, const void *x, const void *W, half_t *y, int M, int N, int K, const half_t *bias,
                    const half_t *residual, const float *xscale, const float *wscale, hipStream_t st, int wq) {
    constexpr int cus = 256;
    constexpr bool no_split = false;
    const int tm = (M + 255) / 256, tn4 = (N + 255) / 256, tn2 = (N + 127) / 128;
    const int tiles4 = tm * tn4, tiles2 = tm * tn2;
    auto rounds = [&](int t) { return (t + cus - 1) / cus; };
    const float narrow = 0.6f;
    float best = static_cast<float>(rounds(tiles4));
    int plan = 4;
    if (gemm256_wn(M, N) != 4 && (N % 4 != 0 || tiles4 < cus / 2)) {  // the 256-wide grid fills less than half the chip
        plan = 2;
    } else if (!no_split && N % 4 == 0) {   // (from half the chip on, the rounds decide: 768 x 12288 is ONE round of 144 wide tiles or
                                            //  TWO of 288 narrow ones -- 97 against 119 us)
        if (rounds(tiles2) * narrow < best) {
            best = rounds(tiles2) * narrow;
            plan = 2;
        }
        const int a_tn = (tiles4 / cus) * cus / tm;  // 256-wide column tiles that make whole rounds
        if (a_tn > 0 && a_tn < tn4) {
            const int nB = N - a_tn * 256, tilesB = tm * ((nB + 127) / 128);
            const float c = static_cast<float>(rounds(tm * a_tn)) + rounds(tilesB) * narrow;
            if (c < best) {
                best = c;
                plan = 42;
            }
        }
    }
    if (plan == 42) {
        const int a_tn = (tiles4 / cus) * cus / tm;
        gemm256_range(fp8, 4, x, W, y, M, N, K, bias, residual, xscale, wscale, st, 0, a_tn * 256, wq);
        gemm256_range(fp8, 2, x, W, y, M, N, K, bias, residual, xscale, wscale, st, a_tn * 256, N - a_tn * 256, wq);
    } else {
        gemm256_range(fp8, plan, x, W, y, M, N, K, bias, residual, xscale, wscale, st, 0, N, wq);
    }
}

// ---- QKV projection of a prefill with RoPE + KV-cache append as its epilogue (gemm8p.cuh ROPE forms; context_attention.cpp:158-205
// as one launch sequence).  kind: 0 = fp16 operands, 1 = e4m3 operands (xscale / wscale), 8 = int8 weights (wscale = fp16 row scales).
// N = (head_num + 2 kv_head_num) * 128; the tile plan is gemm256_launch's, with every column range a whole number of tiles (a tile
// never straddles the end of the matrix: the ROPE forms fetch their weight rows unclamped, in permuted order).
bool gemm256_qkv_rope_eligible(int kind, int M, int N, int K, const void *x, const void *W, const void *wscale, const void *qkv) {
    const bool fp8 = kind == 1;
    return N % 128 == 0 && K % (fp8 ? 128 : 64) == 0 && gemm256_fills(M, N) &&
           (static_cast<size_t>(N) + 512) * K * (fp8 ? 1 : 2) < (size_t{1} << 32) &&
           (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W)) % 16 == 0 &&
           (reinterpret_cast<uintptr_t>(wscale) % (fp8 ? 16 : 8)) == 0 && reinterpret_cast<uintptr_t>(qkv) % 8 == 0;
}
template <bool FP8, int WQ, int WN>
static void qkv_rope_range(const void *x, const void *W, half_t *qkv, int M, int N, int K, const float *xscale, const float *wscale,
                           const half_t *bias, const QkvRopeArgs *rap, int layer, hipStream_t st, int nb, int n) {
    // W is offset to the range like every range launch; C, the scales and the bias stay whole: the epilogue indexes them by the
    // absolute output column col0 + tile column.  rap: DEVICE pointer (prefill_token_table); bias: this layer's QKV bias or null
    const void *Wr = static_cast<const unsigned char *>(W) + static_cast<size_t>(nb) * K * ((FP8 || WQ) ? 1 : 2);
    const int tm = (M + 255) / 256, tn = n / (64 * WN);
    if constexpr (WN == 4) {
        constexpr int lds_bytes = 2 * 4 * 128 * 128;
        static const bool attr = [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm8p_kernel<FP8, false, false, WQ, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            return true;
        }();
        (void)attr;
        gemm8p_kernel<FP8, false, false, WQ, true><<<tm * tn, 512, lds_bytes, st>>>(x, Wr, qkv, M, n, K, bias, nullptr, tn, xscale, wscale, N,
                                                                                   g256_group_m(), nb, rap, layer);
    } else {
        constexpr int ring_bytes = 9 * 128 * 128;
        static const bool attr = [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm8p_n128_kernel<FP8, false, false, WQ, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, ring_bytes);
            return true;
        }();
        (void)attr;
        gemm8p_n128_kernel<FP8, false, false, WQ, true><<<tm * tn, 512, ring_bytes, st>>>(x, Wr, qkv, M, n, K, bias, nullptr, tn, xscale, wscale, N,
                                                                                         g256_group_m(), nb, rap, layer);
    }
}
void gemm256_qkv_rope_launch(int kind, const void *x, const void *W, half_t *qkv, int M, int N, int K, const float *xscale,
                             const float *wscale, const half_t *bias, const QkvRopeArgs *rap, int layer, hipStream_t st) {
    constexpr int cus = 256;
    const int tm = (M + 255) / 256, tn4 = N / 256, tn2 = N / 128;
    auto rounds = [&](int t) { return (t + cus - 1) / cus; };
    const float narrow = 0.6f;
    // candidates: all 128-wide; all 256-wide (N a multiple of 256); whole rounds 256-wide + the rest 128-wide
    int plan = 2, a_tn = 0;
    float best = rounds(tm * tn2) * narrow;
    if (gemm256_wn(M, N) == 4 || tm * tn4 >= cus / 2) {   // (from half the chip on, the rounds decide: gemm256_launch's rule)
        if (N % 256 == 0 && static_cast<float>(rounds(tm * tn4)) < best) {
            best = static_cast<float>(rounds(tm * tn4));
            plan = 4;
        }
        const int a = (tm * tn4 / cus) * cus / tm;   // 256-wide column tiles that make whole rounds
        if (a > 0 && a * 256 < N) {
            const float c = static_cast<float>(rounds(tm * a)) + rounds(tm * ((N - a * 256) / 128)) * narrow;
            if (c < best) {
                best = c;
                plan = 42;
                a_tn = a;
            }
        }
    }
    auto range = [&](int wn, int nb, int n) {
#define LLMIE_QR(F8_, WQ_) (wn == 4 ? qkv_rope_range<F8_, WQ_, 4>(x, W, qkv, M, N, K, xscale, wscale, bias, rap, layer, st, nb, n) \
                                    : qkv_rope_range<F8_, WQ_, 2>(x, W, qkv, M, N, K, xscale, wscale, bias, rap, layer, st, nb, n))
        if (kind == 1) LLMIE_QR(true, 0);
        else if (kind == 8) LLMIE_QR(false, 8);
        else LLMIE_QR(false, 0);
#undef LLMIE_QR
    };
    if (plan == 42) {
        range(4, 0, a_tn * 256);
        range(2, a_tn * 256, N - a_tn * 256);
    } else {
        range(plan, 0, N);
    }
}

// int8 [N, K] weights through the eight-phase kernels (gemm8p.cuh, WQ = 8): prefill-sized M whose 256-row grid fills the chip
bool g8p_w8_eligible(int M, int K, int N, const void *x, const void *wq, const void *scale, const void *y) {
    return K % 64 == 0 && N % 4 == 0 && gemm256_fills(M, N) && (static_cast<size_t>(N) + 512) * K * 2 < (size_t{1} << 32) &&
           (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wq)) % 16 == 0 &&
           (reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(y)) % 8 == 0;
}
bool g8p_w8_swiglu_eligible(int M, int K, int two_inter, const void *x, const void *wq, const void *scale, const void *y) {
    return K % 64 == 0 && two_inter % 8 == 0 && gemm256_swiglu_fills(M, two_inter) &&
           (static_cast<size_t>(two_inter) + 512) * K * 2 < (size_t{1} << 32) &&
           (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wq)) % 16 == 0 &&
           (reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(y)) % 8 == 0;
}

int linear_f16_nk(const half_t *x, const half_t *W, half_t *y, int M, int K, int N, int epi,
                  const half_t *bias, const half_t *residual, SlabWs ws, hipStream_t st) {
    const bool aligned = (K % 8 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W)) % 16 == 0);
    bool done = false;
    if (gemv_f16_eligible(M, K, x, W)) {
        const GemvArgs a{x, W, y, K, N, bias, residual, nullptr, nullptr, 0.f, epi, 0, nullptr, 0};
        done = dispatch_gemv(M, a, st);
    }
    // decode / short-prefill batches: split-K over the caller's slabs (without a workspace: the non-split kernels below).
    // Round 3, 192 < M where the eight-phase kernels' 256-row grid does not fill the chip (N = 4096 at 256-1023 tokens): such
    // shapes used to fall to the 128 x 128 kernel -- 82 us for the O and 190 us for the down projection of a 7B layer WHATEVER the
    // row count.  Two cheaper forms, picked by a two-term time model fitted to the measurements (tools/dev/s2_run18.sh sweep):
    //   split-K passes of 128 rows:  passes x (weight bytes / 4.2 TB/s + 10 us)          O: 19.5 us per pass, down: 30 us
    //   256 x 128 eight-phase tiles: rounds of 256 tiles x K / 64 k-tiles x 0.9 us        O: 58 us, down: 140 us per round
    bool mid_rows = false, mid_tiles = false;
    if (M > 192 && epi != EPI_SWIGLU && aligned && K % 64 == 0 && !gemm256_fills(M, N)) {
        const int passes = (M + kSplitKPassRows - 1) / kSplitKPassRows;
        const float t_split = passes * (static_cast<float>(N) * K * 2.f / 4.2e6f + 10.f);
        const int tiles2 = ((M + 255) / 256) * ((N + 127) / 128);
        const float t_tiles = static_cast<float>((tiles2 + 255) / 256) * (K / 64) * 0.9f;
        const bool split_ok = ws.p && K % 128 == 0 && K >= 512 && ws.floats >= linear_splitk_ws_floats(16, kSplitKPassRows, K, N);
        const bool tiles_ok = g8p_fits(N, K, false) && reinterpret_cast<uintptr_t>(y) % 8 == 0 &&
                              (reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual)) % 8 == 0;
        mid_rows = split_ok && (!tiles_ok || t_split < t_tiles);
        mid_tiles = tiles_ok && !mid_rows;
    }
    if (!done && ws.p && aligned && (M <= 192 || mid_rows) && K % 128 == 0 && K >= 512)
        return linear_splitk(16, x, W, nullptr, y, M, K, N, epi, bias, residual, ws, st);
    if (!done && aligned && M <= 64 && K % 32 == 0 && (epi != EPI_SWIGLU || (N / 2) % 16 == 0)) {
        done = (epi == EPI_SWIGLU) ? dispatch_skinny<EPI_SWIGLU>(M, x, W, y, K, N, bias, residual, st)
                                   : dispatch_skinny<EPI_NONE>(M, x, W, y, K, N, bias, residual, st);
    }
    if (!done && epi == EPI_SWIGLU && aligned && K % 64 == 0 && gemm256_swiglu_fills(M, N) && reinterpret_cast<uintptr_t>(y) % 8 == 0) {
        gemm256_swiglu_launch(false, x, W, y, M, N, K, nullptr, nullptr, st);
        return launch_status("linear(gemm256 SwiGLU)");
    }
    if (!done && epi == EPI_SWIGLU) {
        set_error("linear: fused SwiGLU epilogue without a split-K workspace needs M<=64, K%%32==0, (N/2)%%16==0 (M=%d K=%d N=%d)", M, K, N);
        return LLMIE_ERR_UNSUPPORTED;
    }
    if (!done && aligned && K % 64 == 0 && reinterpret_cast<uintptr_t>(y) % 8 == 0) {
        // 256 x 256 LDS-DMA kernel when its grid fills the chip (one 512-thread workgroup per CU); else 128 x 128 tiles
        if ((gemm256_fills(M, N) || mid_tiles) && (reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual)) % 8 == 0) {
            gemm256_launch(false, x, W, y, M, N, K, bias, residual, nullptr, nullptr, st);
            return launch_status("linear(gemm256)");
        }
        dim3 grid((N + 127) / 128, (M + 127) / 128, 1);
        if (bias || residual)
            tiled_mfma_f16_kernel<true><<<grid, 256, 0, st>>>(x, W, y, M, N, K, 0, 0, 0, bias, residual);
        else
            tiled_mfma_f16_kernel<false><<<grid, 256, 0, st>>>(x, W, y, M, N, K, 0, 0, 0, nullptr, nullptr);
        done = true;
    }
    if (!done) launch_generic<half_t>(x, W, y, 1, M, N, K, true, bias, residual, st);
    return launch_status("linear");
}

}  // namespace llmie

using namespace llmie;

// fp32 slab scratch of the split-K forms for this shape (the counterpart of the workspace the reference's cublasWrapper owns)
extern "C" size_t llmie_linear_workspace_bytes(llmie_weight_format fmt, int M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    int wbits;
    switch (fmt) {
        case LLMIE_W_F16: wbits = 16; break;
        case LLMIE_W_INT8: wbits = 8; break;
        case LLMIE_W_INT4: wbits = 4; break;
        case LLMIE_W_FP8: wbits = WF_FP8; break;
        default: return 0;
    }
    // prefill-sized fp16: tiled kernels, no slabs -- except where the 256-row grid does not fill the chip: split-K passes of 128
    // rows may be the cheaper form there (linear_f16_nk's time model)
    if (fmt == LLMIE_W_F16 && M > 192 && gemm256_fills(M, N)) return 0;
    // int8 / int4 at prefill-sized M: room for the fp16 image of W in front of the slabs (llmie_linear_w8a16 / _w4a16)
    const size_t dq = (fmt == LLMIE_W_INT8 || fmt == LLMIE_W_INT4) ? ((linear_wq_dequant_bytes(wbits, M, K, N) + 255) & ~static_cast<size_t>(255)) : 0;
    return dq + linear_splitk_ws_floats(wbits, M, K, N) * sizeof(float);
}

static bool slab_ws_of(void *workspace, size_t bytes, SlabWs *out) {
    *out = SlabWs{static_cast<float *>(workspace), bytes / sizeof(float)};
    return !workspace || reinterpret_cast<uintptr_t>(workspace) % 16 == 0;
}

extern "C" int llmie_linear(const void *x, const void *w, void *y, int M, int K, int N, int trans_b,
                            const void *bias, const void *residual, llmie_dtype dtype, void *workspace,
                            size_t workspace_bytes, llmie_stream stream) {
    LLMIE_REQUIRE(x && w && y, "linear: NULL pointer");
    LLMIE_REQUIRE(M > 0 && K > 0 && N > 0, "linear: bad shape M=%d K=%d N=%d", M, K, N);
    SlabWs ws;
    LLMIE_REQUIRE(slab_ws_of(workspace, workspace_bytes, &ws), "linear: workspace must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    if (dtype == LLMIE_F16) {
        if (trans_b)
            return linear_f16_nk((const half_t *)x, (const half_t *)w, (half_t *)y, M, K, N, EPI_NONE,
                                 (const half_t *)bias, (const half_t *)residual, ws, st);
        launch_generic<half_t>((const half_t *)x, (const half_t *)w, (half_t *)y, 1, M, N, K, false,
                               (const half_t *)bias, (const half_t *)residual, st);
        return launch_status("linear");
    }
    if (dtype == LLMIE_F32) {
        launch_generic<float>((const float *)x, (const float *)w, (float *)y, 1, M, N, K, trans_b != 0,
                              (const float *)bias, (const float *)residual, st);
        return launch_status("linear");
    }
    LLMIE_UNSUPPORTED("linear: dtype %d", (int)dtype);
}

extern "C" int llmie_linear_swiglu(const void *x, const void *w, void *y, int M, int K, int two_inter,
                                   llmie_dtype dtype, void *workspace, size_t workspace_bytes, llmie_stream stream) {
    LLMIE_REQUIRE(x && w && y, "linear_swiglu: NULL pointer");
    LLMIE_REQUIRE(M > 0 && K > 0 && two_inter > 0 && two_inter % 2 == 0, "linear_swiglu: bad shape");
    if (dtype != LLMIE_F16) LLMIE_UNSUPPORTED("linear_swiglu: fp16 only (dtype %d)", (int)dtype);
    SlabWs ws;
    LLMIE_REQUIRE(slab_ws_of(workspace, workspace_bytes, &ws), "linear_swiglu: workspace must be 16-byte aligned");
    return linear_f16_nk((const half_t *)x, (const half_t *)w, (half_t *)y, M, K, two_inter, EPI_SWIGLU, nullptr,
                         nullptr, ws, as_stream(stream));
}

extern "C" int llmie_batched_gemm(const void *a, const void *b, void *c, int batch, int m, int n, int k,
                                  int trans_b, llmie_dtype dtype, llmie_stream stream) {
    LLMIE_REQUIRE(a && b && c, "batched_gemm: NULL pointer");
    LLMIE_REQUIRE(batch > 0 && m > 0 && n > 0 && k > 0, "batched_gemm: bad shape");
    LLMIE_REQUIRE(batch <= 65535, "batched_gemm: batch > 65535");
    hipStream_t st = as_stream(stream);
    if (dtype == LLMIE_F16 && trans_b && k % 64 == 0 && m >= 32 &&
        ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) % 16 == 0) && reinterpret_cast<uintptr_t>(c) % 8 == 0 &&
        (static_cast<size_t>(m) * k) % 8 == 0 && (static_cast<size_t>(n) * k) % 8 == 0 && (static_cast<size_t>(m) * n) % 4 == 0) {
        dim3 grid((n + 127) / 128, (m + 127) / 128, batch);
        tiled_mfma_f16_kernel<false><<<grid, 256, 0, st>>>((const half_t *)a, (const half_t *)b, (half_t *)c, m, n, k,
                                                          static_cast<size_t>(m) * k, static_cast<size_t>(n) * k,
                                                          static_cast<size_t>(m) * n, nullptr, nullptr);
    } else if (dtype == LLMIE_F16)
        launch_generic<half_t>((const half_t *)a, (const half_t *)b, (half_t *)c, batch, m, n, k, trans_b != 0,
                               nullptr, nullptr, st);
    else if (dtype == LLMIE_F32)
        launch_generic<float>((const float *)a, (const float *)b, (float *)c, batch, m, n, k, trans_b != 0,
                              nullptr, nullptr, st);
    else
        LLMIE_UNSUPPORTED("batched_gemm: dtype %d", (int)dtype);
    return launch_status("batched_gemm");
}

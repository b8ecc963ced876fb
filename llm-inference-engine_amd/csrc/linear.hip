// llmie_linear / llmie_batched_gemm: shape dispatch over the kernels in gemm_kernels.cuh.
#include "gemm_kernels.cuh"
#include "llmie_internal.h"

#include <cstdlib>

namespace llmie {

static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

// LLMIE_DECODE_GEMM: 0 = auto, 1 = force fdot2 GEMV (M<=8), 2 = force skinny MFMA
static int decode_gemm_mode() {
    static int mode = env_int("LLMIE_DECODE_GEMM", 0);
    return mode;
}

template <int M, int EPI>
static void launch_gemv(const half_t *x, const half_t *W, half_t *y, int K, int N, const half_t *bias,
                        const half_t *residual, hipStream_t st) {
    const int npairs = (EPI == EPI_SWIGLU) ? N / 2 : (N + 1) / 2;
    int wgs = (npairs + 3) / 4;
    // >= 2 workgroups per CU when there is enough work, at most 8 per CU (grid-stride beyond)
    if (wgs > 2048) wgs = 2048;
    const size_t lds = static_cast<size_t>(M) * K * sizeof(half_t);
    gemv_f16_kernel<M, EPI, 8><<<wgs, 256, lds, st>>>(x, W, y, K, N, bias, residual);
}

template <int EPI>
static bool dispatch_gemv(int M, const half_t *x, const half_t *W, half_t *y, int K, int N,
                          const half_t *bias, const half_t *residual, hipStream_t st) {
    switch (M) {
        case 1: launch_gemv<1, EPI>(x, W, y, K, N, bias, residual, st); return true;
        case 2: launch_gemv<2, EPI>(x, W, y, K, N, bias, residual, st); return true;
        case 3: launch_gemv<3, EPI>(x, W, y, K, N, bias, residual, st); return true;
        case 4: launch_gemv<4, EPI>(x, W, y, K, N, bias, residual, st); return true;
        case 5: launch_gemv<5, EPI>(x, W, y, K, N, bias, residual, st); return true;
        case 6: launch_gemv<6, EPI>(x, W, y, K, N, bias, residual, st); return true;
        case 7: launch_gemv<7, EPI>(x, W, y, K, N, bias, residual, st); return true;
        case 8: launch_gemv<8, EPI>(x, W, y, K, N, bias, residual, st); return true;
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

// fp16, W[N,K]: the decode / prefill projection path.  epi selects the fused epilogue.
int linear_f16_nk(const half_t *x, const half_t *W, half_t *y, int M, int K, int N, int epi,
                  const half_t *bias, const half_t *residual, hipStream_t st) {
    const bool aligned = (K % 8 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W)) % 16 == 0);
    const int mode = decode_gemm_mode();
    bool done = false;
    if (aligned && M <= 8 && mode != 2 && static_cast<size_t>(M) * K * 2 <= 64 * 1024) {
        done = (epi == EPI_SWIGLU) ? dispatch_gemv<EPI_SWIGLU>(M, x, W, y, K, N, bias, residual, st)
                                   : dispatch_gemv<EPI_NONE>(M, x, W, y, K, N, bias, residual, st);
    }
    if (!done && aligned && M <= 64 && K % 32 == 0 && (epi != EPI_SWIGLU || (N / 2) % 16 == 0)) {
        done = (epi == EPI_SWIGLU) ? dispatch_skinny<EPI_SWIGLU>(M, x, W, y, K, N, bias, residual, st)
                                   : dispatch_skinny<EPI_NONE>(M, x, W, y, K, N, bias, residual, st);
    }
    if (!done) {
        if (epi == EPI_SWIGLU) {
            set_error("linear: fused SwiGLU epilogue needs M<=64, K%%32==0, (N/2)%%16==0 (M=%d K=%d N=%d)", M, K, N);
            return LLMIE_ERR_UNSUPPORTED;
        }
        launch_generic<half_t>(x, W, y, 1, M, N, K, true, bias, residual, st);
    }
    return launch_status("linear");
}

}  // namespace llmie

using namespace llmie;

extern "C" int llmie_linear(const void *x, const void *w, void *y, int M, int K, int N, int trans_b,
                            const void *bias, const void *residual, llmie_dtype dtype,
                            llmie_stream stream) {
    LLMIE_REQUIRE(x && w && y, "linear: NULL pointer");
    LLMIE_REQUIRE(M > 0 && K > 0 && N > 0, "linear: bad shape M=%d K=%d N=%d", M, K, N);
    hipStream_t st = as_stream(stream);
    if (dtype == LLMIE_F16) {
        if (trans_b)
            return linear_f16_nk((const half_t *)x, (const half_t *)w, (half_t *)y, M, K, N, EPI_NONE,
                                 (const half_t *)bias, (const half_t *)residual, st);
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
                                   llmie_dtype dtype, llmie_stream stream) {
    LLMIE_REQUIRE(x && w && y, "linear_swiglu: NULL pointer");
    LLMIE_REQUIRE(M > 0 && K > 0 && two_inter > 0 && two_inter % 2 == 0, "linear_swiglu: bad shape");
    if (dtype != LLMIE_F16) LLMIE_UNSUPPORTED("linear_swiglu: fp16 only (dtype %d)", (int)dtype);
    return linear_f16_nk((const half_t *)x, (const half_t *)w, (half_t *)y, M, K, two_inter, EPI_SWIGLU, nullptr,
                         nullptr, as_stream(stream));
}

extern "C" int llmie_batched_gemm(const void *a, const void *b, void *c, int batch, int m, int n, int k,
                                  int trans_b, llmie_dtype dtype, llmie_stream stream) {
    LLMIE_REQUIRE(a && b && c, "batched_gemm: NULL pointer");
    LLMIE_REQUIRE(batch > 0 && m > 0 && n > 0 && k > 0, "batched_gemm: bad shape");
    LLMIE_REQUIRE(batch <= 65535, "batched_gemm: batch > 65535");
    hipStream_t st = as_stream(stream);
    if (dtype == LLMIE_F16)
        launch_generic<half_t>((const half_t *)a, (const half_t *)b, (half_t *)c, batch, m, n, k, trans_b != 0,
                               nullptr, nullptr, st);
    else if (dtype == LLMIE_F32)
        launch_generic<float>((const float *)a, (const float *)b, (float *)c, batch, m, n, k, trans_b != 0,
                              nullptr, nullptr, st);
    else
        LLMIE_UNSUPPORTED("batched_gemm: dtype %d", (int)dtype);
    return launch_status("batched_gemm");
}

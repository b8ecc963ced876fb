// Micro-benchmark for decode GEMV variants (development tool, not part of the library).
// Cycles through NSETS distinct weight matrices per shape so nothing is served from L2/MALL.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemv_bench.hip -o tools/gemv_bench
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../llm-inference-engine_amd/csrc/gemm_kernels.cuh"

namespace llmie { void set_error(const char *, ...) {} }
using namespace llmie;

// ---- V0: first version (stage x, barrier, 16 loads single shot per pair) ----
template <int M, int EPI, int U>
__global__ __launch_bounds__(256) void gemv_v0(const half_t *__restrict__ x, const half_t *__restrict__ W, half_t *__restrict__ y, int K, int N) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half8_t *xs = reinterpret_cast<half8_t *>(smem_raw);
    const int nch = K >> 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const half8_t *xg = reinterpret_cast<const half8_t *>(x);
        for (int i = tid; i < M * nch; i += 256) xs[i] = xg[i];
    }
    __syncthreads();
    const int half_n = N >> 1;
    const int npairs = (EPI == EPI_SWIGLU) ? half_n : ((N + 1) >> 1);
    for (int pair = blockIdx.x * 4 + wave; pair < npairs; pair += gridDim.x * 4) {
        int r0 = (EPI == EPI_SWIGLU) ? pair : 2 * pair, r1 = (EPI == EPI_SWIGLU) ? pair + half_n : min(2 * pair + 1, N - 1);
        const half8_t *w0 = reinterpret_cast<const half8_t *>(W + static_cast<size_t>(r0) * K);
        const half8_t *w1 = reinterpret_cast<const half8_t *>(W + static_cast<size_t>(r1) * K);
        float acc0[M], acc1[M];
#pragma unroll
        for (int m = 0; m < M; ++m) acc0[m] = acc1[m] = 0.f;
        for (int c = lane; c < nch; c += 64 * U) {
            half8_t a0[U], a1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cc = c + 64 * u;
                if (cc < nch) { a0[u] = load_nt(w0 + cc); a1[u] = load_nt(w1 + cc); }
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
        for (int m = 0; m < M; ++m) { acc0[m] = wave_sum(acc0[m]); acc1[m] = wave_sum(acc1[m]); }
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                if (EPI == EPI_SWIGLU) y[static_cast<size_t>(m) * half_n + pair] = from_f32<half_t>((acc0[m] / (1.0f + expf(-acc0[m]))) * acc1[m]);
                else { y[static_cast<size_t>(m) * N + r0] = from_f32<half_t>(acc0[m]); y[static_cast<size_t>(m) * N + r1] = from_f32<half_t>(acc1[m]); }
            }
        }
    }
}

// ---- V2: wave-autonomous, x (and the norm) in registers, no LDS, no barrier.  M = 1, K = KCH*512 ----
template <int EPI, bool NORM, int KCH, int RPW /*rows per wave-iteration: 2 or 4*/>
__global__ __launch_bounds__(256) void gemv_v2(const half_t *__restrict__ x, const half_t *__restrict__ W, half_t *__restrict__ y, int N,
                                               const half_t *__restrict__ gamma, float eps) {
    constexpr int K = KCH * 512;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half_n = N >> 1;
    constexpr int PR = RPW / 2;  // pairs per iteration
    const int npairs = (EPI == EPI_SWIGLU) ? half_n : (N >> 1);
    const int ngroups = npairs / PR;
    const int stride = gridDim.x * 4;
    int grp = blockIdx.x * 4 + wave;
    // x chunks of this lane
    half8_t xr[KCH];
    const half8_t *xg = reinterpret_cast<const half8_t *>(x);
#pragma unroll
    for (int j = 0; j < KCH; ++j) xr[j] = xg[lane + 64 * j];
    half8_t wb[RPW][KCH];
    auto issue = [&](int gidx) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int pair = gidx * PR + (r >> 1);
            const int row = (EPI == EPI_SWIGLU) ? pair + (r & 1) * half_n : 2 * pair + (r & 1);
            const half8_t *w = reinterpret_cast<const half8_t *>(W + static_cast<size_t>(row) * K);
#pragma unroll
            for (int j = 0; j < KCH; ++j) wb[r][j] = load_nt(w + lane + 64 * j);
        }
    };
    if (grp < ngroups) issue(grp);
    if (NORM) {
        const half8_t *gm = reinterpret_cast<const half8_t *>(gamma);
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < KCH; ++j) ss = dot8(xr[j], xr[j], ss);
        ss = wave_sum(ss);
        const float inv = rsqrtf(ss / static_cast<float>(K) + eps);
#pragma unroll
        for (int j = 0; j < KCH; ++j) {
            const half8_t g = gm[lane + 64 * j];
#pragma unroll
            for (int e = 0; e < 8; ++e) xr[j][e] = from_f32<half_t>(to_f32(xr[j][e]) * to_f32(g[e]) * inv);
        }
    }
    for (; grp < ngroups; grp += stride) {
        float acc[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            acc[r] = 0.f;
#pragma unroll
            for (int j = 0; j < KCH; ++j) acc[r] = dot8(wb[r][j], xr[j], acc[r]);
        }
        const int nxt = grp + stride;
        if (nxt < ngroups) issue(nxt);
#pragma unroll
        for (int r = 0; r < RPW; ++r) acc[r] = wave_sum(acc[r]);
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < PR; ++q) {
                const int pair = grp * PR + q;
                if (EPI == EPI_SWIGLU) y[pair] = from_f32<half_t>((acc[2 * q] / (1.0f + expf(-acc[2 * q]))) * acc[2 * q + 1]);
                else { y[2 * pair] = from_f32<half_t>(acc[2 * q]); y[2 * pair + 1] = from_f32<half_t>(acc[2 * q + 1]); }
            }
        }
    }
}

// ---- V0r1: one row per wave (K large) ----
template <int U>
__global__ __launch_bounds__(256) void gemv_v0_r1(const half_t *__restrict__ x, const half_t *__restrict__ W, half_t *__restrict__ y, int K, int N) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half8_t *xs = reinterpret_cast<half8_t *>(smem_raw);
    const int nch = K >> 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const half8_t *xg = reinterpret_cast<const half8_t *>(x);
        for (int i = tid; i < nch; i += 256) xs[i] = xg[i];
    }
    __syncthreads();
    for (int row = blockIdx.x * 4 + wave; row < N; row += gridDim.x * 4) {
        const half8_t *w0 = reinterpret_cast<const half8_t *>(W + static_cast<size_t>(row) * K);
        float acc = 0.f;
        for (int c = lane; c < nch; c += 64 * U) {
            half8_t a0[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int cc = c + 64 * u; if (cc < nch) a0[u] = load_nt(w0 + cc); }
#pragma unroll
            for (int u = 0; u < U; ++u) { const int cc = c + 64 * u; if (cc < nch) acc = dot8(a0[u], xs[cc], acc); }
        }
        acc = wave_sum(acc);
        if (lane == 0) y[row] = from_f32<half_t>(acc);
    }
}
// ---- V3: K split over the 4 waves of a workgroup, x slice in registers, RP row pairs per iteration ----
template <int RPW, int XC /*max x chunks per lane*/>
__global__ __launch_bounds__(256) void gemv_v3(const half_t *__restrict__ x, const half_t *__restrict__ W, half_t *__restrict__ y, int K, int N) {
    __shared__ float red[4][RPW];
    const int nch = K >> 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // chunk j of this lane: cc = (j*4 + wave)*64 + lane
    half8_t xr[XC];
    const half8_t *xg = reinterpret_cast<const half8_t *>(x);
#pragma unroll
    for (int j = 0; j < XC; ++j) { const int cc = (j * 4 + wave) * 64 + lane; xr[j] = cc < nch ? xg[cc] : half8_t{0,0,0,0,0,0,0,0}; }
    const int ngroups = N / RPW;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        half8_t wb[RPW][XC];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const half8_t *w = reinterpret_cast<const half8_t *>(W + static_cast<size_t>(grp * RPW + r) * K);
#pragma unroll
            for (int j = 0; j < XC; ++j) { const int cc = (j * 4 + wave) * 64 + lane; if (cc < nch) wb[r][j] = load_nt(w + cc); else wb[r][j] = half8_t{0,0,0,0,0,0,0,0}; }
        }
        float acc[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            acc[r] = 0.f;
#pragma unroll
            for (int j = 0; j < XC; ++j) acc[r] = dot8(wb[r][j], xr[j], acc[r]);
            acc[r] = wave_sum(acc[r]);
        }
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) red[wave][r] = acc[r];
        }
        __syncthreads();
        if (threadIdx.x < RPW) y[grp * RPW + threadIdx.x] = from_f32<half_t>(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
    }
}

#define HC(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Shape { const char *name; int N, K, epi; };

int main(int argc, char **argv) {
    const int NSETS = 12, REPS = 5;
    Shape shapes[] = {{"qkv 12288x4096", 12288, 4096, EPI_NONE}, {"o 4096x4096", 4096, 4096, EPI_NONE},
                      {"gate_up 22016x4096 swiglu", 22016, 4096, EPI_SWIGLU}, {"down 4096x11008", 4096, 11008, EPI_NONE},
                      {"lm_head 32000x4096", 32000, 4096, EPI_NONE}};
    hipStream_t st;
    HC(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    half_t *x, *y, *gamma;
    HC(hipMalloc(&x, 11008 * 2 * 8)); HC(hipMalloc(&y, 32000 * 2 * 8)); HC(hipMalloc(&gamma, 11008 * 2));
    HC(hipMemset(x, 0x3c, 11008 * 2 * 8)); HC(hipMemset(gamma, 0x3c, 11008 * 2));
    const int maxwg = argc > 1 ? atoi(argv[1]) : 2048;
    for (const Shape &s : shapes) {
        const size_t bytes = static_cast<size_t>(s.N) * s.K * 2;
        std::vector<half_t *> W(NSETS);
        for (auto &w : W) { HC(hipMalloc(&w, bytes)); HC(hipMemset(w, 0x11, bytes)); }
        const int npairs = s.epi == EPI_SWIGLU ? s.N / 2 : (s.N + 1) / 2;
        int wgs = (npairs + 3) / 4; if (wgs > maxwg) wgs = maxwg;
        auto timeit = [&](const char *vname, auto launch) {
            float best = 1e9f, avg = 0;
            for (int r = 0; r < REPS; ++r) {
                HC(hipEventRecord(e0, st));
                for (int i = 0; i < NSETS; ++i) launch(W[i]);
                HC(hipEventRecord(e1, st));
                HC(hipEventSynchronize(e1));
                float ms; HC(hipEventElapsedTime(&ms, e0, e1));
                const float us = ms * 1e3f / NSETS;
                if (r) { best = us < best ? us : best; avg += us / (REPS - 1); }
            }
            printf("  %-34s avg %7.2f us  best %7.2f us  -> %6.2f TB/s (best)\n", vname, avg, best, bytes / best / 1e6);
        };
        printf("%s  (%.1f MB, %d WGs)\n", s.name, bytes / 1e6, wgs);
        const size_t lds = static_cast<size_t>(s.K) * 2;
        if (s.epi == EPI_SWIGLU) {
            timeit("V0 old U=8", [&](half_t *w) { gemv_v0<1, EPI_SWIGLU, 8><<<wgs, 256, lds, st>>>(x, w, y, s.K, s.N); });
            timeit("V1 pipelined", [&](half_t *w) { GemvArgs a{x, w, y, s.K, s.N, nullptr, nullptr, nullptr, nullptr, 0.f}; gemv_f16_kernel<1, EPI_SWIGLU, false><<<wgs, 256, lds, st>>>(a); });
            timeit("V1 pipelined + norm", [&](half_t *w) { GemvArgs a{x, w, y, s.K, s.N, nullptr, nullptr, gamma, nullptr, 1e-5f}; gemv_f16_kernel<1, EPI_SWIGLU, true><<<wgs, 256, lds, st>>>(a); });
            timeit("V2 wave regs RPW=2", [&](half_t *w) { gemv_v2<EPI_SWIGLU, false, 8, 2><<<wgs, 256, 0, st>>>(x, w, y, s.N, gamma, 1e-5f); });
            timeit("V2 wave regs RPW=2 + norm", [&](half_t *w) { gemv_v2<EPI_SWIGLU, true, 8, 2><<<wgs, 256, 0, st>>>(x, w, y, s.N, gamma, 1e-5f); });
            for (int g : {344, 688, 1376}) {
                char nm[64]; snprintf(nm, sizeof nm, "V2 RPW=2 + norm, %d WGs", g);
                timeit(nm, [&](half_t *w) { gemv_v2<EPI_SWIGLU, true, 8, 2><<<g, 256, 0, st>>>(x, w, y, s.N, gamma, 1e-5f); });
            }
        } else {
            timeit("V0 old U=8", [&](half_t *w) { gemv_v0<1, EPI_NONE, 8><<<wgs, 256, lds, st>>>(x, w, y, s.K, s.N); });
            timeit("V0 old U=4", [&](half_t *w) { gemv_v0<1, EPI_NONE, 4><<<wgs, 256, lds, st>>>(x, w, y, s.K, s.N); });
            timeit("V1 pipelined", [&](half_t *w) { GemvArgs a{x, w, y, s.K, s.N, nullptr, nullptr, nullptr, nullptr, 0.f}; gemv_f16_kernel<1, EPI_NONE, false><<<wgs, 256, lds, st>>>(a); });
            timeit("V1 pipelined + norm", [&](half_t *w) { GemvArgs a{x, w, y, s.K, s.N, nullptr, nullptr, gamma, nullptr, 1e-5f}; gemv_f16_kernel<1, EPI_NONE, true><<<wgs, 256, lds, st>>>(a); });
            if (s.K == 11008) {
                timeit("V0r1 1 row/wave U=8 (1024 WGs)", [&](half_t *w) { gemv_v0_r1<8><<<1024, 256, lds, st>>>(x, w, y, s.K, s.N); });
                timeit("V0r1 1 row/wave U=11 (1024 WGs)", [&](half_t *w) { gemv_v0_r1<11><<<1024, 256, lds, st>>>(x, w, y, s.K, s.N); });
                timeit("V0 U=11", [&](half_t *w) { gemv_v0<1, EPI_NONE, 11><<<wgs, 256, lds, st>>>(x, w, y, s.K, s.N); });
                timeit("V3 ksplit RPW=2 (2048 WGs)", [&](half_t *w) { gemv_v3<2, 6><<<2048, 256, 0, st>>>(x, w, y, s.K, s.N); });
                timeit("V3 ksplit RPW=4 (1024 WGs)", [&](half_t *w) { gemv_v3<4, 6><<<1024, 256, 0, st>>>(x, w, y, s.K, s.N); });
                timeit("V3 ksplit RPW=4 (512 WGs x2)", [&](half_t *w) { gemv_v3<4, 6><<<512, 256, 0, st>>>(x, w, y, s.K, s.N); });
                timeit("V3 ksplit RPW=2 (1024 WGs x2)", [&](half_t *w) { gemv_v3<2, 6><<<1024, 256, 0, st>>>(x, w, y, s.K, s.N); });
            }
            if (s.K == 4096) {
                for (int g : {256, 384, 512, 768, 1024}) {
                    char nm[64]; snprintf(nm, sizeof nm, "V2 RPW=2 + norm, %d WGs", g);
                    timeit(nm, [&](half_t *w) { gemv_v2<EPI_NONE, true, 8, 2><<<g, 256, 0, st>>>(x, w, y, s.N, gamma, 1e-5f); });
                }
                timeit("V3 ksplit RPW=8 (N/8 WGs)", [&](half_t *w) { gemv_v3<8, 2><<<s.N / 8, 256, 0, st>>>(x, w, y, s.K, s.N); });
                timeit("V2 wave regs RPW=2", [&](half_t *w) { gemv_v2<EPI_NONE, false, 8, 2><<<wgs, 256, 0, st>>>(x, w, y, s.N, gamma, 1e-5f); });
                timeit("V2 wave regs RPW=2 + norm", [&](half_t *w) { gemv_v2<EPI_NONE, true, 8, 2><<<wgs, 256, 0, st>>>(x, w, y, s.N, gamma, 1e-5f); });
                timeit("V2 wave regs RPW=4 + norm", [&](half_t *w) { gemv_v2<EPI_NONE, true, 8, 4><<<(npairs / 2 + 3) / 4 > maxwg ? maxwg : (npairs / 2 + 3) / 4, 256, 0, st>>>(x, w, y, s.N, gamma, 1e-5f); });
            }
        }
        for (auto &w : W) HC(hipFree(w));
    }
    return 0;
}

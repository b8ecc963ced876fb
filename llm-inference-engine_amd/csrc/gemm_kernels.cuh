// Dense linear kernels for gfx950 behind llmie_linear / llmie_batched_gemm.
//
//   gemv_f16_kernel        M <= 8 tokens, fp16 W[N,K]: weight streaming, one wave per row pair,
//                          16-byte non-temporal loads straight to VGPRs (no LDS round trip for the
//                          stream), x staged once per workgroup in LDS, v_dot2_f32_f16 accumulate,
//                          wave64 butterfly reduce.  HBM-bound: algorithmic bytes = N*K*2.
//   skinny_mfma_f16_kernel 1 <= M <= 64, fp16 W[N,K]: one 16-row weight tile per workgroup, the
//                          waves split K, v_mfma_f32_16x16x32_f16 with W as the A operand (so the
//                          weight fragment is one 16-byte load per lane) and x^T as B, LDS reduce
//                          across the K-split.  HBM-bound.
//   tiled_mfma_f16_kernel  M > 64 (prefill): 128x128x32 LDS-tiled MFMA GEMM.  MFMA-bound.
//   generic_gemm_kernel    any dtype/transposition/shape: 64x64x16 LDS-tiled fp32 FMA.
//
// Epilogues fused here (the reference runs them as separate kernels): + bias[N], + residual[M,N],
// SwiGLU over row pairs (i, I+i) of a gate_up matrix.
#pragma once
#include "device_utils.cuh"

namespace llmie {

enum : int { EPI_NONE = 0, EPI_SWIGLU = 1 };

typedef float floatx4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// GEMV: y[m, r] = sum_k x[m,k] * W[r,k]
// grid: any (grid-stride over row pairs), block 256 (4 waves), dynamic LDS = M*K*2 bytes.
// EPI_NONE  : pair p = rows (2p, 2p+1); y[M,N]
// EPI_SWIGLU: pair p = rows (p, p+N/2); y[M,N/2] = silu(gate)*up
// ------------------------------------------------------------------------------------------
template <int M, int EPI, int U>
__global__ __launch_bounds__(256) void gemv_f16_kernel(const half_t *__restrict__ x,
                                                       const half_t *__restrict__ W,
                                                       half_t *__restrict__ y, int K, int N,
                                                       const half_t *__restrict__ bias,
                                                       const half_t *__restrict__ residual) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half8_t *xs = reinterpret_cast<half8_t *>(smem_raw);  // [M][K/8]
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
        int r0, r1;
        if constexpr (EPI == EPI_SWIGLU) {
            r0 = pair;
            r1 = pair + half_n;
        } else {
            r0 = 2 * pair;
            r1 = min(2 * pair + 1, N - 1);  // odd N: duplicate the last row, result discarded
        }
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
                if constexpr (EPI == EPI_SWIGLU) {
                    const float g = acc0[m], u = acc1[m];
                    y[static_cast<size_t>(m) * half_n + pair] = from_f32<half_t>((g / (1.0f + expf(-g))) * u);
                } else {
                    float v0 = acc0[m], v1 = acc1[m];
                    if (bias) {
                        v0 += to_f32(bias[r0]);
                        v1 += to_f32(bias[r1]);
                    }
                    if (residual) {
                        v0 += to_f32(residual[static_cast<size_t>(m) * N + r0]);
                        if (2 * pair + 1 < N) v1 += to_f32(residual[static_cast<size_t>(m) * N + r1]);
                    }
                    y[static_cast<size_t>(m) * N + r0] = from_f32<half_t>(v0);
                    if (2 * pair + 1 < N) y[static_cast<size_t>(m) * N + r1] = from_f32<half_t>(v1);
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
                                                                  const half_t *__restrict__ residual) {
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
    for (int s0 = wave * U; s0 < ksteps; s0 += NW * U) {
        half8_t a[U][NT], b[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = s0 + u;
            if (s < ksteps) {
#pragma unroll
                for (int t = 0; t < NT; ++t) a[u][t] = load_nt(reinterpret_cast<const half8_t *>(wp[t] + 32 * s));
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    b[u][j] = xok[j] ? *reinterpret_cast<const half8_t *>(xp[j] + 32 * s) : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (s0 + u < ksteps) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int j = 0; j < MT; ++j)
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u][t], b[u][j], acc[t][j], 0, 0, 0);
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
// Generic fallback: C[M,N] = A[M,K] * op(B), any T in {float, half}, fp32 accumulate,
// 64x64 block tile, 16-deep k tile, 256 threads each owning a 4x4 micro tile.
// Batched via blockIdx.z with dense strides.  B is [N,K] if TRANS_B else [K,N].
// ------------------------------------------------------------------------------------------
template <typename T, bool TRANS_B>
__global__ __launch_bounds__(256) void generic_gemm_kernel(const T *__restrict__ A, const T *__restrict__ B,
                                                           T *__restrict__ C, int M, int N, int K,
                                                           size_t strideA, size_t strideB, size_t strideC,
                                                           const T *__restrict__ bias,
                                                           const T *__restrict__ residual) {
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

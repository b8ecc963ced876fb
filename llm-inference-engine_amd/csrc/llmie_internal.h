// Internal (non-ABI) entry points shared between translation units of libllmie.so.
#pragma once
#include "device_utils.cuh"

namespace llmie {

enum : int { EPI_NONE_ = 0, EPI_SWIGLU_ = 1 };

// fp16 activations, fp16 W[N,K]; epi = 0 plain (bias/residual optional), 1 = SwiGLU over row
// pairs (i, N/2+i) writing y[M, N/2].  Defined in linear.hip.
// Caller-owned fp32 scratch for split-K slabs (nothing on the compute path allocates).  p == nullptr: the split-K forms are
// not available to the call (fp16 falls back to its non-split kernels; shapes that only have a split-K form fail with
// LLMIE_ERR_WORKSPACE); p != nullptr but too small / not 16-byte aligned: LLMIE_ERR_WORKSPACE.
struct SlabWs {
    float *p;
    size_t floats;
};
size_t linear_splitk_ws_floats(int wbits, int M, int K, int N);   // 0 = the shape has no split-K form
int linear_f16_nk(const half_t *x, const half_t *W, half_t *y, int M, int K, int N, int epi,
                  const half_t *bias, const half_t *residual, SlabWs ws, hipStream_t st);

// same with rmsnorm(x + pre_bias) * gamma fused in front (GEMV path only; LLMIE_ERR_UNSUPPORTED otherwise)
bool gemv_f16_eligible(int M, int K, const void *x, const void *W);
int linear_f16_nk_norm(const half_t *x, const half_t *W, half_t *y, int M, int K, int N, int epi, const half_t *bias,
                       const half_t *residual, const half_t *gamma, const half_t *pre_bias, float eps, hipStream_t st);

// split-K skinny MFMA GEMM (fp16 or int8 weights), any M processed 64 tokens per pass; epi may be SwiGLU; linear.hip
// partial products of a split-K projection: fp32 slabs [KS][M][N] in the caller's SlabWs
struct SplitKSlabs {
    float *slab;
    int KS, M, N;
};
enum : int { WF_FP8 = 108 };  // wbits code of e4m3 weights + e4m3 activations (16 / 8 = fp16 / int8 weights, fp16 activations)
int linear_splitk_partial(int wbits, const void *x, const void *W, int M, int K, int N, hipStream_t st, SplitKSlabs *out,
                          SlabWs ws, const half_t *gscale = nullptr /* wbits 4: group-128 scales [N, K/128], applied in the kernel */);
int splitk_finalize(const SplitKSlabs &sk, const SlabScale &sc, half_t *y, int epi, const half_t *bias, const half_t *residual,
                    hipStream_t st);
bool splitk_rownorm_eligible(int N);
// y (fp16, may be null) and/or xq + xscale (per-token e4m3, may be null) receive the normalised row
int splitk_rownorm(const SplitKSlabs &sk, const SlabScale &wscale, const half_t *bias, half_t *resid, const half_t *gamma,
                   float eps, half_t *y, uint8_t *xq, float *xscale, hipStream_t st);
// RMSNorm (plain or add-residual-bias form) emitting per-token e4m3 instead of the normalised fp16 row; norm.hip
bool rmsnorm_quant_eligible(int hidden);
int rmsnorm_quant_f16(const half_t *x, half_t *resid, const half_t *bias, const half_t *gamma, float eps, int tokens, int hidden,
                      bool fused, uint8_t *xq, float *xscale, hipStream_t st);
// out-of-place RMSNorm (x untouched): the prefill's residual stream stays in its own buffer; norm.hip
bool rmsnorm_oop_eligible(int hidden);
int rmsnorm_oop_f16(const half_t *x, half_t *y, const half_t *gamma, float eps, int tokens, int hidden, hipStream_t st);
// per-token e4m3 quantisation of fp16 rows (scale amax/448); fp8_linear.hip
int quantize_rows_fp8(const half_t *x, uint8_t *xq, float *xscale, int M, int K, hipStream_t st);
int linear_splitk(int wbits, const half_t *x, const void *W, const half_t *scale, half_t *y, int M, int K, int N, int epi,
                  const half_t *bias, const half_t *residual, SlabWs ws, hipStream_t st);
// 256 x 256 LDS-DMA tiled GEMM (gemm256.cuh; K % 64 == 0 fp16, K % 128 == 0 fp8; 16-byte aligned operands); linear.hip
bool gemm256_fills(int M, int N);
// wq = 8: W is int8 [N, K] under fp16 activations and `wscale` carries its fp16 per-row scales (gemm8p.cuh, WQ = 8)
void gemm256_launch(bool fp8, const void *x, const void *W, half_t *y, int M, int N, int K, const half_t *bias,
                    const half_t *residual, const float *xscale, const float *wscale, hipStream_t st, int wq = 0);
bool gemm256_swiglu_fills(int M, int two_inter);
void gemm256_swiglu_launch(bool fp8, const void *x, const void *W, half_t *y, int M, int two_inter, int K, const float *xscale,
                           const float *wscale, hipStream_t st, int wq = 0);
bool g8p_w8_eligible(int M, int K, int N, const void *x, const void *wq, const void *scale, const void *y);
// QKV projection with RoPE + KV-cache append as its epilogue (gemm8p.cuh ROPE forms): kind 0 = fp16, 1 = e4m3 operands, 8 = int8
// weights (wscale = fp16 row scales); q columns -> qkv (rotated), k / v columns -> the caches only; linear.hip
bool gemm256_qkv_rope_eligible(int kind, int M, int N, int K, const void *x, const void *W, const void *wscale, const void *qkv);
// rap: the epilogue's layer-invariant operands in DEVICE memory (prefill_token_table writes them); bias: the layer's QKV bias or null
void gemm256_qkv_rope_launch(int kind, const void *x, const void *W, half_t *qkv, int M, int N, int K, const float *xscale,
                             const float *wscale, const half_t *bias, const QkvRopeArgs *rap, int layer, hipStream_t st);
bool g8p_w8_swiglu_eligible(int M, int K, int two_inter, const void *x, const void *wq, const void *scale, const void *y);
// fp16 image of int8 / int4 weights (row scales / group scales applied, one rounding): the operand of the prefill-sized
// projections that have no in-kernel de-quantising form; quant_linear.hip
int dequantize_weights_f16(int wbits, const void *wq, const half_t *scale, half_t *w16, int N, int K, int group, hipStream_t st);
// rows from which a weight-only projection runs as an MFMA-bound tiled GEMM (prefill) instead of split-K passes
constexpr int kWqPrefillRows = 192;
// bytes of fp16 scratch linear_wq needs at M rows beside the split-K slabs (0: none)
size_t linear_wq_dequant_bytes(int wbits, int M, int K, int N);
// quantised-weight (int8 / int4) decode GEMV on the K-split kernel; defined in linear.hip
struct GemvArgs;
bool ksplit_eligible(int M, int K, int wbits);
bool gemv_q_launch(int wbits, int M, const GemvArgs &a, hipStream_t st);
bool gemv_fp8_launch(int M, const GemvArgs &a, hipStream_t st);
// fp8 linear on the GEMV path (M <= 8, ksplit_eligible(M, K, 8)); optional fused norm prologue / SwiGLU epilogue; fp8_linear.hip
int linear_fp8_gemv(const half_t *x, const uint8_t *wq, const float *wscale, half_t *y, int M, int K, int N, int epi,
                    const half_t *bias, const half_t *residual, const half_t *gamma, const half_t *pre_bias, float eps,
                    hipStream_t st);
// llmie_linear_fp8 with the activation scratch and the split-K slabs as separate areas; fp8_linear.hip
int linear_fp8(const half_t *x, const uint8_t *w_fp8, const float *w_scale, half_t *y, int M, int K, int N, const half_t *bias,
               const half_t *residual, void *act_ws, size_t act_ws_bytes, SlabWs slabs, hipStream_t st);
// weight-only int8/int4 linear with optional fused norm prologue / SwiGLU epilogue (quant_linear.hip)
// M >= kWqPrefillRows (prefill): int8 through the eight-phase kernels' int8 form where eligible, else (and int4) a de-quantised
// fp16 image in `deq` (linear_wq_dequant_bytes) + the fp16 GEMM; without `deq` those shapes keep the split-K passes
int linear_wq(int wbits, const half_t *x, const void *wq, const half_t *scale, half_t *y, int M, int K, int N, int group,
              int epi, const half_t *bias, const half_t *residual, const half_t *gamma, const half_t *pre_bias, float eps,
              SlabWs ws, hipStream_t st, void *deq = nullptr, size_t deq_bytes = 0);

// ---- packed-weight batch-decode projections (pk_gemm.cuh / pk_linear.hip): 1 <= M <= 32 rows on tile-packed weight images ----
enum : int { PKF_F16 = 16, PKF_I8 = 8, PKF_I4 = 4, PKF_FP8 = 108 };                 // = PK_F16 ... of pk_gemm.cuh
enum : int { PKE_PLAIN = 0, PKE_SWIGLU = 1 };
enum : int { PKX_X = 1, PKX_Y = 2, PKX_RES = 4 };                                    // operands in the x32 activation layout
size_t pk_packed_bytes(int wf, int N, int K, int swiglu);
size_t pk_packed_scale_bytes(int wf, int N, int K, int swiglu);   // int4: the group-scale image beside the weight image (else 0)
int pk_pack(int wf, const void *src, const void *src_scale, void *dst, void *dst_scale, int N, int K, int swiglu, hipStream_t st);
// image -> fp16 row-major [N, K] with the scales applied (fp16 / int8 / int4 images; scale = the caller's row / group-128 scales)
int pk_unpack_f16(int wf, const void *packed, const half_t *scale, half_t *w16, int N, int K, int swiglu, hipStream_t st);
bool pk_eligible(int wf, int M, int K, int N, int epi);
size_t pk_slab_floats(int wf, int M, int K, int N);
int pk_linear(int wf, const half_t *x, const void *Wp, const void *scale, half_t *y, int M, int K, int N, int epi, int x32_flags,
              const half_t *residual, const half_t *gamma, const half_t *pre_bias, float eps, float *slab_ws, size_t slab_ws_floats,
              hipStream_t st);
int x32_convert(const half_t *src, half_t *dst, int M, int K, int to_x32, hipStream_t st);
// Persistent chain (pk_chain_kernel): up to 5 dependent phases -- each a pk_linear of the same format and row count whose x is an
// x32 image -- in ONE launch with grid barriers in between and the next phase's weight ring prefetched across each barrier.
// `sync`: pk_chain_sync_bytes() ZEROED bytes of this launch; `err`: the owner's device error word (non-zero after a barrier
// timed out).  A failed pk_chain_add leaves the chain unusable (the caller falls back to the launch sequence).
struct PkChain {
    int wf, M, nph, lds;
    bool ok;
    unsigned long long *stamps;   // diagnostic: device buffer [256][16] of phase-edge timestamps (null: none)
    alignas(16) unsigned char args[1024];
};
void pk_chain_begin(PkChain *ch, int wf, int M);
int pk_chain_add(PkChain *ch, int slot /* 0 = O, 1 = gate/up, 2 = down, 4 = next QKV */, const half_t *x, const void *Wp, const void *scale, half_t *y, int K, int N, int epi, int x32_flags,
                 const half_t *residual, const half_t *gamma, const half_t *pre_bias, float eps, float *slab_ws, size_t slab_ws_floats);
int pk_chain_launch(PkChain *ch, unsigned *sync, unsigned *err, hipStream_t st);
size_t pk_chain_sync_bytes();

// decode attention with optional RoPE (rope may be null) fused in front (rope = [max_pos][head_size/2] (cos,sin) table); attention_decode.hip
int decoder_mha_rope(const void *qkv, const void *qkv_bias, void *k_cache, void *v_cache, void *out, int layer, int batch,
                     int head_num, int kv_head_num, int head_size, int max_seq_len, int step, const int32_t *step_dev,
                     void *workspace, size_t workspace_bytes, const float2 *rope, int rot_dim,
                     int32_t *tickets /* [batch,kvh] zeroed arrival counters: in-launch merge; null = merge kernel */,
                     llmie_dtype dtype, hipStream_t st,
                     const SplitKSlabs *qkv_slabs = nullptr /* q/k/v read from the QKV projection's split-K slabs (qkv unused) */,
                     const SlabScale *qkv_scale = nullptr,
                     int kv_fp8 = 0 /* caches are e4m3 bytes, stored = e4m3(x / scale) */, float k_scale = 1.f, float v_scale = 1.f,
                     const int32_t *block_table = nullptr /* paged cache: [batch, max_pages] pool pages of 128 tokens; the cache
                                                             pointers are then pools [L, num_pages, kvh, 128, hs] */,
                     int max_pages = 0, int num_pages = 0,
                     int ragged = 0 /* step_dev is an array: step_dev[b] = context length of sequence b incl. this token */,
                     int out_x32 = 0 /* out is the x32 activation image (batch <= 32) instead of row-major [batch, H] */);

// fused tail of a decode step (topk_sampling.hip): round 1 of the top-k alone, and round 2 + sampling (+ the next step's input
// embedding into next_hidden, + *step_dev += 1 by the last row to finish; `ticket` = one zeroed word) in one launch
int topk_round1_only(const void *probs, int32_t *tmp_ids, void *tmp_vals, int rows, int vocab, int K, int bpr, llmie_dtype dtype, hipStream_t st);
int decode_tail(const int32_t *tmp_ids, const void *tmp_vals, int32_t *ids, void *vals, int K, int bpr, int32_t *seq_len, uint8_t *finished,
                int32_t *out_id, int rows, int step, int32_t *step_dev, int end_id, int vocab, const void *embed, void *next_hidden, int hidden,
                int advance, unsigned *ticket, llmie_dtype dtype, hipStream_t st);

// prefill attention (RoPE + KV append + flash attention) on the packed QKV buffer; prefill.hip
int prefill_attention_f16(half_t *qkv, const half_t *qkv_bias, void *k_cache, void *v_cache, half_t *out,
                          const int32_t *cum_seqlens, const int32_t *history_len, const float2 *rope, int layer, int batch,
                          int num_tokens, int max_q_len, int head_num, int kv_head_num, int head_size, int max_seq_len,
                          int rotary_dim, hipStream_t st, int kv_fp8 = 0 /* caches are e4m3 bytes */, float k_scale = 1.f,
                          float v_scale = 1.f, const int32_t *block_table = nullptr /* paged cache, see decoder_mha_rope */,
                          int max_pages = 0, int num_pages = 0,
                          int rope_done = 0 /* RoPE + append already done by the QKV projection's epilogue (gemm256_qkv_rope_launch) */);
// short prefills: slab consumer of the QKV projection with RoPE + KV-cache append folded in (q -> qkv, k / v -> the caches only)
bool splitk_finalize_qkv_rope_eligible(const SplitKSlabs &sk, int head_size, const void *qkv, const void *bias);
int splitk_finalize_qkv_rope(const SplitKSlabs &sk, const SlabScale &sc, half_t *qkv, const half_t *qkv_bias, void *k_cache, void *v_cache,
                             const int32_t *cum_seqlens, const int32_t *history_len, const float2 *rope, int layer, int batch, int head_num,
                             int kv_head_num, int max_seq_len, int rotary_dim, hipStream_t st, int kv_fp8, float k_scale, float v_scale,
                             const int32_t *block_table, int max_pages, int num_pages);
// tok_b[t] / tok_tpos[t] = sequence / cache position (history + position) of packed token t: operands of that epilogue
// (also copies `args` -- whose tok_b / tok_tpos it fills in -- to args_dev)
int prefill_token_table(const int32_t *cum_seqlens, const int32_t *history_len, int batch, int num_tokens, int32_t *tok_b, int32_t *tok_tpos,
                        QkvRopeArgs args, QkvRopeArgs *args_dev, hipStream_t st);

}  // namespace llmie

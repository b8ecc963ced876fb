/*
 * llmie_oracle.h -- CPU oracle for the Llama-2 decoder hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and there only as the checker / reported CPU baseline.
 *
 * Plain C (gcc), fp32 storage, fp32 accumulation unless a function says
 * otherwise.  Each function restates the algorithm of one reference kernel
 * (or of the reference's own unit-test CPU checker where that checker is
 * sound) and cites the reference file:line it follows.  The reference is
 * CUDA-only and cannot be built in this image (no nvcc/cuBLAS/cuRAND/CUB; its
 * CPU checkers live inside .cu files that include CUDA headers), so parity is
 * pinned by the reference's own known answers, see tests/golden/ and
 * tests/test_oracle_golden.py.  Where no reference test pins a result
 * (batched GEMM, sampling RNG stream, decode attention: the reference checker
 * is self-declared broken) the header comment of the function says
 * "parity unpinned" and DESIGN.md repeats it.
 *
 * Paths are relative to the reference root (chongchen1999/llm-inference-engine
 * @ 2024_10_08).
 */
#ifndef LLMIE_ORACLE_H
#define LLMIE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/kernels/input_embedding.cu:4-22, checker tests/unit_tests/test_input_embedding.cu:15-23
 * out[t,:] = table[ids[t],:]                                   (bit-exact gather) */
void orc_input_embedding(const int32_t *ids, const float *table, float *out,
                         int num_tokens, int hidden);

/* src/kernels/cal_padding_offset.cu:17-43, doc example includes/cal_padding_offset.cuh:8-15
 * cum_seqlens[bs+1] = exclusive cumsum(lens); padding_offset is PACKED: the
 * first sum(lens) entries are written, entry for packed token t of sequence b
 * is sum_{j<b}(max_q_len - lens[j]); the rest is left untouched. */
void orc_cal_padding_offset(int32_t *padding_offset, int32_t *cum_seqlens,
                            const int32_t *lens, int batch, int max_q_len);

/* src/kernels/build_causal_mask.cu:4-23, checker tests/unit_tests/test_build_causal_mask.cu:13-31
 * mask[b,q,k] = (q<qlen[b]) && (k<klen[b]) && (k <= q + klen[b]-qlen[b]) as 0/1 */
void orc_build_causal_mask(float *mask, const int32_t *q_lens, const int32_t *k_lens,
                           int batch, int max_q_len, int max_k_len);

/* src/kernels/rmsnorm.cu:35-80, checker tests/unit_tests/test_rmsnorm.cu:10-27
 * resid = x (if resid != NULL); x = x * gamma * rsqrt(mean(x^2) + eps)  in place */
void orc_rmsnorm(float *x, float *resid, const float *gamma, float eps,
                 int num_tokens, int hidden);

/* src/kernels/add_residual_and_rmsnorm.cu:43-121 (the fp32 kernel is the spec; the
 * unit-test checker test_add_residual_and_rmsnorm.cu:12-43 is defective)
 * out += resid; resid = out; out += bias (opt); out = gamma*out*rsqrt(mean(out^2)+eps)
 * NOTE the residual is updated BEFORE the bias add, as the reference kernel does. */
void orc_fused_add_bias_residual_rmsnorm(float *resid, float *out, const float *bias,
                                         const float *gamma, float eps,
                                         int num_tokens, int hidden);

/* src/kernels/add_residual.cu:51-76, checker tests/unit_tests/test_add_residual.cu:10-21
 * out += resid */
void orc_add_residual(const float *resid, float *out, int num_tokens, int hidden);

/* src/kernels/linear.cu:10-87, checker tests/unit_tests/test_linear.cu:17-33
 * trans_b != 0:  y[M,N] = x[M,K] . W[N,K]^T   (W row-major [N,K], HF layout)
 * trans_b == 0:  y[M,N] = x[M,K] . W[K,N]
 * (intent of the reference; the checker pins trans_b only). */
void orc_linear(const float *x, const float *w, float *y, int M, int K, int N, int trans_b);

/* src/kernels/linear.cu:89-158 (cuBLAS strided batched; parity unpinned by any
 * reference test).  Per batch: C[m,n] = A[m,k] . B (B is [n,k] if trans_b else [k,n]) */
void orc_batched_gemm(const float *a, const float *b, float *c,
                      int batch, int m, int n, int k, int trans_b);

/* src/kernels/qkv_bias_and_rope.cu:5-79 + includes/rope_utils.cuh:6-19,
 * checker tests/unit_tests/test_qkv_bias_and_rope.cu:14-72 (hard-coded +64 -> hs/2).
 * Un-pad + transpose packed QKV[T, nh+2kvh, hs] into q[bs,nh,S,hs], k/v[bs,kvh,S,hs],
 * rotate-half RoPE on q,k at position history_len[b] + local_token; optional bias
 * [ (nh+2kvh)*hs ] is added before the rotation (reference never adds it: 9-K10;
 * pass NULL to match the reference).  Padded slots are left untouched. */
void orc_qkv_bias_transpose_rope(float *q, float *k, float *v, const float *qkv,
                                 const float *bias, const int32_t *padding_offset,
                                 const int32_t *history_len,
                                 int batch, int seq_len, int num_tokens,
                                 int head_num, int kv_head_num, int head_size,
                                 int rotary_dim, float rotary_base);

/* src/kernels/rope.cu:4-43 (fp32 kernel; intended semantics):
 * in place on qkv[bs, nh+2kvh, hs]: every q head and every kv head is rotated
 * exactly once at position step-1, row stride (nh+2kvh)*hs for every batch. */
void orc_rope_decode(float *qkv, int batch, int head_num, int kv_head_num, int head_size,
                     int step, int rotary_dim, float rotary_base);

/* src/kernels/decoder_self_attention.cu:56-188 (fp32 kernel math; correct batch
 * stride, any step -- SURVEY 9-K4; the unit-test checker is self-declared broken so
 * this is "parity unpinned" beyond the kernel source itself).
 * Per (b, q-head h): q,k,v (+bias opt) -> cache[layer,b,h/rep,step-1,:] = k,v ->
 * logits[t] = scale * q.K[t], t<step -> p = exp(l-max)/(sum+1e-6) -> out = sum p[t] V[t].
 * k_cache/v_cache point at the WHOLE cache [L,bs,kvh,max_seq,hs]. */
void orc_decoder_mha(const float *qkv, const float *qkv_bias,
                     float *k_cache, float *v_cache, float *out,
                     int layer, int batch, int head_num, int kv_head_num, int head_size,
                     int max_seq_len, int step);

/* src/kernels/concat_past_kv.cu:10-42
 * cache[layer,b,h,history[b]+t,:] = src[b,h,t,:] for t < cur_len[b]   (bit-exact copy) */
void orc_concat_kv(const float *src, float *cache, const int32_t *cur_len,
                   const int32_t *history_len, int layer, int batch, int kv_head_num,
                   int max_q_len, int max_seq_len, int head_size);

/* src/kernels/repeat_kv.cu:13-49 with the intended source head h/rep
 * dst[b,h,t,:] = cache[layer,b,h/rep,t,:] for t < ctx_len[b]; other slots untouched */
void orc_repeat_kv(const float *cache, float *dst, const int32_t *ctx_len,
                   int layer, int batch, int head_num, int kv_head_num,
                   int max_k_len, int max_seq_len, int head_size);

/* src/kernels/scale_and_mask_and_softmax.cu:64-127 with a true row max (9-K7)
 * p = softmax_k(scale*qk + (1-mask)*(-10000)), denominator + 1e-6 */
void orc_scale_mask_softmax(const float *qk, const float *mask, float *out, float scale,
                            int batch, int head_num, int q_len, int k_len);

/* src/kernels/transpose_and_remove_padding.cu:15-43
 * [bs,nh,S,hs] -> [T,nh,hs] dropping pads via padding_offset   (bit-exact copy) */
void orc_transpose_remove_padding(const float *src, float *dst, const int32_t *padding_offset,
                                  int num_tokens, int batch, int seq_len,
                                  int head_num, int head_size);

/* src/kernels/silu_and_mul.cu:25-41, checker tests/unit_tests/test_silu_and_mul.cu:16-32
 * out[t,i] = silu(in[t,0,i]) * in[t,1,i],  silu(x) = x / (1 + exp(-x)) */
void orc_silu_and_mul(const float *in, float *out, int num_tokens, int inter);

/* src/kernels/topk.cu:24-140 (intended: SURVEY 9-K8).  Per row the K largest
 * values, descending; ties -> lower id first; ids are row-local.
 * Known answer: tests/unit_tests/test_topk.cu:41-43 (probs[i]=i). */
void orc_topk(const float *probs, int32_t *ids, float *vals, int rows, int vocab, int K);

/* src/kernels/sampling.cu:14-71.  w_i = exp(v_i - v_0); thr = u * sum(w), u in (0,1];
 * first i with running (thr -= w_i) < 0 wins, else id[0]; id % vocab; seq_len++ unless
 * finished; finished = (id == end_id).  The reference draws u from cuRAND XORWOW
 * (seed=step, subsequence=batch) -- parity unpinned (no test checks the stream), so
 * u comes from orc_uniform_philox(seed=step, stream=batch) instead. */
void orc_sampling(const int32_t *topk_id, const float *topk_val, int32_t *seq_len,
                  uint8_t *finished, int32_t *out_id,
                  int batch, int K, int step, int end_id, int vocab);

/* Philox4x32-10, key = (seed, 0x4c4c4d49 "LLMI"), counter = (stream, 0, 0, 0);
 * returns ((x0 >> 8) + 1) * 2^-24 in (0,1].  Same integer recipe on the GPU. */
float orc_uniform_philox(uint32_t seed, uint32_t stream);

/* ---- weight-only quantisation (new work, no reference implementation; README.md:36-39
 * plans it).  Oracle = fp32 GEMM over the de-quantised weights. ---- */

/* y[M,N] = x[M,K] . (scale[n] * q[n,k])^T, q int8 row-major [N,K], per-row scale */
void orc_linear_w8(const float *x, const int8_t *wq, const float *scale, float *y,
                   int M, int K, int N);
/* int4: two nibbles per byte (low nibble = even k), value = nibble - 8,
 * scale per (row, group of `group` k): scale[n, k/group] */
void orc_linear_w4(const float *x, const uint8_t *wq, const float *scale, float *y,
                   int M, int K, int N, int group);

/* ---- composed decode step, used for layer-level parity and the cpu_baseline leg ---- */
typedef struct {
    int head_num, kv_head_num, head_size, inter_size, num_layers, vocab;
    int max_seq_len, rotary_dim;
    float rotary_base, rms_eps;
} orc_llama_cfg;

typedef struct {
    const float *attn_norm;   /* [H] */
    const float *qkv;         /* [(nh+2kvh)*hs, H] */
    const float *qkv_bias;    /* NULL or [(nh+2kvh)*hs] */
    const float *o;           /* [H, H] */
    const float *o_bias;      /* NULL or [H] */
    const float *ffn_norm;    /* [H] */
    const float *gate_up;     /* [2I, H] */
    const float *down;        /* [H, I] */
} orc_layer_weights;

/* src/layers/self_decoder.cpp:24-122 (+ self_attention.cpp:63-151, ffn.cpp:76-144):
 * runs num_layers decoder layers in place on hidden[bs,H] at `step`;
 * caches are [L,bs,kvh,max_seq,hs].  scratch must hold
 * bs*(H + (nh+2kvh)*hs + H + 3*I) floats. */
void orc_self_decoder(const orc_llama_cfg *cfg, const orc_layer_weights *layers,
                      float *hidden, float *k_cache, float *v_cache,
                      int batch, int step, float *scratch);

void orc_set_num_threads(int n); /* OpenMP team size used by the functions below (bench.py: the box's CPU share) */
int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif

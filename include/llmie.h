/*
 * llmie.h -- C ABI of the MI355X-native Llama-2 decoder hot path.
 *
 * Drop-in boundary for chongchen1999/llm-inference-engine (reference @ 2024_10_08).
 * The reference has no FFI of its own: its boundary is the set of C++ `launch*`
 * templates in src/kernels/includes/ *.cuh plus the layer classes in
 * src/layers/includes/ *.h.  Every entry point below replaces exactly one of those
 * (cited per function, paths relative to the reference root); the C++ templates
 * of the same names shipped in llm-inference-engine_amd/src/ are thin adaptors that
 * unpack TensorWrapper<T> into these calls, so user_entry.cpp / examples/cpp build
 * unchanged on top (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - plain pointers + sizes, no C++/torch types.  All data pointers are DEVICE
 *     pointers (HBM) unless the name ends in _host.
 *   - `dtype` selects the element type of every `void *` tensor of the call:
 *     LLMIE_F32 (float) or LLMIE_F16 (IEEE half); accumulation is always fp32.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  No
 *     entry point allocates, frees or synchronises: all of them are legal inside
 *     hipStreamBeginCapture/EndCapture (hipGraph).
 *   - return value: 0 on success, a negative llmie_status otherwise;
 *     llmie_last_error() returns a thread-local message.  The C++ adaptors turn a
 *     non-zero status into the reference's LLM_CHECK behaviour (std::runtime_error,
 *     src/utils/macro.h:74-94).
 *   - tensors are dense row-major; offsets are computed in 64-bit (the
 *     reference's int32 offsets overflow for the 7B KV cache at batch >= 8).
 */
#ifndef LLMIE_H
#define LLMIE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history (llmie_abi_version() returns the library's; a consumer compares it with the header it was built against):
 *   1  round 1.
 *   2  llmie_linear, llmie_linear_swiglu, llmie_linear_w8a16, llmie_linear_w4a16 gained (workspace, workspace_bytes) in front
 *      of the stream argument and llmie_linear_fp8_workspace_bytes gained N; w8a16 with M > 64 / w4a16 with M > 8 need a
 *      workspace (NULL -> LLMIE_ERR_UNSUPPORTED).  A version-1 consumer would pass its stream where the slab pointer goes.
 *   3  round 3: additions only are listed at the entries they concern (int8 / int4 weight-only prefill, decoder config flags). */
#define LLMIE_ABI_VERSION 3

typedef enum { LLMIE_F32 = 0, LLMIE_F16 = 1 } llmie_dtype;

typedef enum {
    LLMIE_OK = 0,
    LLMIE_ERR_INVALID_ARG = -1,   /* NULL pointer, non-positive size, shape mismatch */
    LLMIE_ERR_UNSUPPORTED = -2,   /* shape/dtype outside what the kernels implement   */
    LLMIE_ERR_LAUNCH = -3,        /* hipGetLastError() != hipSuccess after the launch */
    LLMIE_ERR_WORKSPACE = -4      /* caller-provided workspace too small              */
} llmie_status;

typedef void *llmie_stream; /* hipStream_t */

typedef enum {
    LLMIE_W_F16 = 0,     /* fp16 weights [N,K]                                   */
    LLMIE_W_INT8 = 1,    /* int8 [N,K] + fp16 per-row scale                      */
    LLMIE_W_INT4 = 2,    /* packed int4 [N,K/2] + fp16 scale per (row, group)    */
    LLMIE_W_FP8 = 3,     /* e4m3 [N,K] + fp32 per-row scale                      */
    LLMIE_W_F32 = 4      /* fp32 weights (dtype LLMIE_F32 engines only)          */
} llmie_weight_format;

int llmie_abi_version(void);
const char *llmie_last_error(void);
/* "gfx950" -- the only architecture this library carries code objects for */
const char *llmie_target_arch(void);

/* ------------------------------------------------------------------------- */
/* 1. per-kernel entry points (one per reference launcher)                   */
/* ------------------------------------------------------------------------- */

/* replaces launchInputEmbedding        src/kernels/input_embedding.cu:24-51
 * out[t,:] = table[ids[t],:]; ids outside [0,vocab) -> row of zeros is NOT written
 * (status LLMIE_OK; the row is left untouched, as the reference would fault). */
int llmie_input_embedding(const int32_t *ids, const void *table, void *out,
                          int num_tokens, int hidden, int vocab,
                          llmie_dtype dtype, llmie_stream stream);

/* replaces launchCalPaddingOffset      src/kernels/cal_padding_offset.cu:45-70
 * padding_offset is packed (first sum(lens) ints written), cum_seqlens has batch+1 ints */
int llmie_cal_padding_offset(int32_t *padding_offset, int32_t *cum_seqlens,
                             const int32_t *input_lengths, int batch, int max_q_len,
                             llmie_stream stream);

/* replaces launchBuildCausalMasks      src/kernels/build_causal_mask.cu:25-42 */
int llmie_build_causal_mask(void *mask, const int32_t *q_lens, const int32_t *k_lens,
                            int batch, int max_q_len, int max_k_len,
                            llmie_dtype dtype, llmie_stream stream);

/* replaces launchRMSNorm               src/kernels/rmsnorm.cu:130-159
 * resid (nullable) = x;  x = x * gamma * rsqrt(mean(x^2)+eps)  in place */
int llmie_rmsnorm(void *x, void *resid, const void *gamma, float eps,
                  int num_tokens, int hidden, llmie_dtype dtype, llmie_stream stream);

/* replaces launchFusedAddBiasResidualAndRMSNorm  src/kernels/add_residual_and_rmsnorm.cu:170-201
 * out += resid; resid = out; out += bias (nullable); out = gamma*out*rsqrt(mean(out^2)+eps) */
int llmie_fused_add_bias_residual_rmsnorm(void *resid, void *out, const void *bias,
                                          const void *gamma, float eps,
                                          int num_tokens, int hidden,
                                          llmie_dtype dtype, llmie_stream stream);

/* replaces launchAddResidual           src/kernels/add_residual.cu:51-76     out += resid */
int llmie_add_residual(const void *resid, void *out, int num_tokens, int hidden,
                       llmie_dtype dtype, llmie_stream stream);

/* replaces launchLinearGemm            src/kernels/linear.cu:10-87 (+ cublas_utils.cpp:29-93)
 * trans_b != 0: y[M,N] = x[M,K] . W[N,K]^T ; trans_b == 0: y = x . W[K,N].
 * bias (nullable, [N]) and residual (nullable, [M,N], may alias y) are fused epilogues the
 * reference does in separate kernels; pass NULL for the plain reference semantics.
 * workspace: the counterpart of the scratch the reference's cublasWrapper argument owns (cublas_utils.cpp:29-93) -- caller-owned
 * fp32 slabs of the split-K forms (decode / short-prefill batches of up to 192 rows), llmie_linear_workspace_bytes() bytes,
 * 16-byte aligned.  NOTHING on the compute path allocates: a first call may already run under hipGraph capture.  NULL: fp16
 * runs its non-split kernels (slower at 8 < M <= 192); a non-NULL workspace that is too small is LLMIE_ERR_WORKSPACE. */
size_t llmie_linear_workspace_bytes(llmie_weight_format fmt, int M, int K, int N);
int llmie_linear(const void *x, const void *w, void *y, int M, int K, int N, int trans_b,
                 const void *bias, const void *residual,
                 llmie_dtype dtype, void *workspace, size_t workspace_bytes, llmie_stream stream);

/* replaces launchLinearGemm(gate_and_up) + launchSiluAndMul   src/layers/ffn.cpp:105-122,
 * src/kernels/silu_and_mul.cu:61-82 in one kernel: w is the fused gate_up matrix [2I,K]
 * (rows [0,I) gate, [I,2I) up); y[M,I] = silu(x.Wg^T) * (x.Wu^T).  fp16 only (decode path).
 * workspace: llmie_linear_workspace_bytes(LLMIE_W_F16, M, K, two_inter), see llmie_linear; without it 64 < M needs
 * prefill-sized shapes. */
int llmie_linear_swiglu(const void *x, const void *w, void *y, int M, int K, int two_inter,
                        llmie_dtype dtype, void *workspace, size_t workspace_bytes, llmie_stream stream);

/* replaces launchLinearStridedBatchGemm src/kernels/linear.cu:89-158 (+ cublas_utils.cpp:95-154)
 * per batch i: C_i[m,n] = A_i[m,k] . B_i  (B_i is [n,k] if trans_b else [k,n]); dense strides */
int llmie_batched_gemm(const void *a, const void *b, void *c, int batch, int m, int n, int k,
                       int trans_b, llmie_dtype dtype, llmie_stream stream);

/* replaces launchFusedQKVAddBiasAndTransposeAndRope  src/kernels/qkv_bias_and_rope.cu:86-138
 * QKV[T, nh+2kvh, hs] -> q[bs,nh,S,hs], k/v[bs,kvh,S,hs]; RoPE at history_len[b]+local_token */
int llmie_qkv_bias_transpose_rope(void *q, void *k, void *v, const void *qkv, const void *bias,
                                  const int32_t *padding_offset, const int32_t *history_len,
                                  int batch, int seq_len, int num_tokens,
                                  int head_num, int kv_head_num, int head_size,
                                  int rotary_dim, float rotary_base,
                                  llmie_dtype dtype, llmie_stream stream);

/* replaces launchRope                  src/kernels/rope.cu:60-98
 * in place on qkv[bs, nh+2kvh, hs]; position = step-1.  If step_dev != NULL the position is
 * read from that device int (graph replay with a moving step) and `step` is ignored. */
int llmie_rope_decode(void *qkv, int batch, int head_num, int kv_head_num, int head_size,
                      int step, const int32_t *step_dev, int rotary_dim, float rotary_base,
                      llmie_dtype dtype, llmie_stream stream);

/* replaces launchDecoderMaskedMultiHeadAttention   src/kernels/decoder_self_attention.cu:211-270
 * qkv[bs, nh+2kvh, hs] (+bias) ; caches [L,bs,kvh,max_seq,hs]; out [bs, nh*hs].
 * Appends k,v at slot step-1 of `layer`, then attends over t < step.
 * workspace: llmie_decoder_mha_workspace_bytes() bytes (split-KV partials). */
size_t llmie_decoder_mha_workspace_bytes(int batch, int head_num, int head_size, int max_seq_len);
int llmie_decoder_mha(const void *qkv, const void *qkv_bias, void *k_cache, void *v_cache,
                      void *out, int layer, int batch, int head_num, int kv_head_num,
                      int head_size, int max_seq_len, int step, const int32_t *step_dev,
                      void *workspace, size_t workspace_bytes,
                      llmie_dtype dtype, llmie_stream stream);

/* Fused form used by the decoder engine: launchRope + launchDecoderMaskedMultiHeadAttention in one launch
 * (src/layers/self_attention.cpp:100-118).  rope_table: [max_pos][head_size/2] pairs (cos, sin) of
 * pos / base^(2j/rot_dim) (NULL = q,k already rotated); q and the new k are rotated in-kernel before the
 * bias add, exactly the reference's order.  tickets: [batch, kv_head_num] int32 arrival counters that must be
 * ZERO before the first call (the kernel re-arms them): the last workgroup of each (batch, kv head) merges the
 * split partials in the same launch (no merge kernel).  NULL tickets = separate merge kernel.
 * Supported: head_size in {32,64,128,256}, head_num/kv_head_num in {1,2,4,8}; else LLMIE_ERR_UNSUPPORTED. */
int llmie_decoder_mha_rope(const void *qkv, const void *qkv_bias, void *k_cache, void *v_cache,
                           void *out, int layer, int batch, int head_num, int kv_head_num,
                           int head_size, int max_seq_len, int step, const int32_t *step_dev,
                           void *workspace, size_t workspace_bytes, const void *rope_table,
                           int rotary_dim, int32_t *tickets, llmie_dtype dtype, llmie_stream stream);

/* Ragged batch (continuous batching; no reference counterpart: the reference's decode step shares ONE `step` across the batch,
 * decoder_self_attention.cu:211-270 / self_decoder.cpp:69): ctx_len_dev[b] = tokens of sequence b INCLUDING this step's; the RoPE
 * position, the append slot and the attention span are per sequence.  A value outside [1, max_seq_len] leaves that sequence's
 * cache and output untouched.  block_table (nullable) = paged cache, see llmie_decoder_forward_paged.  Row b equals the row a
 * batch-1 call at step ctx_len[b] produces, bit for bit (same kernels, same chunking). */
int llmie_decoder_mha_ragged(const void *qkv, const void *qkv_bias, void *k_cache, void *v_cache, void *out, int layer,
                             int batch, int head_num, int kv_head_num, int head_size, int max_seq_len,
                             const int32_t *ctx_len_dev, void *workspace, size_t workspace_bytes, const void *rope_table,
                             int rotary_dim, const int32_t *block_table, int max_pages, int num_pages,
                             llmie_dtype dtype, llmie_stream stream);

/* replaces launchConcatKVCache         src/kernels/concat_past_kv.cu:44-89  (one call = K or V) */
int llmie_concat_kv(const void *src, void *cache, const int32_t *cur_len,
                    const int32_t *history_len, int layer, int batch, int kv_head_num,
                    int max_q_len, int max_seq_len, int head_size,
                    llmie_dtype dtype, llmie_stream stream);

/* replaces launchRepeatKVCache         src/kernels/repeat_kv.cu:51-106      (one call = K or V) */
int llmie_repeat_kv(const void *cache, void *dst, const int32_t *ctx_len, int layer, int batch,
                    int head_num, int kv_head_num, int max_k_len, int max_seq_len, int head_size,
                    llmie_dtype dtype, llmie_stream stream);

/* replaces launchFusedScaleMaskAndSoftmax  src/kernels/scale_and_mask_and_softmax.cu:213-341
 * out may alias qk */
int llmie_scale_mask_softmax(const void *qk, const void *mask, void *out, float scale,
                             int batch, int head_num, int q_len, int k_len,
                             llmie_dtype dtype, llmie_stream stream);

/* replaces launchFusedTransposeAndRemovePadding  src/kernels/transpose_and_remove_padding.cu:45-74 */
int llmie_transpose_remove_padding(const void *src, void *dst, const int32_t *padding_offset,
                                   int num_tokens, int batch, int seq_len, int head_num,
                                   int head_size, llmie_dtype dtype, llmie_stream stream);

/* replaces launchSiluAndMul            src/kernels/silu_and_mul.cu:61-82   in[T,2,I] -> out[T,I] */
int llmie_silu_and_mul(const void *in, void *out, int num_tokens, int inter,
                       llmie_dtype dtype, llmie_stream stream);

/* replaces launchTopKForBeamSearch     src/kernels/topk.cu:104-140
 * probs[rows,vocab] -> ids/vals[rows,K] descending (ties: lower id first).
 * tmp_ids/tmp_vals: [rows, blocks_per_row, K] scratch (the reference's round-1 buffers);
 * 1 <= K <= 32, 1 <= blocks_per_row <= 64. */
int llmie_topk(const void *probs, int32_t *tmp_ids, void *tmp_vals, int32_t *ids, void *vals,
               int rows, int vocab, int K, int blocks_per_row,
               llmie_dtype dtype, llmie_stream stream);

/* replaces launchSampling              src/kernels/sampling.cu:73-102
 * topk_val is NOT modified (the reference overwrites it with the exponentials).
 * finished is a byte per sequence (C++ bool).  Uniform draw: Philox4x32-10(seed=step, stream=b).
 * If step_dev != NULL the seed is read from that device int. */
int llmie_sampling(const int32_t *topk_id, const void *topk_val, int32_t *seq_len,
                   uint8_t *finished, int32_t *out_id, int batch, int K, int step,
                   const int32_t *step_dev, int end_id, int vocab,
                   llmie_dtype dtype, llmie_stream stream);

/* ------------------------------------------------------------------------- */
/* 2. weight-only quantised / fp8 linears (reference: planned only,           */
/*    README.md:36-39, linear.cuh:12 TODO)                                    */
/* ------------------------------------------------------------------------- */

/* y[M,N] = x[M,K] . (scale[n]*Wq[n,k])^T ; x,y,scale,bias fp16; Wq int8 row-major [N,K].
 * workspace: llmie_linear_workspace_bytes(LLMIE_W_INT8 / LLMIE_W_INT4, M, K, N) bytes of caller-owned split-K slabs (see
 * llmie_linear); without it int8 serves M <= 64 and int4 the GEMV sizes only. */
int llmie_linear_w8a16(const void *x, const int8_t *wq, const void *scale, void *y,
                       int M, int K, int N, const void *bias, const void *residual,
                       void *workspace, size_t workspace_bytes, llmie_stream stream);
/* int4: two nibbles per byte (low nibble = even k), value = nibble-8, scale[n, k/group] fp16 */
int llmie_linear_w4a16(const void *x, const uint8_t *wq, const void *scale, void *y,
                       int M, int K, int N, int group, const void *bias, const void *residual,
                       void *workspace, size_t workspace_bytes, llmie_stream stream);
/* fp8 e4m3 (OCP) weights [N,K] with per-row fp32 scale; x fp16 is quantised per token to e4m3
 * on the fly (scale = amax/448); y[m,n] = w_scale[n] * x_scale[m] * sum_k wq[n,k] xq[m,k], fp32 accumulate; y fp16.
 * M <= 8: K-split GEMV (same arithmetic on the VALU); 8 < M: split-K fp8 MFMA (K % 256 == 0, K >= 512; 64 < M <= 128 rows per
 * pass take the 128-row LDS-DMA form);
 * prefill-sized M x N (>= 192 tiles of 256 x 256 or 256 x 128): tiled v_mfma_scale_f32_16x16x128_f8f6f4 GEMM
 * (K % 128 == 0).  workspace = quantised activations + per-token scales + the fp32 slabs of the split-K form, all caller-owned:
 * llmie_linear_fp8_workspace_bytes(M, K, N) bytes, 256-byte aligned. */
int llmie_linear_fp8(const void *x, const uint8_t *w_fp8, const float *w_scale, void *y,
                     int M, int K, int N, const void *bias, const void *residual,
                     void *workspace, size_t workspace_bytes, llmie_stream stream);
size_t llmie_linear_fp8_workspace_bytes(int M, int K, int N);   /* N = 0: the activation part only (llmie_linear_fp8_swiglu) */
/* y[M, two_inter/2] = silu(gate) * up of the fused gate_up projection in fp8 (ffn.cpp:105-122 in one launch after the
 * activation quantisation); prefill-sized shapes only (LLMIE_ERR_UNSUPPORTED otherwise: use llmie_linear_fp8 +
 * llmie_silu_and_mul) */
int llmie_linear_fp8_swiglu(const void *x, const uint8_t *w_fp8, const float *w_scale, void *y, int M, int K,
                            int two_inter, void *workspace, size_t workspace_bytes, llmie_stream stream);
/* offline quantisers (device side): w fp16 [N,K] -> int8/int4/fp8 + scales */
int llmie_quantize_w8(const void *w, int8_t *wq, void *scale, int N, int K, llmie_stream stream);
int llmie_quantize_w4(const void *w, uint8_t *wq, void *scale, int N, int K, int group,
                      llmie_stream stream);
int llmie_quantize_fp8(const void *w, uint8_t *wq, float *scale, int N, int K, llmie_stream stream);

/* Tile-packed weight images for decode batches (no reference counterpart: the reference streams row-major weights through
 * cuBLAS, src/kernels/linear.cu:10-87).  An MFMA operand wants 16 different weight rows across the lanes of one load, a
 * DRAM stream wants one contiguous KiB per wave instruction; the packed image gives both: tile = 16 rows, block = KB
 * consecutive k of the tile = 1 KiB in MFMA fragment order (KB = 32 fp16, 64 int8 / e4m3, 128 int4); K % KB == 0, rows
 * past N are zero.  swiglu_pairs != 0: w is a fused gate_up matrix [2I, K], packed tile 2p = gate rows [16p, 16p+16),
 * tile 2p+1 = the matching up rows.  int4: the group-128 scales are re-laid [tile][block][16] fp16 into packed_scale
 * (llmie_packed_scale_bytes; 0 for the other formats, which keep their per-row scale vector as it is).
 * llmie_linear_packed: y[M, N] (or [M, N/2] for swiglu != 0) = [swiglu]( rmsnorm(x + pre_bias) * gamma . W^T ) (+ residual),
 * 1 <= M <= 32 fp16 rows; gamma == NULL: no norm.  x is read once per workgroup into registers (8 waves split K), the
 * weights stream through a per-wave LDS-DMA ring.  workspace: llmie_linear_packed_workspace_bytes() bytes (non-zero only where K
 * exceeds the register-resident slice, e.g. the 7B down projection: fp32 split-K slabs + one reduce launch).
 */
size_t llmie_packed_weight_bytes(llmie_weight_format fmt, int N, int K, int swiglu_pairs);
size_t llmie_packed_scale_bytes(llmie_weight_format fmt, int N, int K, int swiglu_pairs);
int llmie_pack_weight(llmie_weight_format fmt, const void *w, const void *scale, void *packed, void *packed_scale,
                      int N, int K, int swiglu_pairs, llmie_stream stream);
size_t llmie_linear_packed_workspace_bytes(llmie_weight_format fmt, int M, int K, int N);
/* x32_flags: which operands are in the fragment-ordered activation layout "x32" instead of row-major (LLMIE_X32_X = x,
 * LLMIE_X32_Y = y, LLMIE_X32_RES = residual): a [<= 32 rows, C] fp16 matrix, C % 32 == 0, stored
 * [C / 32][2 row tiles][64 lanes][8 halves], lane = 16 * ((c % 32) / 8) + (m % 16), i.e. element (m, c) at half offset
 * ((c / 32) * 2 + m / 16) * 512 + ((c % 32) / 8) * 128 + (m % 16) * 8 + c % 8; always 32 rows of storage
 * (llmie_x32_bytes(C) = 64 C bytes).  Each (32 columns, 16 rows) piece is the 1 KiB MFMA operand of the consumer, so a
 * kernel chain that keeps its activations in x32 loads them with contiguous KiB reads and no transposition;
 * llmie_x32_convert moves a matrix between the two layouts (to_x32 != 0: rows >= M are zero-filled). */
enum { LLMIE_X32_X = 1, LLMIE_X32_Y = 2, LLMIE_X32_RES = 4 };
size_t llmie_x32_bytes(int C);
int llmie_x32_convert(const void *src, void *dst, int M, int C, int to_x32, llmie_stream stream);
int llmie_linear_packed(llmie_weight_format fmt, const void *x, const void *packed, const void *scale, void *y,
                        int M, int K, int N, int swiglu, int x32_flags, const void *residual, const void *gamma,
                        const void *pre_bias, float eps, void *workspace, size_t workspace_bytes,
                        llmie_stream stream);

/* ------------------------------------------------------------------------- */
/* 3. fused decoder engine (what LlamaSelfDecoder<T>::forward and             */
/*    LlamaModel<T>::generateNextToken run on; src/layers/self_decoder.cpp:24-122, */
/*    src/models/llama/llama.cpp:219-318)                                     */
/* ------------------------------------------------------------------------- */

typedef enum { LLMIE_KV_NATIVE = 0, LLMIE_KV_FP8 = 1 } llmie_kv_format;

typedef struct {
    const void *data;      /* [N,K] in the format's storage */
    const void *scale;     /* NULL for F16/F32 */
    const void *bias;      /* nullable, [N], activation dtype */
} llmie_matrix;

typedef struct {
    const void *attn_norm_gamma;   /* [H] */
    llmie_matrix qkv;              /* N=(nh+2kvh)*hs, K=H   (self_attn.qkv, layer_weights.cpp:28-31) */
    llmie_matrix o;                /* N=H, K=H                                                     */
    const void *ffn_norm_gamma;    /* [H] */
    llmie_matrix gate_up;          /* N=2I, K=H  rows [0,I)=gate, [I,2I)=up (layer_weights.cpp:41-44) */
    llmie_matrix down;             /* N=H, K=I */
} llmie_layer_weights;

typedef struct {
    int head_num, kv_head_num, head_size, inter_size, num_layers, vocab_size;
    int max_seq_len, max_batch;
    int rotary_dim;
    float rotary_base, rms_eps;
    llmie_dtype dtype;             /* activation / KV dtype */
    llmie_weight_format wfmt;      /* storage of the 4 big matrices per layer */
    int int4_group;                /* group size for LLMIE_W_INT4 */
    /* KV-cache storage (the reference stores T): LLMIE_KV_NATIVE = dtype, LLMIE_KV_FP8 = e4m3 bytes with one
     * static scale per cache, stored = e4m3(x / scale), same [L, batch, kvh, max_seq, hs] indexing (fp16 engines,
     * head_size 64/128 decode, 128 prefill, head_num/kv_head_num in {1,2,4}); a scale <= 0 means 1. */
    llmie_kv_format kv_fmt;
    float k_scale, v_scale;
    /* ABI 3: LLMIE_DEC_* bits (0 = the round-2 behaviour) */
    int flags;
} llmie_decoder_config;

/* Weight residency (ABI 3).  An engine whose max_batch lies above the GEMV range (fp16 5, int8 / fp8 2, int4 2 rows) builds
 * tile-packed images of its four matrices per layer inside its workspace at create time -- a SNAPSHOT: weights updated in place
 * afterwards are seen by the batch <= GEMV-range and prefill paths, not by the packed one; create synchronises the device before
 * it packs, so uploads on any stream have landed.  By default that image is a second copy next to the caller's row-major matrices.
 *   LLMIE_DEC_NO_PACKED_COPY  no image is built: batches 4..32 take the split-K batch path on the row-major weights (slower there,
 *                             nothing doubled).
 *   LLMIE_DEC_PACKED_ONLY     the images are the ONLY weights the engine reads after create: the `data` arrays of the four
 *                             matrices of every layer may be freed or reused once llmie_decoder_create has returned (scales,
 *                             biases and norm gammas stay referenced).  Needs max_batch <= 32 (fp8: 16) and shapes the packed
 *                             kernels take; every decode batch runs on the packed kernels, prefill (fp16 / int8 / int4) unpacks
 *                             one matrix at a time into its workspace.  An int8 Llama-2-7B decoder is then resident at ~6.5 GB
 *                             instead of ~13 GB.
 * llmie_decoder_resident_weight_bytes: bytes of layer-matrix storage (row-major matrices that must stay + images) of a config. */
/* llmie_decoder_repack (ABI 3): the weights were updated in place, or -- for a packed-only engine -- are to be replaced: rebuild
 * the images from `layers` (row-major matrices in the engine's format; their scale / bias / gamma pointers replace the ones given
 * at create).  The pack kernels are enqueued on `stream`. */
#define LLMIE_DEC_NO_PACKED_COPY 1
#define LLMIE_DEC_PACKED_ONLY 2
size_t llmie_decoder_resident_weight_bytes(const llmie_decoder_config *cfg);

typedef struct llmie_decoder llmie_decoder; /* opaque */

/* Workspace is caller-owned device memory (no allocation inside). */
size_t llmie_decoder_workspace_bytes(const llmie_decoder_config *cfg);
/* layers[num_layers] is copied (pointers only). Returns NULL on invalid config. */
llmie_decoder *llmie_decoder_create(const llmie_decoder_config *cfg,
                                    const llmie_layer_weights *layers,
                                    void *workspace, size_t workspace_bytes);
void llmie_decoder_destroy(llmie_decoder *dec);
int llmie_decoder_repack(llmie_decoder *dec, const llmie_layer_weights *layers, llmie_stream stream);   /* see LLMIE_DEC_* above */

/* One decode step through all layers, in place semantics of LlamaSelfDecoder::forward:
 * hidden_in[bs,H] -> hidden_out[bs,H] (may alias).  Caches [L, batch, kvh, max_seq, hs].
 * step = context length INCLUDING the new token (reference `step`); if step_dev != NULL it is
 * read on the device (one int shared by the batch) so a captured graph can be replayed. */
int llmie_decoder_forward(llmie_decoder *dec, const void *hidden_in, void *hidden_out,
                          void *k_cache, void *v_cache, int batch, int step,
                          const int32_t *step_dev, llmie_stream stream);

/* Paged KV cache (the reference only has the dense slab): K and V live in pools of 128-token pages
 * [L, num_pages, kvh, LLMIE_KV_PAGE_TOKENS, hs] (element type = the engine's kv_fmt) and block_table[b * max_pages + p]
 * (device int32) names the pool page holding tokens [128 p, 128 p + 128) of sequence b.  Same arithmetic as
 * llmie_decoder_forward (bit-identical outputs for the same cache contents); fused decode paths only.
 * llmie_decoder_prefill_paged is llmie_decoder_prefill writing / reading the pages directly.
 * llmie_kv_pages_copy moves the first ctx_len[b] tokens of every sequence between a dense cache
 * [L, batch, kvh, max_seq, hs] and the pools (to_pages != 0: dense -> pages, e.g. after llmie_decoder_prefill). */
#define LLMIE_KV_PAGE_TOKENS 128
int llmie_decoder_forward_paged(llmie_decoder *dec, const void *hidden_in, void *hidden_out, void *k_pool, void *v_pool,
                                const int32_t *block_table, int max_pages, int num_pages, int batch, int step,
                                const int32_t *step_dev, llmie_stream stream);
/* Ragged batch (continuous batching; the reference steps a whole batch at ONE position, self_decoder.cpp:69-119): sequence b
 * is at context length ctx_len_dev[b] (device int32 [batch], including this step's token).  Only the attention launch differs
 * (RoPE position, append slot, span per sequence); row b is what llmie_decoder_forward at step ctx_len[b] gives that row. */
int llmie_decoder_forward_ragged(llmie_decoder *dec, const void *hidden_in, void *hidden_out, void *k_cache, void *v_cache,
                                 int batch, const int32_t *ctx_len_dev, llmie_stream stream);
int llmie_decoder_forward_paged_ragged(llmie_decoder *dec, const void *hidden_in, void *hidden_out, void *k_pool, void *v_pool,
                                       const int32_t *block_table, int max_pages, int num_pages, int batch,
                                       const int32_t *ctx_len_dev, llmie_stream stream);
int llmie_decoder_prefill_paged(llmie_decoder *dec, const void *hidden_in, void *hidden_out, void *k_pool, void *v_pool,
                                const int32_t *block_table, int max_pages, int num_pages, const int32_t *input_lengths,
                                const int32_t *history_lengths, int batch, int num_tokens, int max_q_len, void *workspace,
                                size_t workspace_bytes, llmie_stream stream);
int llmie_kv_pages_copy(void *dense, void *pool, const int32_t *block_table, const int32_t *ctx_len, int to_pages,
                        int layers, int batch, int kv_head_num, int max_seq_len, int head_size, int max_pages,
                        int num_pages, int elem_bytes, llmie_stream stream);

/* Prefill through all layers = LlamaContextDecoder<T>::forward (src/layers/context_decoder.cpp:58-199,
 * context_attention.cpp:143-312) on PACKED tokens: hidden_in/out [num_tokens, H] hold the sequences back to back
 * (input_lengths[b] tokens each, device int32), history_lengths[b] tokens of each sequence are already in the caches;
 * k/v of the new tokens are appended at history+pos.  Attention is a flash kernel: causal mask from the lengths,
 * no padding buffers, no [bs,nh,q,k] score matrix.  fp16 engines with fp16 or fp8 weights (fp8: every projection
 * input is quantised per token, llmie_linear_fp8 semantics) and head_size 128.
 * Caches are [L, batch, kvh, max_seq, hs] with the batch of THIS call.  workspace: caller-owned scratch. */
size_t llmie_decoder_prefill_workspace_bytes(const llmie_decoder_config *cfg, int max_tokens, int max_batch);
int llmie_decoder_prefill(llmie_decoder *dec, const void *hidden_in, void *hidden_out, void *k_cache, void *v_cache,
                          const int32_t *input_lengths, const int32_t *history_lengths, int batch, int num_tokens,
                          int max_q_len, void *workspace, size_t workspace_bytes, llmie_stream stream);

/* LM head + top-k + sampling tail (llama.cpp:247-318): final RMSNorm(gamma) -> logits =
 * x . lm_head[V,H]^T -> top-K -> sample.  logits[bs,V] and topk buffers caller-owned.
 * `hidden` is CLOBBERED and its content afterwards is unspecified: the fused-norm GEMV form (fp16, small batches) leaves it
 * as it was, every other form RMS-normalises it in place as the reference does (llama.cpp:247) -- do not read it, and do not
 * call this twice on one buffer.  Candidates the top-k could not fill (NaN logits) are skipped by the sampler; a row without
 * any valid candidate emits end_id and sets finished.  In fp16 the sampler keeps exp(v - max) in fp32 where sampling.cu:31
 * rounds it to T first: the chosen token can differ at a bin edge (parity unpinned there, the cuRAND stream is not reproducible
 * either). */
int llmie_lm_head_sample(llmie_decoder *dec, void *hidden /* [bs,H]; clobbered (may be normalised in place) */,
                         const void *final_norm_gamma, const llmie_matrix *lm_head,
                         llmie_weight_format lm_fmt, void *logits,
                         int32_t *tmp_ids, void *tmp_vals, int32_t *topk_ids, void *topk_vals,
                         int K, int blocks_per_row, int32_t *seq_len, uint8_t *finished,
                         int32_t *out_ids, int batch, int step, const int32_t *step_dev,
                         int end_id, llmie_stream stream);

/* ABI 3.  llmie_lm_head_sample with the tail of the step fused into ONE launch behind round 1 of the top-k: round 2 + sampling
 * (bit-identical ids / values / picks / seq_len / finished) + -- if next_hidden != NULL -- the next step's input embedding
 * (next_hidden[b, :] = embed_table[out_ids[b], :], llmie_input_embedding's rule for ids outside the table; llama.cpp:219 of the
 * next token) + -- if advance_step != 0 -- *step_dev += 1 once every row has read it (llmie_advance_step).  Replaces four
 * launches of the batch-1 step (top-k round 2, sampling, advance_step, input_embedding). */
int llmie_lm_head_sample_next(llmie_decoder *dec, void *hidden, const void *final_norm_gamma, const llmie_matrix *lm_head,
                              llmie_weight_format lm_fmt, void *logits, int32_t *tmp_ids, void *tmp_vals, int32_t *topk_ids,
                              void *topk_vals, int K, int blocks_per_row, int32_t *seq_len, uint8_t *finished, int32_t *out_ids,
                              int batch, int step, int32_t *step_dev, int end_id, const void *embed_table, void *next_hidden,
                              int advance_step, llmie_stream stream);

/* Per-kernel timing of the engine (eager launches only, never inside graph capture): between
 * profile_begin and profile_end every kernel the engine launches is bracketed by hipEvents
 * recorded on the launch stream.  profile_end synchronises the stream (the one entry point that
 * does) and returns, per op kind, the summed device time in ms and the number of launches. */
enum {
    LLMIE_OP_ATTN_NORM = 0, LLMIE_OP_QKV_GEMM, LLMIE_OP_ROPE, LLMIE_OP_MHA, LLMIE_OP_O_GEMM,
    LLMIE_OP_FFN_NORM, LLMIE_OP_GATE_UP_SWIGLU, LLMIE_OP_DOWN_GEMM, LLMIE_OP_FINAL_NORM,
    LLMIE_OP_LM_HEAD, LLMIE_OP_TOPK, LLMIE_OP_SAMPLING,
    LLMIE_OP_CHAIN, /* ABI 3: one persistent launch = O -> gate/up -> down -> next QKV of the packed batch-decode path */
    LLMIE_OP_COUNT /* prefill re-uses the layer op kinds */
};
int llmie_decoder_profile_begin(llmie_decoder *dec, int max_events);
int llmie_decoder_profile_end(llmie_decoder *dec, llmie_stream stream, double *ms_by_op /*[LLMIE_OP_COUNT]*/,
                              int *launches_by_op /*[LLMIE_OP_COUNT]*/);

/* ABI 3.  The persistent chain launches of the batch-decode path (4 < batch <= 32) synchronise their workgroups with in-kernel
 * grid barriers whose spins are bounded: a barrier that expires (a workgroup was not resident) sets a device-side error word and
 * the launch returns early instead of hanging.  This call synchronises `stream`, reads the word and returns LLMIE_ERR_LAUNCH with
 * a message if it is set (and clears it); LLMIE_OK otherwise.  Not on the compute path: for tests and health checks. */
int llmie_decoder_status(llmie_decoder *dec, llmie_stream stream);
/* ABI 3, diagnostic: arm (device pointer to 256 x 16 uint64) or disarm (NULL) the phase-edge timestamps of the chain launches:
 * per workgroup the s_memrealtime (100 MHz) values at kernel start, then before / behind every grid barrier, then at the end. */
int llmie_decoder_debug_stamps(llmie_decoder *dec, void *stamps_dev);

/* device-side helper for graph replay: *step_dev += 1 */
int llmie_advance_step(int32_t *step_dev, llmie_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* LLMIE_H */

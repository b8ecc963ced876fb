/*
 * llmie_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see llmie_oracle.h).
 *
 * Plain C restatement of the reference algorithms for the Llama-2 decoder hot
 * path.  Written from the semantics of the reference kernels / unit-test
 * checkers cited in llmie_oracle.h; no reference source is copied.
 * Build: make -C oracle   (gcc -O3 -fopenmp -> oracle/libllmie_oracle.so)
 */
#include "llmie_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void orc_set_num_threads(int n) {
    if (n > 0) omp_set_num_threads(n);
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_input_embedding(const int32_t *ids, const float *table, float *out,
                         int num_tokens, int hidden) {
    for (int t = 0; t < num_tokens; ++t) {
        memcpy(out + (size_t)t * hidden, table + (size_t)ids[t] * hidden,
               sizeof(float) * (size_t)hidden);
    }
}

void orc_cal_padding_offset(int32_t *padding_offset, int32_t *cum_seqlens,
                            const int32_t *lens, int batch, int max_q_len) {
    int total = 0, pad = 0, w = 0;
    for (int b = 0; b < batch; ++b) {
        cum_seqlens[b] = total;
        for (int j = 0; j < lens[b]; ++j) padding_offset[w++] = pad;
        pad += max_q_len - lens[b];
        total += lens[b];
    }
    cum_seqlens[batch] = total;
}

void orc_build_causal_mask(float *mask, const int32_t *q_lens, const int32_t *k_lens,
                           int batch, int max_q_len, int max_k_len) {
    for (int b = 0; b < batch; ++b) {
        const int ql = q_lens[b], kl = k_lens[b];
        float *m = mask + (size_t)b * max_q_len * max_k_len;
        for (int q = 0; q < max_q_len; ++q)
            for (int k = 0; k < max_k_len; ++k)
                m[(size_t)q * max_k_len + k] =
                    (q < ql && k < kl && k <= q + (kl - ql)) ? 1.0f : 0.0f;
    }
}

void orc_rmsnorm(float *x, float *resid, const float *gamma, float eps,
                 int num_tokens, int hidden) {
    for (int t = 0; t < num_tokens; ++t) {
        float *row = x + (size_t)t * hidden;
        float ss = 0.0f;
        for (int i = 0; i < hidden; ++i) {
            if (resid) resid[(size_t)t * hidden + i] = row[i];
            ss += row[i] * row[i];
        }
        const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
        for (int i = 0; i < hidden; ++i) row[i] = row[i] * gamma[i] * inv;
    }
}

void orc_fused_add_bias_residual_rmsnorm(float *resid, float *out, const float *bias,
                                         const float *gamma, float eps,
                                         int num_tokens, int hidden) {
    for (int t = 0; t < num_tokens; ++t) {
        float *o = out + (size_t)t * hidden;
        float *r = resid ? resid + (size_t)t * hidden : NULL;
        float ss = 0.0f;
        for (int i = 0; i < hidden; ++i) {
            float v = o[i];
            if (r) { v += r[i]; r[i] = v; }
            if (bias) v += bias[i];
            o[i] = v;
            ss += v * v;
        }
        const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
        if (gamma)
            for (int i = 0; i < hidden; ++i) o[i] = gamma[i] * o[i] * inv;
    }
}

void orc_add_residual(const float *resid, float *out, int num_tokens, int hidden) {
    const size_t n = (size_t)num_tokens * hidden;
    for (size_t i = 0; i < n; ++i) out[i] += resid[i];
}

void orc_linear(const float *x, const float *w, float *y, int M, int K, int N, int trans_b) {
    if (trans_b) {
#pragma omp parallel for schedule(static)
        for (int n = 0; n < N; ++n) {
            const float *wr = w + (size_t)n * K;
            for (int m = 0; m < M; ++m) {
                const float *xr = x + (size_t)m * K;
                float acc = 0.0f;
                for (int k = 0; k < K; ++k) acc += xr[k] * wr[k];
                y[(size_t)m * N + n] = acc;
            }
        }
    } else {
#pragma omp parallel for schedule(static)
        for (int m = 0; m < M; ++m) {
            float *yr = y + (size_t)m * N;
            for (int n = 0; n < N; ++n) yr[n] = 0.0f;
            for (int k = 0; k < K; ++k) {
                const float xv = x[(size_t)m * K + k];
                const float *wr = w + (size_t)k * N;
                for (int n = 0; n < N; ++n) yr[n] += xv * wr[n];
            }
        }
    }
}

void orc_batched_gemm(const float *a, const float *b, float *c,
                      int batch, int m, int n, int k, int trans_b) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < batch; ++i) {
        orc_linear(a + (size_t)i * m * k, b + (size_t)i * n * k, c + (size_t)i * m * n,
                   m, k, n, trans_b);
    }
}

static void rope_pair(float x0, float x1, int d, int rotary_dim, float base, float pos,
                      float *o0, float *o1) {
    /* includes/rope_utils.cuh:6-19: angle = pos / base^(2d/rot_dim) */
    const float ang = pos / powf(base, (float)(2 * d) / (float)rotary_dim);
    const float c = cosf(ang), s = sinf(ang);
    *o0 = x0 * c - x1 * s;
    *o1 = x1 * c + x0 * s;
}

void orc_qkv_bias_transpose_rope(float *q, float *k, float *v, const float *qkv,
                                 const float *bias, const int32_t *padding_offset,
                                 const int32_t *history_len,
                                 int batch, int seq_len, int num_tokens,
                                 int head_num, int kv_head_num, int head_size,
                                 int rotary_dim, float rotary_base) {
    (void)batch;
    const int qkv_heads = head_num + 2 * kv_head_num;
    const int half = head_size / 2;
    for (int t = 0; t < num_tokens; ++t) {
        const int dst_tok = t + padding_offset[t];
        const int b = dst_tok / seq_len;
        const int s = dst_tok % seq_len;
        const float pos = (float)(history_len[b] + s);
        const float *row = qkv + (size_t)t * qkv_heads * head_size;
        for (int h = 0; h < qkv_heads; ++h) {
            const float *src = row + (size_t)h * head_size;
            const float *bs = bias ? bias + (size_t)h * head_size : NULL;
            float *dst;
            int rotate = 1;
            if (h < head_num) {
                dst = q + (((size_t)b * head_num + h) * seq_len + s) * head_size;
            } else if (h < head_num + kv_head_num) {
                dst = k + (((size_t)b * kv_head_num + (h - head_num)) * seq_len + s) * head_size;
            } else {
                dst = v + (((size_t)b * kv_head_num + (h - head_num - kv_head_num)) * seq_len + s) * head_size;
                rotate = 0;
            }
            for (int d = 0; d < half; ++d) {
                float x0 = src[d] + (bs ? bs[d] : 0.0f);
                float x1 = src[d + half] + (bs ? bs[d + half] : 0.0f);
                if (rotate && d < rotary_dim / 2) {
                    rope_pair(x0, x1, d, rotary_dim, rotary_base, pos, &dst[d], &dst[d + half]);
                } else {
                    dst[d] = x0;
                    dst[d + half] = x1;
                }
            }
        }
    }
}

void orc_rope_decode(float *qkv, int batch, int head_num, int kv_head_num, int head_size,
                     int step, int rotary_dim, float rotary_base) {
    const int qkv_heads = head_num + 2 * kv_head_num;
    const int half = head_size / 2;
    const float pos = (float)(step - 1);
    for (int b = 0; b < batch; ++b)
        for (int h = 0; h < head_num + kv_head_num; ++h) {
            float *x = qkv + ((size_t)b * qkv_heads + h) * head_size;
            for (int d = 0; d < half && d < rotary_dim / 2; ++d)
                rope_pair(x[d], x[d + half], d, rotary_dim, rotary_base, pos, &x[d], &x[d + half]);
        }
}

void orc_decoder_mha(const float *qkv, const float *qkv_bias,
                     float *k_cache, float *v_cache, float *out,
                     int layer, int batch, int head_num, int kv_head_num, int head_size,
                     int max_seq_len, int step) {
    const int qkv_heads = head_num + 2 * kv_head_num;
    const int rep = head_num / kv_head_num;
    const float scale = 1.0f / sqrtf((float)head_size);
    const size_t layer_off = (size_t)layer * batch * kv_head_num * max_seq_len * head_size;
    /* 1. append k,v of this step (once per kv head) */
    for (int b = 0; b < batch; ++b)
        for (int g = 0; g < kv_head_num; ++g) {
            const float *ks = qkv + ((size_t)b * qkv_heads + head_num + g) * head_size;
            const float *vs = qkv + ((size_t)b * qkv_heads + head_num + kv_head_num + g) * head_size;
            const size_t slot = layer_off +
                (((size_t)b * kv_head_num + g) * max_seq_len + (step - 1)) * head_size;
            for (int d = 0; d < head_size; ++d) {
                k_cache[slot + d] = ks[d] + (qkv_bias ? qkv_bias[(size_t)(head_num + g) * head_size + d] : 0.0f);
                v_cache[slot + d] = vs[d] + (qkv_bias ? qkv_bias[(size_t)(head_num + kv_head_num + g) * head_size + d] : 0.0f);
            }
        }
    /* 2. attention over t < step */
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < batch; ++b)
        for (int h = 0; h < head_num; ++h) {
            const int g = h / rep;
            const float *qs = qkv + ((size_t)b * qkv_heads + h) * head_size;
            const float *K = k_cache + layer_off + ((size_t)b * kv_head_num + g) * max_seq_len * head_size;
            const float *V = v_cache + layer_off + ((size_t)b * kv_head_num + g) * max_seq_len * head_size;
            float *logits = (float *)malloc(sizeof(float) * (size_t)step);
            float *qv = (float *)malloc(sizeof(float) * (size_t)head_size);
            for (int d = 0; d < head_size; ++d)
                qv[d] = qs[d] + (qkv_bias ? qkv_bias[(size_t)h * head_size + d] : 0.0f);
            float mx = -INFINITY;
            for (int t = 0; t < step; ++t) {
                float acc = 0.0f;
                for (int d = 0; d < head_size; ++d) acc += qv[d] * K[(size_t)t * head_size + d];
                logits[t] = acc * scale;
                if (logits[t] > mx) mx = logits[t];
            }
            float sum = 0.0f;
            for (int t = 0; t < step; ++t) { logits[t] = expf(logits[t] - mx); sum += logits[t]; }
            sum += 1e-6f;
            float *o = out + ((size_t)b * head_num + h) * head_size;
            for (int d = 0; d < head_size; ++d) o[d] = 0.0f;
            for (int t = 0; t < step; ++t) {
                const float p = logits[t] / sum;
                for (int d = 0; d < head_size; ++d) o[d] += p * V[(size_t)t * head_size + d];
            }
            free(logits);
            free(qv);
        }
}

void orc_concat_kv(const float *src, float *cache, const int32_t *cur_len,
                   const int32_t *history_len, int layer, int batch, int kv_head_num,
                   int max_q_len, int max_seq_len, int head_size) {
    const size_t layer_off = (size_t)layer * batch * kv_head_num * max_seq_len * head_size;
    for (int b = 0; b < batch; ++b)
        for (int h = 0; h < kv_head_num; ++h)
            for (int t = 0; t < cur_len[b] && t < max_q_len; ++t) {
                const float *s = src + (((size_t)b * kv_head_num + h) * max_q_len + t) * head_size;
                float *d = cache + layer_off +
                    (((size_t)b * kv_head_num + h) * max_seq_len + history_len[b] + t) * head_size;
                memcpy(d, s, sizeof(float) * (size_t)head_size);
            }
}

void orc_repeat_kv(const float *cache, float *dst, const int32_t *ctx_len,
                   int layer, int batch, int head_num, int kv_head_num,
                   int max_k_len, int max_seq_len, int head_size) {
    const int rep = head_num / kv_head_num;
    const size_t layer_off = (size_t)layer * batch * kv_head_num * max_seq_len * head_size;
    for (int b = 0; b < batch; ++b)
        for (int h = 0; h < head_num; ++h)
            for (int t = 0; t < ctx_len[b] && t < max_k_len; ++t) {
                const float *s = cache + layer_off +
                    (((size_t)b * kv_head_num + h / rep) * max_seq_len + t) * head_size;
                float *d = dst + (((size_t)b * head_num + h) * max_k_len + t) * head_size;
                memcpy(d, s, sizeof(float) * (size_t)head_size);
            }
}

void orc_scale_mask_softmax(const float *qk, const float *mask, float *out, float scale,
                            int batch, int head_num, int q_len, int k_len) {
    for (int b = 0; b < batch; ++b)
        for (int h = 0; h < head_num; ++h)
            for (int q = 0; q < q_len; ++q) {
                const float *row = qk + (((size_t)b * head_num + h) * q_len + q) * k_len;
                const float *mrow = mask + ((size_t)b * q_len + q) * k_len;
                float *orow = out + (((size_t)b * head_num + h) * q_len + q) * k_len;
                float mx = -INFINITY;
                for (int k = 0; k < k_len; ++k) {
                    const float v = scale * row[k] + (1.0f - mrow[k]) * (-10000.0f);
                    orow[k] = v;
                    if (v > mx) mx = v;
                }
                float sum = 0.0f;
                for (int k = 0; k < k_len; ++k) { orow[k] = expf(orow[k] - mx); sum += orow[k]; }
                const float inv = 1.0f / (sum + 1e-6f);
                for (int k = 0; k < k_len; ++k) orow[k] *= inv;
            }
}

void orc_transpose_remove_padding(const float *src, float *dst, const int32_t *padding_offset,
                                  int num_tokens, int batch, int seq_len,
                                  int head_num, int head_size) {
    (void)batch;
    for (int t = 0; t < num_tokens; ++t) {
        const int pt = t + padding_offset[t];
        const int b = pt / seq_len, s = pt % seq_len;
        for (int h = 0; h < head_num; ++h)
            memcpy(dst + ((size_t)t * head_num + h) * head_size,
                   src + (((size_t)b * head_num + h) * seq_len + s) * head_size,
                   sizeof(float) * (size_t)head_size);
    }
}

void orc_silu_and_mul(const float *in, float *out, int num_tokens, int inter) {
    for (int t = 0; t < num_tokens; ++t)
        for (int i = 0; i < inter; ++i) {
            const float g = in[((size_t)t * 2) * inter + i];
            const float u = in[((size_t)t * 2 + 1) * inter + i];
            out[(size_t)t * inter + i] = (g / (1.0f + expf(-g))) * u;
        }
}

void orc_topk(const float *probs, int32_t *ids, float *vals, int rows, int vocab, int K) {
    for (int r = 0; r < rows; ++r) {
        const float *p = probs + (size_t)r * vocab;
        int32_t *oi = ids + (size_t)r * K;
        float *ov = vals + (size_t)r * K;
        int n = 0;
        for (int i = 0; i < vocab; ++i) {
            /* insert (p[i], i) into a descending list; strict > keeps the lower id on ties */
            int pos = n;
            while (pos > 0 && p[i] > ov[pos - 1]) --pos;
            if (pos >= K) continue;
            const int last = (n < K) ? n : K - 1;
            for (int j = last; j > pos; --j) { ov[j] = ov[j - 1]; oi[j] = oi[j - 1]; }
            ov[pos] = p[i];
            oi[pos] = i;
            if (n < K) ++n;
        }
        for (int j = n; j < K; ++j) { ov[j] = -INFINITY; oi[j] = -1; }
    }
}

static uint32_t mulhilo32(uint32_t a, uint32_t b, uint32_t *hi) {
    const uint64_t p = (uint64_t)a * (uint64_t)b;
    *hi = (uint32_t)(p >> 32);
    return (uint32_t)p;
}

float orc_uniform_philox(uint32_t seed, uint32_t stream) {
    uint32_t c0 = stream, c1 = 0, c2 = 0, c3 = 0;
    uint32_t k0 = seed, k1 = 0x4c4c4d49u;
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, hi1;
        const uint32_t lo0 = mulhilo32(0xD2511F53u, c0, &hi0);
        const uint32_t lo1 = mulhilo32(0xCD9E8D57u, c2, &hi1);
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n1 = lo1;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        const uint32_t n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return (float)((c0 >> 8) + 1u) * (1.0f / 16777216.0f);
}

void orc_sampling(const int32_t *topk_id, const float *topk_val, int32_t *seq_len,
                  uint8_t *finished, int32_t *out_id,
                  int batch, int K, int step, int end_id, int vocab) {
    for (int b = 0; b < batch; ++b) {
        const int32_t *id = topk_id + (size_t)b * K;
        const float *val = topk_val + (size_t)b * K;
        float sum = 0.0f;
        for (int i = 0; i < K; ++i) sum += expf(val[i] - val[0]);
        float thr = orc_uniform_philox((uint32_t)step, (uint32_t)b) * sum;
        int chosen = id[0] % vocab;
        for (int i = 0; i < K; ++i) {
            thr -= expf(val[i] - val[0]);
            if (thr < 0.0f) { chosen = id[i] % vocab; break; }
        }
        out_id[b] = chosen;
        if (!finished[b]) ++seq_len[b];
        finished[b] = (uint8_t)(chosen == end_id);
    }
}

void orc_linear_w8(const float *x, const int8_t *wq, const float *scale, float *y,
                   int M, int K, int N) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n)
        for (int m = 0; m < M; ++m) {
            float acc = 0.0f;
            for (int k = 0; k < K; ++k)
                acc += x[(size_t)m * K + k] * (float)wq[(size_t)n * K + k];
            y[(size_t)m * N + n] = acc * scale[n];
        }
}

void orc_linear_w4(const float *x, const uint8_t *wq, const float *scale, float *y,
                   int M, int K, int N, int group) {
    const int groups = K / group;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n)
        for (int m = 0; m < M; ++m) {
            float acc = 0.0f;
            for (int g = 0; g < groups; ++g) {
                float part = 0.0f;
                for (int k = g * group; k < (g + 1) * group; ++k) {
                    const uint8_t byte = wq[((size_t)n * K + k) >> 1];
                    const int nib = (k & 1) ? (byte >> 4) : (byte & 0xF);
                    part += x[(size_t)m * K + k] * (float)(nib - 8);
                }
                acc += part * scale[(size_t)n * groups + g];
            }
            y[(size_t)m * N + n] = acc;
        }
}

void orc_self_decoder(const orc_llama_cfg *cfg, const orc_layer_weights *layers,
                      float *hidden, float *k_cache, float *v_cache,
                      int batch, int step, float *scratch) {
    const int nh = cfg->head_num, kvh = cfg->kv_head_num, hs = cfg->head_size;
    const int H = nh * hs, I = cfg->inter_size, QKV = (nh + 2 * kvh) * hs;
    float *resid = scratch;                       /* [bs,H]   */
    float *qkv = resid + (size_t)batch * H;       /* [bs,QKV] */
    float *mha = qkv + (size_t)batch * QKV;       /* [bs,H]   */
    float *gu = mha + (size_t)batch * H;          /* [bs,2,I] */
    float *act = gu + (size_t)batch * 2 * I;      /* [bs,I]   */
    for (int l = 0; l < cfg->num_layers; ++l) {
        const orc_layer_weights *w = &layers[l];
        orc_rmsnorm(hidden, resid, w->attn_norm, cfg->rms_eps, batch, H);
        orc_linear(hidden, w->qkv, qkv, batch, H, QKV, 1);
        orc_rope_decode(qkv, batch, nh, kvh, hs, step, cfg->rotary_dim, cfg->rotary_base);
        orc_decoder_mha(qkv, w->qkv_bias, k_cache, v_cache, mha, l, batch, nh, kvh, hs,
                        cfg->max_seq_len, step);
        orc_linear(mha, w->o, hidden, batch, H, H, 1);
        orc_fused_add_bias_residual_rmsnorm(resid, hidden, w->o_bias, w->ffn_norm,
                                            cfg->rms_eps, batch, H);
        orc_linear(hidden, w->gate_up, gu, batch, H, 2 * I, 1);
        orc_silu_and_mul(gu, act, batch, I);
        orc_linear(act, w->down, hidden, batch, I, H, 1);
        orc_add_residual(resid, hidden, batch, H);
    }
}

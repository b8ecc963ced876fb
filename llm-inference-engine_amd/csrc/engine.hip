// Fused decoder engine: llmie_decoder_* and llmie_lm_head_sample.
// This is what LlamaSelfDecoder<T>::forward (self_decoder.cpp:24-122) and
// LlamaModel<T>::generateNextToken / LMHeadAndTopKSample (llama.cpp:219-318) run on.
//
// Differences from the reference's control flow, none of them numerical beyond rounding:
//   * zero allocation / zero synchronisation per token: all scratch lives in one caller-owned
//     workspace carved at create time (the reference mallocs+frees 5 buffers and syncs the
//     device ~10x per layer, self_attention.cpp:29-45,150);
//   * the position can live in device memory (step_dev) so the whole token step can be captured
//     once in a hipGraph and replayed;
//   * SwiGLU is the epilogue of the gate/up GEMV and the second residual add is the epilogue of
//     the down GEMV (reference: launchSiluAndMul / launchAddResidual as separate kernels).
#include "llmie_internal.h"

#include <cmath>
#include <cstdlib>
#include <new>
#include <vector>

using namespace llmie;

struct llmie_decoder {
    llmie_decoder_config cfg;
    std::vector<llmie_layer_weights> layers;
    int H, QKV, I;
    size_t esz;
    // workspace carve-up
    char *resid, *qkv, *mha, *normed, *act, *gu;
    void *attn_ws;
    size_t attn_ws_bytes;
    float2 *rope_table;  // [max_seq_len][head_size/2] (cos, sin), host-computed at create
    void *fp8_ws;        // LLMIE_W_FP8 engines: activation quantisation + split-K scratch of llmie_linear_fp8
    size_t fp8_ws_bytes;
    SlabWs slab_ws;      // fp32 slabs of the split-K projections (batch path, row-major engines)
    unsigned *tail_ticket = nullptr;   // one zeroed word: arrival ticket of the fused decode tail (llmie_lm_head_sample_next)
    int ragged = 0;      // set for the duration of a *_ragged forward: step_dev is the per-sequence context-length array
    // packed-weight batch path (gemv_max < batch <= 32): tile-packed images of the four matrices of every layer (built once at
    // create time into the caller's workspace: the MI355X's 288 GB buy a second, stream-friendly copy of the weights), the
    // activations between the packed kernels in the x32 layout, and the split-K slabs of the down projection
    struct PackedLayer {
        const unsigned char *qkv, *o, *gate_up, *down;
        const void *sc[4];   // int4: packed group-scale images of the four matrices (else null: the caller's row scales are used)
    };
    std::vector<PackedLayer> packed;
    int pk_wf = 0;                    // PKF_* of the engine's weight format, 0 = no packed path
    bool packed_only = false;         // LLMIE_DEC_PACKED_ONLY: the row-major matrices are gone, every path reads the images
    half_t *hx = nullptr, *actx = nullptr;   // x32 images: residual stream [32, H], SwiGLU output [32, I]
    half_t *mhax = nullptr;                  // x32 image of the attention output [32, H]
    float *pk_slab = nullptr;
    size_t pk_slab_floats = 0;
    // persistent chain launches of the packed path: one zeroed block of barrier counters per layer, one device error word
    unsigned char *pk_base = nullptr;   // first byte of the packed images (the workspace area carved for them)
    unsigned *pk_sync = nullptr, *pk_err = nullptr;
    unsigned long long *pk_stamps = nullptr;   // diagnostic: phase-edge timestamps of the LAST chain launch of a step
    // paged KV cache of the current llmie_decoder_forward_paged call (null: dense caches)
    const int32_t *page_table = nullptr;
    int max_pages = 0, num_pages = 0;
    // profiling (eager only)
    bool profiling = false;
    std::vector<hipEvent_t> ev;      // pairs: start, stop
    std::vector<int> ev_op;          // op kind of pair i
    size_t ev_used = 0;
};

// brackets one engine op with events when profiling
struct OpTimer {
    llmie_decoder *d;
    hipStream_t st;
    bool on;
    size_t idx;
    OpTimer(llmie_decoder *dec, int op, llmie_stream s) : d(dec), st(as_stream(s)), on(false), idx(0) {
        if (d->profiling && d->ev_used + 1 <= d->ev_op.size()) {
            idx = d->ev_used++;
            d->ev_op[idx] = op;
            on = hipEventRecord(d->ev[2 * idx], st) == hipSuccess;
        }
    }
    ~OpTimer() {
        if (on) (void)hipEventRecord(d->ev[2 * idx + 1], st);
    }
};
#define TIMED(op, expr)                 \
    do {                                \
        OpTimer _t(dec, op, stream);    \
        rc = (expr);                    \
    } while (0);                        \
    if (rc) return rc

static size_t align_up(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

static bool config_ok(const llmie_decoder_config *c) {
    if (!c) return false;
    if (c->head_num <= 0 || c->kv_head_num <= 0 || c->head_size <= 0 || c->inter_size <= 0 ||
        c->num_layers <= 0 || c->max_seq_len <= 0 || c->max_batch <= 0)
        return false;
    if (c->head_num % c->kv_head_num) return false;
    if (c->rotary_dim <= 0 || c->rotary_dim % 2) return false;
    if (c->dtype != LLMIE_F32 && c->dtype != LLMIE_F16) return false;
    if (c->dtype == LLMIE_F32 && c->wfmt != LLMIE_W_F32) return false;
    if (c->dtype == LLMIE_F16 && c->wfmt == LLMIE_W_F32) return false;
    if (c->kv_fmt != LLMIE_KV_NATIVE && c->kv_fmt != LLMIE_KV_FP8) return false;
    if (c->flags & ~(LLMIE_DEC_NO_PACKED_COPY | LLMIE_DEC_PACKED_ONLY)) return false;
    if ((c->flags & LLMIE_DEC_NO_PACKED_COPY) && (c->flags & LLMIE_DEC_PACKED_ONLY)) return false;
    if (c->kv_fmt == LLMIE_KV_FP8) {  // e4m3 cache: fp16 engines on the fused attention kernels only
        const int rep = c->head_num / c->kv_head_num;
        if (c->dtype != LLMIE_F16 || (c->head_size != 128 && c->head_size != 64) || (rep != 1 && rep != 2 && rep != 4)) return false;
    }
    return true;
}

struct Carve {
    size_t off = 0;
    size_t take(size_t bytes) {
        const size_t o = off;
        off += align_up(bytes);
        return o;
    }
};

// Largest decode batch on the GEMV path.  Its dot products are VALU work that grows with the batch while a packed-weight
// (MFMA) step is nearly flat, so the switch sits at the measured crossover (MI355X, 7B, ctx 512, tokens/s GEMV vs packed,
// round 2): fp16 b3 924 / 905, b4 1140 / 1186; int8 b2 918 / 828, b3 1172 / 1211; fp8 b2 804 / 720, b3 1009 / 1044;
// int4 (ctx 2048) b1 460 / 442, b2 710 / 812.  Re-measured at the end of round 3 (the GEMV had gained ~10 % since: four rows per
// iteration, two rows per instruction for int4): fp16 b4 1302 / 1203, b5 1516 / 1452, b6 1710 / 1731; int8 b3 1240 / 1244, b4 1477 /
// 1637; fp8 b3 1036 / 1171; int4 (ctx 512) b2 1012 / 939.
static int gemv_max_batch(llmie_weight_format wfmt) {
    switch (wfmt) {
        case LLMIE_W_F16: return 5;
        case LLMIE_W_INT8: return 2;
        case LLMIE_W_FP8: return 2;
        case LLMIE_W_INT4: return 2;
        default: return 4;
    }
}

// weight format / shapes the packed batch path covers; max rows it will be asked for
static int packed_wf(const llmie_decoder_config *c) {
    if (c->dtype != LLMIE_F16) return 0;
    const int wf = c->wfmt == LLMIE_W_F16 ? PKF_F16
                   : (c->wfmt == LLMIE_W_INT8 ? PKF_I8 : (c->wfmt == LLMIE_W_FP8 ? PKF_FP8 : (c->wfmt == LLMIE_W_INT4 && c->int4_group == 128 ? PKF_I4 : 0)));
    // batches up to the GEMV crossover never take the packed path: no second copy of the weights for such engines
    const int gemv_max = gemv_max_batch(c->wfmt);
    const bool only = (c->flags & LLMIE_DEC_PACKED_ONLY) != 0;   // the images are all there is: built whatever the batch
    if (!wf || (c->flags & LLMIE_DEC_NO_PACKED_COPY) || (!only && c->max_batch <= gemv_max)) return 0;
    if (only && c->max_batch > (wf == PKF_FP8 ? 16 : 32)) return 0;
    const int H = c->head_num * c->head_size, QKV = (c->head_num + 2 * c->kv_head_num) * c->head_size, I = c->inter_size;
    const int m = c->max_batch < 32 ? c->max_batch : 32;
    const int mm = m;
    if (H % 32 || I % 32) return 0;
    if (!pk_eligible(wf, mm, H, QKV, PKE_PLAIN) || !pk_eligible(wf, mm, H, H, PKE_PLAIN) || !pk_eligible(wf, mm, H, 2 * I, PKE_SWIGLU) ||
        !pk_eligible(wf, mm, I, H, PKE_PLAIN))
        return 0;
    return wf;
}
struct PackedCarve {
    size_t per_layer[4], layer_bytes, hx, actx, mhax, slab, sync, total;
};
static PackedCarve packed_carve(const llmie_decoder_config *c, int wf) {
    PackedCarve p{};
    if (!wf) return p;
    const int H = c->head_num * c->head_size, QKV = (c->head_num + 2 * c->kv_head_num) * c->head_size, I = c->inter_size;
    // (each matrix: weight image, then -- int4 -- its group-scale image)
    p.per_layer[0] = align_up(pk_packed_bytes(wf, QKV, H, 0)) + align_up(pk_packed_scale_bytes(wf, QKV, H, 0));
    p.per_layer[1] = align_up(pk_packed_bytes(wf, H, H, 0)) + align_up(pk_packed_scale_bytes(wf, H, H, 0));
    p.per_layer[2] = align_up(pk_packed_bytes(wf, 2 * I, H, 1)) + align_up(pk_packed_scale_bytes(wf, 2 * I, H, 1));
    p.per_layer[3] = align_up(pk_packed_bytes(wf, H, I, 0)) + align_up(pk_packed_scale_bytes(wf, H, I, 0));
    p.layer_bytes = p.per_layer[0] + p.per_layer[1] + p.per_layer[2] + p.per_layer[3];
    p.hx = align_up(static_cast<size_t>(H) * 64);
    p.actx = align_up(static_cast<size_t>(I) * 64);
    p.mhax = p.hx;
    const int m = c->max_batch < 32 ? c->max_batch : 32;
    size_t sl = 0;
    for (int mm = 1; mm <= m; ++mm) {
        const size_t f = pk_slab_floats(wf, mm, I, H);
        sl = f > sl ? f : sl;
    }
    p.slab = align_up(sl * sizeof(float) + 16);
    p.sync = align_up(static_cast<size_t>(c->num_layers) * pk_chain_sync_bytes()) + 256;   // barrier counters per layer + the error word
    p.total = p.layer_bytes * c->num_layers + p.hx + p.actx + p.mhax + p.slab + p.sync;
    return p;
}

// fp32 floats of split-K slab scratch the engine's projections can need at up to `rows` activation rows (0: none)
static size_t engine_slab_floats(const llmie_decoder_config *c, int rows) {
    int wbits;
    switch (c->wfmt) {
        case LLMIE_W_F16: wbits = 16; break;
        case LLMIE_W_INT8: wbits = 8; break;
        case LLMIE_W_INT4: wbits = 4; break;
        case LLMIE_W_FP8: wbits = WF_FP8; break;
        default: return 0;
    }
    if (c->dtype != LLMIE_F16) return 0;
    const int H = c->head_num * c->head_size, QKV = (c->head_num + 2 * c->kv_head_num) * c->head_size, I = c->inter_size;
    const int shapes[5][2] = {{H, QKV}, {H, H}, {H, 2 * I}, {I, H}, {H, c->vocab_size}};   // (the LM head of llmie_lm_head_sample)
    size_t m = 0;
    // every row count up to `rows` (the plan -- kernel form, K slices -- changes with it); the LM head may come in a different
    // format than the layers (fp16 beside int8 / fp8 layers): all formats for that shape
    const int lm_bits[4] = {16, 8, 4, WF_FP8};
    for (int r = 1; r <= rows; ++r) {
        for (int i = 0; i < 4; ++i) {
            const size_t f = linear_splitk_ws_floats(wbits, r, shapes[i][0], shapes[i][1]);
            m = f > m ? f : m;
        }
        for (int b : lm_bits) {
            const size_t f = linear_splitk_ws_floats(b, r, shapes[4][0], shapes[4][1]);
            m = f > m ? f : m;
        }
    }
    return m;
}

static size_t carve(const llmie_decoder_config *c, size_t *offs /*[12]*/) {
    const size_t e = c->dtype == LLMIE_F16 ? 2 : 4;
    const size_t B = c->max_batch, H = static_cast<size_t>(c->head_num) * c->head_size;
    const size_t QKV = static_cast<size_t>(c->head_num + 2 * c->kv_head_num) * c->head_size;
    const size_t I = c->inter_size;
    Carve k;
    offs[0] = k.take(B * H * e);        // resid
    offs[1] = k.take(B * QKV * e);      // qkv
    offs[2] = k.take(B * H * e);        // mha
    offs[3] = k.take(B * H * e);        // normed / attn_out
    offs[4] = k.take(B * I * e);        // act
    offs[5] = k.take(B * 2 * I * e);    // gate_up (unfused paths)
    offs[6] = k.take(llmie_decoder_mha_workspace_bytes(c->max_batch, c->head_num, c->head_size, c->max_seq_len));
    offs[7] = k.take(static_cast<size_t>(c->max_seq_len) * (c->head_size / 2) * sizeof(float2));  // RoPE table
    offs[8] = k.take(256);   // (spare)
    const int kmax = c->inter_size > static_cast<int>(H) ? c->inter_size : static_cast<int>(H);
    // fp8: three activation-quantisation units (normed input, attention output, SwiGLU output), see decoder_forward
    offs[9] = k.take(c->wfmt == LLMIE_W_FP8 ? 3 * llmie_linear_fp8_workspace_bytes(c->max_batch, kmax, 0) : 256);
    offs[10] = k.take(packed_carve(c, packed_wf(c)).total + 256);   // packed weight images + x32 activations + slabs
    offs[11] = k.take(engine_slab_floats(c, c->max_batch) * sizeof(float) + 256);   // split-K slabs of the row-major paths
    return k.off;
}

extern "C" size_t llmie_decoder_workspace_bytes(const llmie_decoder_config *cfg) {
    if (!config_ok(cfg)) return 0;
    size_t offs[12];
    return carve(cfg, offs);
}

// (re)build the tile-packed images of every layer from row-major matrices (create; llmie_decoder_repack)
static int pack_layers(llmie_decoder *d, const llmie_layer_weights *layers, hipStream_t st) {
    const llmie_decoder_config *cfg = &d->cfg;
    const PackedCarve pc = packed_carve(cfg, d->pk_wf);
    unsigned char *pb = d->pk_base;
    const int H = d->H, QKV = d->QKV, I = d->I;
    d->packed.resize(cfg->num_layers);
    int prc = LLMIE_OK;
    for (int l = 0; l < cfg->num_layers && prc == LLMIE_OK; ++l) {
        unsigned char *q = pb + static_cast<size_t>(l) * pc.layer_bytes;
        unsigned char *po = q + pc.per_layer[0], *pg = po + pc.per_layer[1], *pd = pg + pc.per_layer[2];
        const llmie_layer_weights &w = layers[l];
        const bool i4 = d->pk_wf == PKF_I4;
        unsigned char *sq = q + align_up(pk_packed_bytes(d->pk_wf, QKV, H, 0)), *so = po + align_up(pk_packed_bytes(d->pk_wf, H, H, 0));
        unsigned char *sg = pg + align_up(pk_packed_bytes(d->pk_wf, 2 * I, H, 1)), *sd = pd + align_up(pk_packed_bytes(d->pk_wf, H, I, 0));
        prc = pk_pack(d->pk_wf, w.qkv.data, i4 ? w.qkv.scale : nullptr, q, i4 ? sq : nullptr, QKV, H, 0, st);
        if (!prc) prc = pk_pack(d->pk_wf, w.o.data, i4 ? w.o.scale : nullptr, po, i4 ? so : nullptr, H, H, 0, st);
        if (!prc) prc = pk_pack(d->pk_wf, w.gate_up.data, i4 ? w.gate_up.scale : nullptr, pg, i4 ? sg : nullptr, 2 * I, H, 1, st);
        if (!prc) prc = pk_pack(d->pk_wf, w.down.data, i4 ? w.down.scale : nullptr, pd, i4 ? sd : nullptr, H, I, 0, st);
        d->packed[l] = llmie_decoder::PackedLayer{q, po, pg, pd, {i4 ? sq : nullptr, i4 ? so : nullptr, i4 ? sg : nullptr, i4 ? sd : nullptr}};
    }
    return prc;
}

// Weights changed in place (or, for a LLMIE_DEC_PACKED_ONLY engine, new weights altogether): rebuild the tile-packed images from
// `layers` (row-major matrices in the engine's format; scale / bias / gamma pointers replace the ones given at create).  Enqueued
// on `stream`; engines without images just take the new pointers.
extern "C" int llmie_decoder_repack(llmie_decoder *dec, const llmie_layer_weights *layers, llmie_stream stream) {
    LLMIE_REQUIRE(dec && layers, "decoder_repack: NULL pointer");
    for (int l = 0; l < dec->cfg.num_layers; ++l) {
        const llmie_layer_weights &w = layers[l];
        LLMIE_REQUIRE(w.attn_norm_gamma && w.ffn_norm_gamma && w.qkv.data && w.o.data && w.gate_up.data && w.down.data,
                      "decoder_repack: layer %d has a NULL weight", l);
    }
    int rc = LLMIE_OK;
    if (dec->pk_wf) rc = pack_layers(dec, layers, as_stream(stream));
    if (rc) return rc;
    dec->layers.assign(layers, layers + dec->cfg.num_layers);
    if (dec->packed_only)
        for (llmie_layer_weights &w : dec->layers) w.qkv.data = w.o.data = w.gate_up.data = w.down.data = nullptr;
    return LLMIE_OK;
}

extern "C" size_t llmie_decoder_resident_weight_bytes(const llmie_decoder_config *cfg) {
    if (!config_ok(cfg)) return 0;
    const size_t H = static_cast<size_t>(cfg->head_num) * cfg->head_size, QKV = static_cast<size_t>(cfg->head_num + 2 * cfg->kv_head_num) * cfg->head_size;
    const size_t I = cfg->inter_size, elems = QKV * H + H * H + 3 * H * I, rows = QKV + H + 2 * I + H;
    size_t row_major = 0, scales = 0;
    switch (cfg->wfmt) {
        case LLMIE_W_F16: row_major = elems * 2; break;
        case LLMIE_W_F32: row_major = elems * 4; break;
        case LLMIE_W_INT8: row_major = elems; scales = rows * 2; break;
        case LLMIE_W_FP8: row_major = elems; scales = rows * 4; break;
        case LLMIE_W_INT4: row_major = elems / 2; scales = (elems / (cfg->int4_group > 0 ? cfg->int4_group : 128)) * 2; break;
        default: return 0;
    }
    const int wf = packed_wf(cfg);
    const size_t images = wf ? packed_carve(cfg, wf).layer_bytes : 0;
    const size_t per_layer = ((cfg->flags & LLMIE_DEC_PACKED_ONLY) && wf ? 0 : row_major) + scales + images;
    return per_layer * cfg->num_layers;
}

extern "C" llmie_decoder *llmie_decoder_create(const llmie_decoder_config *cfg, const llmie_layer_weights *layers,
                                               void *workspace, size_t workspace_bytes) {
    if (!config_ok(cfg)) {
        set_error("decoder_create: invalid config");
        return nullptr;
    }
    if (!layers || !workspace) {
        set_error("decoder_create: NULL layers/workspace");
        return nullptr;
    }
    size_t offs[12];
    const size_t need = carve(cfg, offs);
    if (workspace_bytes < need) {
        set_error("decoder_create: workspace too small (%zu < %zu)", workspace_bytes, need);
        return nullptr;
    }
    if (reinterpret_cast<uintptr_t>(workspace) % 256) {
        set_error("decoder_create: workspace must be 256-byte aligned");
        return nullptr;
    }
    for (int l = 0; l < cfg->num_layers; ++l) {
        const llmie_layer_weights &w = layers[l];
        if (!w.attn_norm_gamma || !w.ffn_norm_gamma || !w.qkv.data || !w.o.data || !w.gate_up.data || !w.down.data) {
            set_error("decoder_create: layer %d has a NULL weight", l);
            return nullptr;
        }
        if (cfg->wfmt != LLMIE_W_F16 && cfg->wfmt != LLMIE_W_F32 &&
            (!w.qkv.scale || !w.o.scale || !w.gate_up.scale || !w.down.scale)) {
            set_error("decoder_create: layer %d lacks quantisation scales", l);
            return nullptr;
        }
    }
    llmie_decoder *d = new (std::nothrow) llmie_decoder();
    if (!d) return nullptr;
    d->cfg = *cfg;
    d->layers.assign(layers, layers + cfg->num_layers);
    d->H = cfg->head_num * cfg->head_size;
    d->QKV = (cfg->head_num + 2 * cfg->kv_head_num) * cfg->head_size;
    d->I = cfg->inter_size;
    d->esz = cfg->dtype == LLMIE_F16 ? 2 : 4;
    char *base = static_cast<char *>(workspace);
    d->resid = base + offs[0];
    d->qkv = base + offs[1];
    d->mha = base + offs[2];
    d->normed = base + offs[3];
    d->act = base + offs[4];
    d->gu = base + offs[5];
    d->attn_ws = base + offs[6];
    d->attn_ws_bytes = llmie_decoder_mha_workspace_bytes(cfg->max_batch, cfg->head_num, cfg->head_size, cfg->max_seq_len);
    d->rope_table = reinterpret_cast<float2 *>(base + offs[7]);
    d->tail_ticket = reinterpret_cast<unsigned *>(base + offs[8]);
    if (hipMemset(d->tail_ticket, 0, 256) != hipSuccess) {
        set_error("decoder_create: clearing the workspace words failed");
        delete d;
        return nullptr;
    }
    d->fp8_ws = base + offs[9];
    {
        const int kmax = cfg->inter_size > d->H ? cfg->inter_size : d->H;
        d->fp8_ws_bytes = cfg->wfmt == LLMIE_W_FP8 ? llmie_linear_fp8_workspace_bytes(cfg->max_batch, kmax, 0) : 0;
    }
    d->slab_ws = SlabWs{reinterpret_cast<float *>(base + offs[11]), engine_slab_floats(cfg, cfg->max_batch)};
    if (!d->slab_ws.floats) d->slab_ws.p = nullptr;
    d->pk_wf = packed_wf(cfg);
    if ((cfg->flags & LLMIE_DEC_PACKED_ONLY) && !d->pk_wf) {
        set_error("decoder_create: LLMIE_DEC_PACKED_ONLY needs max_batch <= 32 (fp8: 16) and shapes / a format the packed kernels take");
        delete d;
        return nullptr;
    }
    if (d->pk_wf) {
        // (uploads of the weights on other streams -- torch side streams are non-blocking -- must have landed before they are packed)
        if (hipDeviceSynchronize() != hipSuccess) {
            set_error("decoder_create: device synchronise failed");
            delete d;
            return nullptr;
        }
        // one-time re-tiling of every matrix into the stream-friendly image (null stream, synchronous: create is not on the
        // compute path); the row-major originals stay in use for batch <= gemv_max (GEMV) and for prefill
        const PackedCarve pc = packed_carve(cfg, d->pk_wf);
        unsigned char *pb = reinterpret_cast<unsigned char *>(base + offs[10]);
        d->pk_base = pb;
        const int prc = pack_layers(d, layers, nullptr);
        unsigned char *tail = pb + pc.layer_bytes * cfg->num_layers;
        d->hx = reinterpret_cast<half_t *>(tail);
        d->actx = reinterpret_cast<half_t *>(tail + pc.hx);
        d->mhax = reinterpret_cast<half_t *>(tail + pc.hx + pc.actx);
        d->pk_slab = reinterpret_cast<float *>(tail + pc.hx + pc.actx + pc.mhax);
        d->pk_slab_floats = (pc.slab - 16) / sizeof(float);
        d->pk_sync = reinterpret_cast<unsigned *>(tail + pc.hx + pc.actx + pc.mhax + pc.slab);
        d->pk_err = reinterpret_cast<unsigned *>(tail + pc.hx + pc.actx + pc.mhax + pc.slab + pc.sync - 256);
        if (prc != LLMIE_OK || hipMemset(tail, 0, pc.hx + pc.actx + pc.mhax) != hipSuccess || hipMemset(d->pk_sync, 0, pc.sync) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess) {
            set_error("decoder_create: packing the weights for the batch path failed");
            delete d;
            return nullptr;
        }
        if (cfg->flags & LLMIE_DEC_PACKED_ONLY) {   // the caller may free the row-major matrices now: forget them
            d->packed_only = true;
            for (llmie_layer_weights &w : d->layers) w.qkv.data = w.o.data = w.gate_up.data = w.down.data = nullptr;
        }
    }
    {
        // cos/sin of angle = pos / base^(2j/rot_dim) (rope_utils.cuh:6-19), evaluated on the host in fp32 with libm --
        // one synchronous upload at create time (the only host<->device copy the engine ever does)
        const int half = cfg->head_size / 2;
        std::vector<float2> tab(static_cast<size_t>(cfg->max_seq_len) * half);
        for (int pos = 0; pos < cfg->max_seq_len; ++pos)
            for (int j = 0; j < half; ++j) {
                float2 v{1.f, 0.f};
                if (j < cfg->rotary_dim / 2) {
                    const float ang = static_cast<float>(pos) / powf(cfg->rotary_base, static_cast<float>(2 * j) / static_cast<float>(cfg->rotary_dim));
                    v = float2{cosf(ang), sinf(ang)};
                }
                tab[static_cast<size_t>(pos) * half + j] = v;
            }
        if (hipMemcpy(d->rope_table, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("decoder_create: RoPE table upload failed");
            delete d;
            return nullptr;
        }
    }
    return d;
}

extern "C" void llmie_decoder_destroy(llmie_decoder *dec) {
    if (!dec) return;
    for (hipEvent_t e : dec->ev) (void)hipEventDestroy(e);
    delete dec;
}

extern "C" int llmie_decoder_profile_begin(llmie_decoder *dec, int max_events) {
    LLMIE_REQUIRE(dec && max_events > 0, "decoder_profile_begin: bad arguments");
    while (dec->ev.size() < 2 * static_cast<size_t>(max_events)) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) {
            set_error("decoder_profile_begin: hipEventCreate failed");
            return LLMIE_ERR_LAUNCH;
        }
        dec->ev.push_back(e);
    }
    dec->ev_op.assign(max_events, 0);
    dec->ev_used = 0;
    dec->profiling = true;
    return LLMIE_OK;
}

extern "C" int llmie_decoder_profile_end(llmie_decoder *dec, llmie_stream stream, double *ms_by_op, int *launches_by_op) {
    LLMIE_REQUIRE(dec && ms_by_op && launches_by_op, "decoder_profile_end: NULL pointer");
    dec->profiling = false;
    if (hipStreamSynchronize(as_stream(stream)) != hipSuccess) {
        set_error("decoder_profile_end: stream synchronise failed");
        return LLMIE_ERR_LAUNCH;
    }
    for (int i = 0; i < LLMIE_OP_COUNT; ++i) {
        ms_by_op[i] = 0.0;
        launches_by_op[i] = 0;
    }
    for (size_t i = 0; i < dec->ev_used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, dec->ev[2 * i], dec->ev[2 * i + 1]) == hipSuccess) {
            ms_by_op[dec->ev_op[i]] += ms;
            launches_by_op[dec->ev_op[i]] += 1;
        }
    }
    dec->ev_used = 0;
    return LLMIE_OK;
}

extern "C" int llmie_decoder_status(llmie_decoder *dec, llmie_stream stream) {
    LLMIE_REQUIRE(dec, "decoder_status: NULL decoder");
    if (hipStreamSynchronize(as_stream(stream)) != hipSuccess) {
        set_error("decoder_status: stream synchronise failed: %s", hipGetErrorString(hipGetLastError()));
        return LLMIE_ERR_LAUNCH;
    }
    if (!dec->pk_err) return LLMIE_OK;
    unsigned word = 0;
    if (hipMemcpy(&word, dec->pk_err, sizeof(word), hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("decoder_status: reading the device error word failed");
        return LLMIE_ERR_LAUNCH;
    }
    if (word) {
        (void)hipMemset(dec->pk_err, 0, sizeof(word));
        (void)hipMemset(dec->pk_sync, 0, static_cast<size_t>(dec->cfg.num_layers) * pk_chain_sync_bytes());   // counters are mid-flight: start over
        set_error("decoder: a grid barrier of the persistent chain launch timed out (code 0x%08x, phase %u): not every workgroup was "
                  "resident; set LLMIE_NO_CHAIN=1 to use the launch sequence", word, word & 0xfu);
        return LLMIE_ERR_LAUNCH;
    }
    return LLMIE_OK;
}

// Diagnostic, not on the product path: the chain launches of this decoder write s_memrealtime (100 MHz) stamps of their phase
// edges for every workgroup into `stamps_dev` ([256][16] uint64, device memory; every layer overwrites it); null disarms.
extern "C" int llmie_decoder_debug_stamps(llmie_decoder *dec, void *stamps_dev) {
    LLMIE_REQUIRE(dec, "decoder_debug_stamps: NULL decoder");
    dec->pk_stamps = static_cast<unsigned long long *>(stamps_dev);
    return LLMIE_OK;
}

// y = x . W^T (+bias)(+residual) | swiglu, dispatching on the engine's weight format
static int engine_linear(const llmie_decoder *d, llmie_weight_format fmt, const void *x, const llmie_matrix &w,
                         void *y, int M, int K, int N, bool swiglu, const void *residual, bool use_bias,
                         llmie_stream stream) {
    const void *bias = use_bias ? w.bias : nullptr;
    switch (fmt) {
        case LLMIE_W_F16:
            return linear_f16_nk((const half_t *)x, (const half_t *)w.data, (half_t *)y, M, K, N,
                                 swiglu ? EPI_SWIGLU_ : EPI_NONE_, (const half_t *)bias, (const half_t *)residual, d->slab_ws,
                                 as_stream(stream));
        case LLMIE_W_F32:
            if (swiglu) {
                int rc = llmie_linear(x, w.data, d->gu, M, K, N, 1, bias, nullptr, LLMIE_F32, nullptr, 0, stream);
                if (rc) return rc;
                return llmie_silu_and_mul(d->gu, y, M, N / 2, LLMIE_F32, stream);
            }
            return llmie_linear(x, w.data, y, M, K, N, 1, bias, residual, LLMIE_F32, nullptr, 0, stream);
        case LLMIE_W_INT8:
        case LLMIE_W_INT4: {
            const int bits = fmt == LLMIE_W_INT8 ? 8 : 4;
            // GEMV (M <= 8) and split-K MFMA (int8) paths take the SwiGLU epilogue directly
            int rc = linear_wq(bits, (const half_t *)x, w.data, (const half_t *)w.scale, (half_t *)y, M, K, N, d->cfg.int4_group,
                               swiglu ? EPI_SWIGLU_ : EPI_NONE_, (const half_t *)bias, (const half_t *)residual, nullptr, nullptr,
                               0.f, d->slab_ws, as_stream(stream));
            return rc;
        }
        case LLMIE_W_FP8: {
            if (swiglu) {
                int rc = linear_fp8((const half_t *)x, (const uint8_t *)w.data, (const float *)w.scale, (half_t *)d->gu, M, K, N,
                                    (const half_t *)bias, nullptr, d->fp8_ws, d->fp8_ws_bytes, d->slab_ws, as_stream(stream));
                if (rc) return rc;
                return llmie_silu_and_mul(d->gu, y, M, N / 2, LLMIE_F16, stream);
            }
            return linear_fp8((const half_t *)x, (const uint8_t *)w.data, (const float *)w.scale, (half_t *)y, M, K, N,
                              (const half_t *)bias, (const half_t *)residual, d->fp8_ws, d->fp8_ws_bytes, d->slab_ws, as_stream(stream));
        }
        default:
            set_error("engine: weight format %d not supported by this build", (int)fmt);
            return LLMIE_ERR_UNSUPPORTED;
    }
}

extern "C" int llmie_decoder_forward(llmie_decoder *dec, const void *hidden_in, void *hidden_out, void *k_cache,
                                     void *v_cache, int batch, int step, const int32_t *step_dev,
                                     llmie_stream stream) {
    LLMIE_REQUIRE(dec && hidden_in && hidden_out && k_cache && v_cache, "decoder_forward: NULL pointer");
    const llmie_decoder_config &c = dec->cfg;
    LLMIE_REQUIRE(batch >= 1 && batch <= c.max_batch, "decoder_forward: batch %d outside [1,%d]", batch, c.max_batch);
    LLMIE_REQUIRE(step_dev || (step >= 1 && step <= c.max_seq_len), "decoder_forward: step %d outside [1,%d]", step,
                  c.max_seq_len);
    const int H = dec->H, QKV = dec->QKV, I = dec->I;
    const llmie_dtype dt = c.dtype;
    int rc;
    if (hidden_out != hidden_in) {
        hipError_t e = hipMemcpyAsync(hidden_out, hidden_in, static_cast<size_t>(batch) * H * dec->esz,
                                      hipMemcpyDeviceToDevice, as_stream(stream));
        if (e != hipSuccess) {
            set_error("decoder_forward: copy failed: %s", hipGetErrorString(e));
            return LLMIE_ERR_LAUNCH;
        }
    }
    void *h = hidden_out;  // running hidden state, updated in place like decoder_output in the reference

    // ---- fused fp16 decode path (batch <= 8): 5 launches per layer ----
    //   qkv  = rmsnorm(h)*g1 . Wqkv^T                      (norm fused into the GEMV prologue)
    //   mha  = attention(rope(q), rope(k) -> cache, v)     (RoPE + bias + KV append fused into the attention)
    //   h   += mha . Wo^T                                   (residual epilogue; h is the residual stream)
    //   act  = swiglu(rmsnorm(h + o.bias)*g2 . Wgu^T)      (norm prologue + SwiGLU epilogue)
    //   h   += act . Wd^T
    // Same math as self_decoder.cpp:69-119 (residual updated before the o.bias add, as the reference).
    static const int fused_off = getenv("LLMIE_NO_FUSED_DECODE") ? 1 : 0;
    const int rep = c.head_num / c.kv_head_num;
    const bool hs_ok = c.head_size == 32 || c.head_size == 64 || c.head_size == 128 || c.head_size == 256;
    const bool rep_ok = rep == 1 || rep == 2 || rep == 4 || rep == 8;
    const int wbits = c.wfmt == LLMIE_W_F16 ? 16 : (c.wfmt == LLMIE_W_INT8 ? 8 : (c.wfmt == LLMIE_W_INT4 ? 4 : 0));
    const bool fp8 = c.wfmt == LLMIE_W_FP8;
    const int kv8 = c.kv_fmt == LLMIE_KV_FP8;
    const float k_scale = c.k_scale > 0.f ? c.k_scale : 1.f, v_scale = c.v_scale > 0.f ? c.v_scale : 1.f;
    // GEMV form up to gemv_max rows (its dot products are VALU work that grows with the batch), MFMA split-K above
    // measured crossover on MI355X (7B, ctx 512, tokens/s GEMV vs split-K): fp16 b4 1157/1143, b6 1496/1592; int8 b4 1416/1404,
    // b6 1672/1962; fp8 b3 947/944, b4 1127/1213; int4 b2 796/745, b3 896/1074
    const int gemv_max = gemv_max_batch(c.wfmt);
    static const int batch_fused_off = getenv("LLMIE_NO_FUSED_BATCH") ? 1 : 0;
    const bool int4_ok = wbits == 4 && c.int4_group == 128 && batch <= 64;  // int4 MFMA form: group-128 scales, 64 rows per pass
    const bool batch_path_ok = !batch_fused_off && c.dtype == LLMIE_F16 && (wbits == 16 || wbits == 8 || int4_ok || fp8) && hs_ok &&
                               rep_ok && batch <= 128 && H % 256 == 0 && I % 256 == 0 && H >= 512 && I >= 512 && splitk_rownorm_eligible(H);
    const bool gemv_ok = !dec->packed_only && (batch <= gemv_max || !batch_path_ok) &&
                         (wbits == 16 ? gemv_f16_eligible(batch, H, h, dec->layers[0].qkv.data)
                                      : ((wbits != 0 || fp8) && ksplit_eligible(batch, H, fp8 ? 8 : wbits)));
    if (!fused_off && c.dtype == LLMIE_F16 && (wbits != 0 || fp8) && hs_ok && rep_ok && H % 8 == 0 && gemv_ok) {
        hipStream_t st = as_stream(stream);
        // (the in-launch merge of the attention partials -- the tickets argument of llmie_decoder_mha_rope -- measured SLOWER than the
        // separate 4.8 us merge kernel on MI355X, 2.97 vs 2.81 ms per token: every workgroup pays the release fence; the engine
        // always uses the merge kernel)
        // y = [swiglu]( rmsnorm(x + pre_bias)*gamma . W^T ) + residual on the streaming GEMV of the weight format
        auto lin = [&](const void *x, const llmie_matrix &w, void *y, int K, int N, int epi, const void *residual,
                       const void *gamma, const void *pre_bias) -> int {
            if (fp8) {
                // e4m3 weights x per-token e4m3 activations; the GEMV handles K rows that fit its register budget,
                // longer ones (down projection at batch > 2) take the MFMA launch sequence of llmie_linear_fp8
                if (ksplit_eligible(batch, K, 8))
                    return linear_fp8_gemv((const half_t *)x, (const uint8_t *)w.data, (const float *)w.scale, (half_t *)y, batch,
                                           K, N, epi, nullptr, (const half_t *)residual, (const half_t *)gamma,
                                           (const half_t *)pre_bias, c.rms_eps, st);
                if (epi != EPI_NONE_ || gamma) {
                    set_error("engine: fp8 projection K=%d at batch %d has no fused form", K, batch);
                    return LLMIE_ERR_UNSUPPORTED;
                }
                return linear_fp8((const half_t *)x, (const uint8_t *)w.data, (const float *)w.scale, (half_t *)y, batch, K, N, nullptr,
                                  (const half_t *)residual, dec->fp8_ws, dec->fp8_ws_bytes, dec->slab_ws, st);
            }
            if (wbits == 16) {
                if (gamma)
                    return linear_f16_nk_norm((const half_t *)x, (const half_t *)w.data, (half_t *)y, batch, K, N, epi, nullptr,
                                              (const half_t *)residual, (const half_t *)gamma, (const half_t *)pre_bias,
                                              c.rms_eps, st);
                return linear_f16_nk((const half_t *)x, (const half_t *)w.data, (half_t *)y, batch, K, N, epi, nullptr,
                                     (const half_t *)residual, dec->slab_ws, st);
            }
            return linear_wq(wbits, (const half_t *)x, w.data, (const half_t *)w.scale, (half_t *)y, batch, K, N, c.int4_group,
                             epi, nullptr, (const half_t *)residual, (const half_t *)gamma, (const half_t *)pre_bias,
                             c.rms_eps, dec->slab_ws, st);
        };
        for (int l = 0; l < c.num_layers; ++l) {
            const llmie_layer_weights &w = dec->layers[l];
            TIMED(LLMIE_OP_QKV_GEMM, lin(h, w.qkv, dec->qkv, H, QKV, EPI_NONE_, nullptr, w.attn_norm_gamma, nullptr));
            TIMED(LLMIE_OP_MHA, decoder_mha_rope(dec->qkv, w.qkv.bias, k_cache, v_cache, dec->mha, l, batch, c.head_num,
                                                 c.kv_head_num, c.head_size, c.max_seq_len, step, step_dev, dec->attn_ws,
                                                 dec->attn_ws_bytes, dec->rope_table, c.rotary_dim,
                                                 nullptr, dt, st, nullptr,
                                                 nullptr, kv8, k_scale, v_scale, dec->page_table, dec->max_pages, dec->num_pages,
                                                 dec->ragged));
            TIMED(LLMIE_OP_O_GEMM, lin(dec->mha, w.o, h, H, H, EPI_NONE_, h, nullptr, nullptr));
            TIMED(LLMIE_OP_GATE_UP_SWIGLU, lin(h, w.gate_up, dec->act, H, 2 * I, EPI_SWIGLU_, nullptr, w.ffn_norm_gamma, w.o.bias));
            TIMED(LLMIE_OP_DOWN_GEMM, lin(dec->act, w.down, h, I, H, EPI_NONE_, h, nullptr, nullptr));
        }
        return LLMIE_OK;
    }

    // ---- packed-weight batch path (gemv_max < batch <= 32; fp16 / int8 weights, fp8 up to 16 rows): 6 launches per layer, no
    // fp32 slab round trip except the down projection's K split --
    //   qkv  = rmsnorm(h) * g1 . Wqkv^T            norm in the kernel's prologue (per-token factor in its epilogue), fp16 qkv rows
    //   mha  = attention(rope(q), rope(k) -> cache, v)
    //   hx   = h + mha . Wo^T                       residual epilogue, x32 image of the residual stream
    //   actx = swiglu(rmsnorm(hx + o.bias) * g2 . Wgu^T)
    //   hx  += actx . Wd^T                          K split over workgroups (K = inter_size) + one reduce launch
    // Same math as self_decoder.cpp:69-119.  Weights come from the tile-packed images built at create time.
    static const int packed_off = getenv("LLMIE_NO_PACKED_BATCH") ? 1 : 0;
    // fp8: the per-launch activation quantisation (amax + conversions of the whole register slice) costs ~5 us at 32 rows;
    // measured (7B, ctx 512, tokens/s packed vs split-K batch path): b8 2705 / 2506, b16 5011 / 4450, b24 5656 / 5833, b32 7103 / 7120
    const int pk_rows_max = fp8 ? 16 : 32;
    if (dec->packed_only && (packed_off || fused_off || batch > pk_rows_max || !hs_ok || !rep_ok)) {
        set_error("decoder_forward: a LLMIE_DEC_PACKED_ONLY engine decodes on the packed kernels only (batch <= %d, no path switch)", pk_rows_max);
        return LLMIE_ERR_UNSUPPORTED;
    }
    if (!packed_off && !fused_off && dec->pk_wf && (batch > gemv_max || dec->packed_only) && batch <= pk_rows_max && hs_ok && rep_ok) {
        hipStream_t st = as_stream(stream);
        const int wf = dec->pk_wf;
        half_t *hh = static_cast<half_t *>(h);
        half_t *qkvb = reinterpret_cast<half_t *>(dec->qkv);
        // Round 3, OPT-IN (LLMIE_CHAIN=1): ONE persistent launch per layer for the chain O -> gate/up -> down (-> slab reduce) -> next
        // layer's QKV (pk_gemm.cuh, pk_chain_kernel: grid barriers between the phases, the next phase's weight ring prefetched across
        // each): 2 launches per layer (attention + chain) instead of 6, the launches' own code per phase (bit-identical outputs).
        // Measured SLOWER than the launch sequence (int8, batch 32, MI355X: 94 us per chain against 78 us for the five launches it
        // replaces; phase-edge timestamps, DESIGN.md section 9: a phase inside the chain takes what its launch takes -- the ~8 us of
        // fixed cost per projection is the kernel's own prologue (activation slice, norm) and epilogue, not the launch -- and a grid
        // barrier costs 5.5-7 us against ~2.5 us for a launch boundary), so the launch sequence stays the default.
        static const int chain_off = getenv("LLMIE_CHAIN") ? 0 : 1;
        bool chain = !chain_off && dec->pk_sync != nullptr;
        if (chain) {
            // probe: can every projection of a layer join a chain at this batch?  (pk_chain_add validates shapes / plans / LDS)
            const llmie_layer_weights &w = dec->layers[0];
            const llmie_decoder::PackedLayer &pw = dec->packed[0];
            PkChain ch;
            pk_chain_begin(&ch, wf, batch);
            chain = ch.ok &&
                    !pk_chain_add(&ch, 0, dec->mhax, pw.o, pw.sc[1] ? pw.sc[1] : w.o.scale, dec->hx, H, H, PKE_PLAIN, PKX_X | PKX_Y | PKX_RES, dec->hx, nullptr, nullptr, 0.f, nullptr, 0) &&
                    !pk_chain_add(&ch, 1, dec->hx, pw.gate_up, pw.sc[2] ? pw.sc[2] : w.gate_up.scale, dec->actx, H, 2 * I, PKE_SWIGLU, PKX_X | PKX_Y, nullptr,
                                  static_cast<const half_t *>(w.ffn_norm_gamma), static_cast<const half_t *>(w.o.bias), c.rms_eps, nullptr, 0) &&
                    !pk_chain_add(&ch, 2, dec->actx, pw.down, pw.sc[3] ? pw.sc[3] : w.down.scale, dec->hx, I, H, PKE_PLAIN, PKX_X | PKX_RES | PKX_Y, dec->hx, nullptr, nullptr,
                                  0.f, dec->pk_slab, dec->pk_slab_floats) &&
                    (c.num_layers == 1 ||
                     !pk_chain_add(&ch, 4, dec->hx, pw.qkv, pw.sc[0] ? pw.sc[0] : w.qkv.scale, qkvb, H, QKV, PKE_PLAIN, PKX_X, nullptr,
                                   static_cast<const half_t *>(w.attn_norm_gamma), nullptr, c.rms_eps, nullptr, 0));
        }
        if (chain) {
            for (int l = 0; l < c.num_layers; ++l) {
                const llmie_layer_weights &w = dec->layers[l];
                const llmie_decoder::PackedLayer &pw = dec->packed[l];
                const bool first = l == 0, last = l + 1 == c.num_layers;
                if (first)
                    TIMED(LLMIE_OP_QKV_GEMM, pk_linear(wf, hh, pw.qkv, pw.sc[0] ? pw.sc[0] : w.qkv.scale, qkvb, batch, H, QKV, PKE_PLAIN, 0, nullptr,
                                                       static_cast<const half_t *>(w.attn_norm_gamma), nullptr, c.rms_eps, nullptr, 0, st));
                TIMED(LLMIE_OP_MHA, decoder_mha_rope(dec->qkv, w.qkv.bias, k_cache, v_cache, dec->mhax, l, batch, c.head_num, c.kv_head_num,
                                                     c.head_size, c.max_seq_len, step, step_dev, dec->attn_ws, dec->attn_ws_bytes,
                                                     dec->rope_table, c.rotary_dim, nullptr, dt, st, nullptr, nullptr, kv8, k_scale, v_scale,
                                                     dec->page_table, dec->max_pages, dec->num_pages, dec->ragged, 1));
                PkChain ch;
                pk_chain_begin(&ch, wf, batch);
                rc = pk_chain_add(&ch, 0, dec->mhax, pw.o, pw.sc[1] ? pw.sc[1] : w.o.scale, dec->hx, H, H, PKE_PLAIN, PKX_X | PKX_Y | (first ? 0 : PKX_RES),
                                  first ? hh : dec->hx, nullptr, nullptr, 0.f, nullptr, 0);
                if (!rc) rc = pk_chain_add(&ch, 1, dec->hx, pw.gate_up, pw.sc[2] ? pw.sc[2] : w.gate_up.scale, dec->actx, H, 2 * I, PKE_SWIGLU, PKX_X | PKX_Y, nullptr,
                                           static_cast<const half_t *>(w.ffn_norm_gamma), static_cast<const half_t *>(w.o.bias), c.rms_eps, nullptr, 0);
                if (!rc) rc = pk_chain_add(&ch, 2, dec->actx, pw.down, pw.sc[3] ? pw.sc[3] : w.down.scale, last ? hh : dec->hx, I, H, PKE_PLAIN,
                                           PKX_X | PKX_RES | (last ? 0 : PKX_Y), dec->hx, nullptr, nullptr, 0.f, dec->pk_slab, dec->pk_slab_floats);
                if (!rc && !last) {
                    const llmie_layer_weights &wn = dec->layers[l + 1];
                    const llmie_decoder::PackedLayer &pn = dec->packed[l + 1];
                    rc = pk_chain_add(&ch, 4, dec->hx, pn.qkv, pn.sc[0] ? pn.sc[0] : wn.qkv.scale, qkvb, H, QKV, PKE_PLAIN, PKX_X, nullptr,
                                      static_cast<const half_t *>(wn.attn_norm_gamma), nullptr, c.rms_eps, nullptr, 0);
                }
                if (rc) return rc;
                ch.stamps = dec->pk_stamps;   // (diagnostic; null unless llmie_decoder_debug_stamps armed it)
                TIMED(LLMIE_OP_CHAIN, pk_chain_launch(&ch, dec->pk_sync + static_cast<size_t>(l) * (pk_chain_sync_bytes() / sizeof(unsigned)), dec->pk_err, st));
            }
            return LLMIE_OK;
        }
        for (int l = 0; l < c.num_layers; ++l) {
            const llmie_layer_weights &w = dec->layers[l];
            const llmie_decoder::PackedLayer &pw = dec->packed[l];
            const bool first = l == 0, last = l + 1 == c.num_layers;
            // layer 0 reads the caller's row-major hidden state; from its output projection on the residual stream lives in hx
            TIMED(LLMIE_OP_QKV_GEMM, pk_linear(wf, first ? hh : dec->hx, pw.qkv, pw.sc[0] ? pw.sc[0] : w.qkv.scale, qkvb, batch, H, QKV, PKE_PLAIN, first ? 0 : PKX_X,
                                               nullptr, static_cast<const half_t *>(w.attn_norm_gamma), nullptr, c.rms_eps, nullptr, 0, st));
            TIMED(LLMIE_OP_MHA, decoder_mha_rope(dec->qkv, w.qkv.bias, k_cache, v_cache, dec->mhax, l, batch, c.head_num, c.kv_head_num,
                                                 c.head_size, c.max_seq_len, step, step_dev, dec->attn_ws, dec->attn_ws_bytes,
                                                 dec->rope_table, c.rotary_dim, nullptr, dt, st, nullptr, nullptr, kv8, k_scale, v_scale,
                                                 dec->page_table, dec->max_pages, dec->num_pages, dec->ragged, 1));
            TIMED(LLMIE_OP_O_GEMM, pk_linear(wf, dec->mhax, pw.o, pw.sc[1] ? pw.sc[1] : w.o.scale, dec->hx, batch, H, H, PKE_PLAIN, PKX_X | PKX_Y | (first ? 0 : PKX_RES),
                                             first ? hh : dec->hx, nullptr, nullptr, 0.f, nullptr, 0, st));
            TIMED(LLMIE_OP_GATE_UP_SWIGLU, pk_linear(wf, dec->hx, pw.gate_up, pw.sc[2] ? pw.sc[2] : w.gate_up.scale, dec->actx, batch, H, 2 * I, PKE_SWIGLU, PKX_X | PKX_Y,
                                                     nullptr, static_cast<const half_t *>(w.ffn_norm_gamma), static_cast<const half_t *>(w.o.bias),
                                                     c.rms_eps, nullptr, 0, st));
            TIMED(LLMIE_OP_DOWN_GEMM, pk_linear(wf, dec->actx, pw.down, pw.sc[3] ? pw.sc[3] : w.down.scale, last ? hh : dec->hx, batch, I, H, PKE_PLAIN,
                                                PKX_X | PKX_RES | (last ? 0 : PKX_Y), dec->hx, nullptr, nullptr, 0.f, dec->pk_slab,
                                                dec->pk_slab_floats, st));
        }
        return LLMIE_OK;
    }

    // ---- fused batch decode path (8 < batch <= 128; fp16, int8, int4 (<= 64) or fp8 weights): 8 launches per layer (+ the attention
    // merge when the context spans several chunks) instead of 11-12.  Every projection is a split-K MFMA launch that
    // leaves fp32 partial slabs; the consumer of each slab does the reduction:
    //   qkv slabs            -> read directly by the attention launch (q rows, new k/v rows; + scale, bias, RoPE, append)
    //   o / down slabs       -> splitk_rownorm: reduction + residual stream update + the NEXT RMSNorm in one launch
    //   gate_up slabs        -> finalize with the SwiGLU epilogue
    // (each small dependent launch costs ~4.5 us on MI355X, a third of a batch-32 int8 layer before this fusion)
    // fp8: activations enter every projection as per-token e4m3; the row kernel emits them directly for the qkv and
    // gate_up inputs, the attention and SwiGLU outputs take a quantize_rows launch each (10 launches per layer).
    if (batch_path_ok) {
        hipStream_t st = as_stream(stream);
        half_t *hh = static_cast<half_t *>(h), *resid = reinterpret_cast<half_t *>(dec->resid);
        half_t *mha = reinterpret_cast<half_t *>(dec->mha), *act = reinterpret_cast<half_t *>(dec->act);
        const int fmt = fp8 ? WF_FP8 : wbits;
        const size_t unit = dec->fp8_ws_bytes, kmax = static_cast<size_t>(I > H ? I : H);
        auto xq_of = [&](int u) { return fp8 ? reinterpret_cast<uint8_t *>(dec->fp8_ws) + u * unit : nullptr; };
        auto xs_of = [&](int u) {
            return fp8 ? reinterpret_cast<float *>(xq_of(u) + ((static_cast<size_t>(c.max_batch) * kmax + 255) & ~size_t(255))) : nullptr;
        };
        uint8_t *xqA = xq_of(0), *xqB = xq_of(1), *xqC = xq_of(2);
        float *xsA = xs_of(0), *xsB = xs_of(1), *xsC = xs_of(2);
        auto scale_of = [&](const llmie_matrix &m, const float *xs) {
            if (fp8) return SlabScale{nullptr, static_cast<const float *>(m.scale), xs};
            return SlabScale{wbits == 8 ? static_cast<const half_t *>(m.scale) : nullptr, nullptr, nullptr};
        };
        // int4: the group scales are applied inside the split-K kernel (the slabs hold scaled values)
        auto gs_of = [&](const llmie_matrix &m) { return wbits == 4 ? static_cast<const half_t *>(m.scale) : nullptr; };
        // self_decoder.cpp:77 (first layer only: later ones get it from the previous layer's down-projection epilogue)
        TIMED(LLMIE_OP_ATTN_NORM, llmie_rmsnorm(h, dec->resid, dec->layers[0].attn_norm_gamma, c.rms_eps, batch, H, dt, stream));
        if (fp8) TIMED(LLMIE_OP_ATTN_NORM, quantize_rows_fp8(hh, xqA, xsA, batch, H, st));
        for (int l = 0; l < c.num_layers; ++l) {
            const llmie_layer_weights &w = dec->layers[l];
            const bool last = l + 1 == c.num_layers;
            const void *xin = fp8 ? static_cast<const void *>(xqA) : hh;
            SplitKSlabs sk;
            TIMED(LLMIE_OP_QKV_GEMM, linear_splitk_partial(fmt, xin, w.qkv.data, batch, H, QKV, st, &sk, dec->slab_ws, gs_of(w.qkv)));
            const SlabScale qsc = scale_of(w.qkv, xsA);
            TIMED(LLMIE_OP_MHA, decoder_mha_rope(nullptr, w.qkv.bias, k_cache, v_cache, dec->mha, l, batch, c.head_num,
                                                 c.kv_head_num, c.head_size, c.max_seq_len, step, step_dev, dec->attn_ws,
                                                 dec->attn_ws_bytes, dec->rope_table, c.rotary_dim, nullptr, dt, st, &sk, &qsc, kv8,
                                                 k_scale, v_scale, dec->page_table, dec->max_pages, dec->num_pages, dec->ragged));
            if (fp8) TIMED(LLMIE_OP_O_GEMM, quantize_rows_fp8(mha, xqB, xsB, batch, H, st));
            TIMED(LLMIE_OP_O_GEMM, linear_splitk_partial(fmt, fp8 ? static_cast<const void *>(xqB) : mha, w.o.data, batch, H, H, st, &sk, dec->slab_ws, gs_of(w.o)));
            // self_decoder.cpp:92  h += resid; resid = h; h += o.bias; h = rmsnorm(h, ffn_gamma)
            TIMED(LLMIE_OP_FFN_NORM, splitk_rownorm(sk, scale_of(w.o, xsB), static_cast<const half_t *>(w.o.bias), resid,
                                                    static_cast<const half_t *>(w.ffn_norm_gamma), c.rms_eps,
                                                    fp8 ? nullptr : hh, xqA, xsA, st));
            // ffn.cpp:105-122  act = silu(h.Wg^T) * (h.Wu^T)
            TIMED(LLMIE_OP_GATE_UP_SWIGLU, linear_splitk_partial(fmt, xin, w.gate_up.data, batch, H, 2 * I, st, &sk, dec->slab_ws, gs_of(w.gate_up)));
            TIMED(LLMIE_OP_GATE_UP_SWIGLU, splitk_finalize(sk, scale_of(w.gate_up, xsA), act, EPI_SWIGLU_, nullptr, nullptr, st));
            if (fp8) TIMED(LLMIE_OP_DOWN_GEMM, quantize_rows_fp8(act, xqC, xsC, batch, I, st));
            TIMED(LLMIE_OP_DOWN_GEMM, linear_splitk_partial(fmt, fp8 ? static_cast<const void *>(xqC) : act, w.down.data, batch, I, H, st, &sk, dec->slab_ws, gs_of(w.down)));
            // ffn.cpp:132 + self_decoder.cpp:111 h = act.Wd^T + resid, then the next layer's resid = h; h = rmsnorm(h)
            const void *next_gamma = last ? nullptr : dec->layers[l + 1].attn_norm_gamma;
            TIMED(LLMIE_OP_ATTN_NORM, splitk_rownorm(sk, scale_of(w.down, xsC), nullptr, resid, static_cast<const half_t *>(next_gamma),
                                                     c.rms_eps, (fp8 && !last) ? nullptr : hh, last ? nullptr : xqA, xsA, st));
        }
        return LLMIE_OK;
    }

    if (dec->page_table) {
        set_error("decoder_forward_paged: the paged KV cache needs the fused decode paths (fp16 engines, batch <= 128)");
        return LLMIE_ERR_UNSUPPORTED;
    }
    if (kv8) {
        set_error("decoder_forward: the fp8 KV cache needs the fused decode paths (batch <= 128, fp16/int8/int4/fp8 weights with "
                  "H and I multiples of 256 at batch > 8)");
        return LLMIE_ERR_UNSUPPORTED;
    }
    for (int l = 0; l < c.num_layers; ++l) {
        const llmie_layer_weights &w = dec->layers[l];
        // self_decoder.cpp:77  resid = h ; h = rmsnorm(h)
        TIMED(LLMIE_OP_ATTN_NORM, llmie_rmsnorm(h, dec->resid, w.attn_norm_gamma, c.rms_eps, batch, H, dt, stream));
        // self_attention.cpp:79  qkv = h . Wqkv^T   (bias is applied inside the MHA kernel, as the reference)
        TIMED(LLMIE_OP_QKV_GEMM, engine_linear(dec, c.wfmt, h, w.qkv, dec->qkv, batch, H, QKV, false, nullptr, false, stream));
        if (hs_ok && rep_ok) {
            // :100-:108 RoPE at position step-1 + fused masked MHA with KV append, one launch (+ merge)
            TIMED(LLMIE_OP_MHA, decoder_mha_rope(dec->qkv, w.qkv.bias, k_cache, v_cache, dec->mha, l, batch, c.head_num,
                                                 c.kv_head_num, c.head_size, c.max_seq_len, step, step_dev, dec->attn_ws,
                                                 dec->attn_ws_bytes, dec->rope_table, c.rotary_dim, nullptr, dt,
                                                 as_stream(stream), nullptr, nullptr, 0, 1.f, 1.f, nullptr, 0, 0, dec->ragged));
        } else {
            if (dec->ragged) {
                set_error("decoder_forward_ragged: needs head_size in {32,64,128,256} and head_num/kv_head_num in {1,2,4,8}");
                return LLMIE_ERR_UNSUPPORTED;
            }
            // :100 RoPE at position step-1
            TIMED(LLMIE_OP_ROPE, llmie_rope_decode(dec->qkv, batch, c.head_num, c.kv_head_num, c.head_size, step, step_dev,
                                                   c.rotary_dim, c.rotary_base, dt, stream));
            // :108 fused masked MHA with KV append
            TIMED(LLMIE_OP_MHA, llmie_decoder_mha(dec->qkv, w.qkv.bias, k_cache, v_cache, dec->mha, l, batch, c.head_num,
                                                  c.kv_head_num, c.head_size, c.max_seq_len, step, step_dev, dec->attn_ws,
                                                  dec->attn_ws_bytes, dt, stream));
        }
        // :131 output projection (no bias here: the fused norm below adds o.bias, self_decoder.cpp:92-98)
        TIMED(LLMIE_OP_O_GEMM, engine_linear(dec, c.wfmt, dec->mha, w.o, h, batch, H, H, false, nullptr, false, stream));
        // self_decoder.cpp:92  h += resid; resid = h; h += o.bias; h = rmsnorm(h, ffn_gamma)
        TIMED(LLMIE_OP_FFN_NORM, llmie_fused_add_bias_residual_rmsnorm(dec->resid, h, w.o.bias, w.ffn_norm_gamma,
                                                                       c.rms_eps, batch, H, dt, stream));
        // ffn.cpp:105-122  act = silu(h.Wg^T) * (h.Wu^T)
        TIMED(LLMIE_OP_GATE_UP_SWIGLU, engine_linear(dec, c.wfmt, h, w.gate_up, dec->act, batch, H, 2 * I, true, nullptr, false, stream));
        // ffn.cpp:132 + self_decoder.cpp:111  h = act . Wd^T + resid
        TIMED(LLMIE_OP_DOWN_GEMM, engine_linear(dec, c.wfmt, dec->act, w.down, h, batch, I, H, false, dec->resid, false, stream));
    }
    return LLMIE_OK;
}

// Ragged batch (continuous batching): ctx_len_dev[b] = context length of sequence b including this step's token.  Every
// projection is row-wise, so only the attention launch sees the difference (RoPE position, append slot, span per sequence).
extern "C" int llmie_decoder_forward_ragged(llmie_decoder *dec, const void *hidden_in, void *hidden_out, void *k_cache,
                                            void *v_cache, int batch, const int32_t *ctx_len_dev, llmie_stream stream) {
    LLMIE_REQUIRE(dec && ctx_len_dev, "decoder_forward_ragged: NULL pointer");
    dec->ragged = 1;
    const int rc = llmie_decoder_forward(dec, hidden_in, hidden_out, k_cache, v_cache, batch, -1, ctx_len_dev, stream);
    dec->ragged = 0;
    return rc;
}

extern "C" int llmie_decoder_forward_paged_ragged(llmie_decoder *dec, const void *hidden_in, void *hidden_out, void *k_pool,
                                                  void *v_pool, const int32_t *block_table, int max_pages, int num_pages, int batch,
                                                  const int32_t *ctx_len_dev, llmie_stream stream) {
    LLMIE_REQUIRE(dec && ctx_len_dev, "decoder_forward_paged_ragged: NULL pointer");
    dec->ragged = 1;
    const int rc = llmie_decoder_forward_paged(dec, hidden_in, hidden_out, k_pool, v_pool, block_table, max_pages, num_pages, batch,
                                               -1, ctx_len_dev, stream);
    dec->ragged = 0;
    return rc;
}

extern "C" int llmie_decoder_forward_paged(llmie_decoder *dec, const void *hidden_in, void *hidden_out, void *k_pool, void *v_pool,
                                           const int32_t *block_table, int max_pages, int num_pages, int batch, int step,
                                           const int32_t *step_dev, llmie_stream stream) {
    LLMIE_REQUIRE(dec && block_table, "decoder_forward_paged: NULL pointer");
    LLMIE_REQUIRE(max_pages > 0 && num_pages > 0 &&
                      static_cast<long long>(max_pages) * LLMIE_KV_PAGE_TOKENS >= dec->cfg.max_seq_len,
                  "decoder_forward_paged: max_pages * %d must cover max_seq_len %d", LLMIE_KV_PAGE_TOKENS, dec->cfg.max_seq_len);
    dec->page_table = block_table;
    dec->max_pages = max_pages;
    dec->num_pages = num_pages;
    const int rc = llmie_decoder_forward(dec, hidden_in, hidden_out, k_pool, v_pool, batch, step, step_dev, stream);
    dec->page_table = nullptr;
    dec->max_pages = dec->num_pages = 0;
    return rc;
}

// slabs of a prefill pass: the engine format's plans up to 192 rows, and -- for projections whose 256-row grid does not fill the chip
// at T rows, which fp16 engines and the fp16 weight images of int4 / packed-only engines may run as 128-row split-K passes
// (linear_f16_nk's time model) -- the fp16 plans at 128 rows
static size_t prefill_slab_floats(const llmie_decoder_config *c, int T) {
    size_t m = engine_slab_floats(c, T < 192 ? T : 192);
    if (T > 192 && c->wfmt != LLMIE_W_FP8) {
        const int H = c->head_num * c->head_size, QKV = (c->head_num + 2 * c->kv_head_num) * c->head_size, I = c->inter_size;
        const int shapes[4][2] = {{H, QKV}, {H, H}, {H, 2 * I}, {I, H}};
        for (const auto &sh : shapes) {
            if (gemm256_fills(T, sh[1])) continue;
            const size_t f = linear_splitk_ws_floats(16, 128, sh[0], sh[1]);
            m = f > m ? f : m;
        }
    }
    return m;
}
static size_t prefill_carve(const llmie_decoder_config *c, int T, int B, size_t *o /*[11]*/) {
    const size_t e = 2, H = static_cast<size_t>(c->head_num) * c->head_size;
    const size_t QKV = static_cast<size_t>(c->head_num + 2 * c->kv_head_num) * c->head_size, I = c->inter_size;
    Carve k;
    o[0] = k.take(static_cast<size_t>(T) * H * e);        // residual
    o[1] = k.take(static_cast<size_t>(T) * QKV * e);      // packed qkv
    o[2] = k.take(static_cast<size_t>(T) * H * e);        // attention output
    o[3] = k.take(static_cast<size_t>(T) * 2 * I * e);    // gate_up
    o[4] = k.take(static_cast<size_t>(T) * I * e);        // act
    o[5] = k.take(static_cast<size_t>(T) * sizeof(int32_t));        // padding offsets (by-product of the prefix kernel)
    o[6] = k.take(static_cast<size_t>(B + 1) * sizeof(int32_t));    // cum_seqlens
    // fp8 engines: per-token e4m3 image + scales of the activation matrix entering each projection
    o[7] = k.take(c->wfmt == LLMIE_W_FP8 ? llmie_linear_fp8_workspace_bytes(T, static_cast<int>(I > H ? I : H), 0) : 256);
    o[8] = k.take(prefill_slab_floats(c, T) * sizeof(float) + 256);   // split-K slabs (short prefills; fp8 passes; mid-size fp16 passes)
    // int8 / int4 engines at prefill-sized T: room for the fp16 image of the largest matrix (projections without an in-kernel
    // de-quantising form: int4, and int8 shapes whose 256-row grid does not fill the chip)
    size_t dq = 0;
    if (c->wfmt == LLMIE_W_INT8 || c->wfmt == LLMIE_W_INT4 || (c->flags & LLMIE_DEC_PACKED_ONLY)) {
        const int bits = c->wfmt == LLMIE_W_INT8 ? 8 : 4;
        const size_t shapes[4][2] = {{H, QKV}, {H, H}, {H, 2 * I}, {I, H}};
        (void)bits;
        for (const auto &sh : shapes) {   // (any T: shapes outside the quantised kernels' K sub-blocks take the image at every size)
            const size_t b = sh[0] * sh[1] * sizeof(half_t);
            dq = b > dq ? b : dq;
        }
    }
    o[9] = k.take(dq + 256);
    o[10] = k.take(static_cast<size_t>(T) * 2 * sizeof(int32_t) + 256);   // (sequence, cache position) of every packed token + QkvRopeArgs
    return k.off;
}

extern "C" size_t llmie_decoder_prefill_workspace_bytes(const llmie_decoder_config *cfg, int max_tokens, int max_batch) {
    if (!config_ok(cfg) || max_tokens <= 0 || max_batch <= 0) return 0;
    size_t o[11];
    return prefill_carve(cfg, max_tokens, max_batch, o);
}

extern "C" int llmie_decoder_prefill(llmie_decoder *dec, const void *hidden_in, void *hidden_out, void *k_cache,
                                     void *v_cache, const int32_t *input_lengths, const int32_t *history_lengths,
                                     int batch, int num_tokens, int max_q_len, void *workspace, size_t workspace_bytes,
                                     llmie_stream stream) {
    LLMIE_REQUIRE(dec && hidden_in && hidden_out && k_cache && v_cache && input_lengths && history_lengths && workspace,
                  "decoder_prefill: NULL pointer");
    const llmie_decoder_config &c = dec->cfg;
    LLMIE_REQUIRE(batch >= 1 && num_tokens >= 1 && max_q_len >= 1 && max_q_len <= num_tokens && max_q_len <= c.max_seq_len,
                  "decoder_prefill: bad shape batch=%d tokens=%d max_q_len=%d", batch, num_tokens, max_q_len);
    LLMIE_REQUIRE(num_tokens <= static_cast<long long>(batch) * max_q_len, "decoder_prefill: num_tokens > batch*max_q_len");
    const bool fp8 = c.wfmt == LLMIE_W_FP8;
    // weight-only int8 / int4 (round 3): the same layer sequence on the quantised matrices -- the projections de-quantise inside
    // the GEMM (int8: gemm8p.cuh WQ form) or through an fp16 image of one matrix at a time (int4; quant_linear.hip)
    const int wqbits = c.wfmt == LLMIE_W_INT8 ? 8 : (c.wfmt == LLMIE_W_INT4 ? 4 : 0);
    if (c.dtype != LLMIE_F16 || (c.wfmt != LLMIE_W_F16 && !fp8 && !wqbits) || c.head_size != 128)
        LLMIE_UNSUPPORTED("decoder_prefill: fp16 activations + fp16 / int8 / int4 / fp8 weights + head_size 128 only (use the per-kernel path)");
    size_t o[11];
    const size_t need = prefill_carve(&c, num_tokens, batch, o);
    if (workspace_bytes < need || reinterpret_cast<uintptr_t>(workspace) % 256) {
        set_error("decoder_prefill: workspace too small or unaligned (%zu < %zu)", workspace_bytes, need);
        return LLMIE_ERR_WORKSPACE;
    }
    char *base = static_cast<char *>(workspace);
    half_t *resid = (half_t *)(base + o[0]), *qkv = (half_t *)(base + o[1]), *attn = (half_t *)(base + o[2]);
    half_t *gu = (half_t *)(base + o[3]), *act = (half_t *)(base + o[4]);
    int32_t *pad = (int32_t *)(base + o[5]), *cum = (int32_t *)(base + o[6]);
    const int H = dec->H, QKV = dec->QKV, I = dec->I, T = num_tokens;
    hipStream_t st = as_stream(stream);
    int rc;
    if (hidden_out != hidden_in) {
        if (hipMemcpyAsync(hidden_out, hidden_in, static_cast<size_t>(T) * H * 2, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            set_error("decoder_prefill: copy failed");
            return LLMIE_ERR_LAUNCH;
        }
    }
    half_t *h = (half_t *)hidden_out;
    void *f8ws = base + o[7];
    const size_t f8ws_bytes = fp8 ? llmie_linear_fp8_workspace_bytes(T, I > H ? I : H, 0) : 0;
    const SlabWs slabs{reinterpret_cast<float *>(base + o[8]), prefill_slab_floats(&c, T)};
    void *deq = base + o[9];
    const size_t deq_bytes = o[10] - o[9];
    QkvRopeArgs *rope_args = (QkvRopeArgs *)(base + o[10]);   // (device copy of the fused QKV epilogue's operands)
    int32_t *tok_b = (int32_t *)(base + o[10] + 256), *tok_tpos = tok_b + T;
    // y = x . W^T (+ residual) in the engine's weight format (fp8: per-token e4m3 activations, fp8 MFMA)
    auto proj = [&](const half_t *x, const llmie_matrix &w, half_t *y, int K, int N, const half_t *residual) -> int {
        if (wqbits)
            return linear_wq(wqbits, x, w.data, (const half_t *)w.scale, y, T, K, N, c.int4_group, EPI_NONE_, nullptr, residual, nullptr,
                             nullptr, 0.f, slabs, st, deq, deq_bytes);
        if (fp8)
            return linear_fp8(x, (const uint8_t *)w.data, (const float *)w.scale, y, T, K, N, nullptr, residual, f8ws, f8ws_bytes,
                              slabs, st);
        return linear_f16_nk(x, (const half_t *)w.data, y, T, K, N, EPI_NONE_, nullptr, residual, slabs, st);
    };
    // context_decoder.cpp:70: exclusive prefix of the lengths (padding offsets are a by-product nobody needs here);
    // the prefix kernel takes [batch, max_q_len] with max_q_len = ceil(T / batch) rows worth of scratch -> use 1 row of T
    if ((rc = llmie_cal_padding_offset(pad, cum, input_lengths, batch, (T + batch - 1) / batch, stream))) return rc;
    // Round 3: RoPE + KV-cache append as the EPILOGUE of the QKV projection (context_attention.cpp:158-205 in one launch sequence;
    // gemm8p.cuh ROPE forms): q is rotated on its way into the packed QKV buffer, k / v go straight to their cache slots and never
    // travel through the buffer; bit-identical to projection + prefill_rope_append_kernel (same arithmetic on the fp16-rounded
    // accumulator).  Prefill-sized T on the eight-phase kernels only; everything else keeps the two launches.
    static const bool rope_fuse_off = getenv("LLMIE_NO_QKV_ROPE_FUSION") != nullptr;
    const bool kv8 = c.kv_fmt == LLMIE_KV_FP8;
    const float ksc = (kv8 && c.k_scale > 0.f) ? c.k_scale : 1.f, vsc = (kv8 && c.v_scale > 0.f) ? c.v_scale : 1.f;
    bool rope_fusable = !rope_fuse_off && T >= kWqPrefillRows && gemm256_fills(T, QKV);
    if (rope_fusable) {
        QkvRopeArgs ra{};
        ra.k_cache = k_cache;
        ra.v_cache = v_cache;
        ra.rope = dec->rope_table;
        ra.table = dec->page_table;
        ra.layer_stride = dec->page_table ? static_cast<size_t>(dec->num_pages) * c.kv_head_num * 128 * c.head_size
                                          : static_cast<size_t>(batch) * c.kv_head_num * c.max_seq_len * c.head_size;
        ra.head_num = c.head_num;
        ra.kv_head_num = c.kv_head_num;
        ra.max_seq_len = c.max_seq_len;
        ra.rotary_dim = c.rotary_dim;
        ra.max_pages = dec->max_pages;
        ra.kv8 = kv8 ? 1 : 0;
        ra.k_inv_scale = 1.0f / ksc;
        ra.v_inv_scale = 1.0f / vsc;
        if ((rc = prefill_token_table(cum, history_lengths, batch, T, tok_b, tok_tpos, ra, rope_args, st))) return rc;
    }
    // kind: 0 = fp16 operands, 1 = e4m3 operands (xs = token scales), 8 = int8 weights
    auto qkv_rope = [&](int l, const llmie_matrix &w, int kind, const void *x, const float *xs, const void *Wd, const void *wsc) -> int {
        gemm256_qkv_rope_launch(kind, x, Wd, qkv, T, QKV, H, xs, static_cast<const float *>(wsc), static_cast<const half_t *>(w.bias), rope_args, l, st);
        return launch_status("decoder_prefill(qkv + rope + append)");
    };
    // the QKV projection of fp16 / int8 / int4 engines; *fused = 1 when its epilogue did RoPE + the cache append
    auto qkv_proj = [&](int l, const llmie_matrix &w, const half_t *x, int *fused) -> int {
        *fused = 0;
        if (rope_fusable && !fp8 && reinterpret_cast<uintptr_t>(w.bias) % 8 == 0) {
            if (!wqbits && gemm256_qkv_rope_eligible(0, T, QKV, H, x, w.data, nullptr, qkv)) {
                *fused = 1;
                return qkv_rope(l, w, 0, x, nullptr, w.data, nullptr);
            }
            if (wqbits == 8 && g8p_w8_eligible(T, H, QKV, x, w.data, w.scale, qkv) && gemm256_qkv_rope_eligible(8, T, QKV, H, x, w.data, w.scale, qkv)) {
                *fused = 1;
                return qkv_rope(l, w, 8, x, nullptr, w.data, w.scale);
            }
            // int4 (and int8 shapes without the in-kernel form): the fp16 image of the matrix, as linear_wq takes it
            if (wqbits && deq_bytes >= static_cast<size_t>(QKV) * H * sizeof(half_t) && H % 8 == 0 && reinterpret_cast<uintptr_t>(w.data) % 8 == 0 &&
                (wqbits == 8 || (c.int4_group % 8 == 0 && H % c.int4_group == 0)) && gemm256_qkv_rope_eligible(0, T, QKV, H, x, deq, nullptr, qkv)) {
                int rc2 = dequantize_weights_f16(wqbits, w.data, static_cast<const half_t *>(w.scale), static_cast<half_t *>(deq), QKV, H, c.int4_group, st);
                if (rc2) return rc2;
                *fused = 1;
                return qkv_rope(l, w, 0, x, nullptr, deq, nullptr);
            }
        }
        return proj(x, w, qkv, H, QKV, nullptr);
    };
    auto attention = [&](int l, const llmie_matrix &wqkv, int fused) -> int {
        return prefill_attention_f16(qkv, (const half_t *)wqkv.bias, k_cache, v_cache, attn, cum, history_lengths, dec->rope_table, l, batch, T,
                                     max_q_len, c.head_num, c.kv_head_num, c.head_size, c.max_seq_len, c.rotary_dim, st, kv8, ksc, vsc,
                                     dec->page_table, dec->max_pages, dec->num_pages, fused);
    };
    // Short prefills (<= 128 tokens, fp16 weights) are weight-stream bound like a decode batch: same launch fusion as the batch
    // decode path -- every projection leaves split-K slabs, the O and down slabs are consumed by the row kernel (reduction +
    // residual stream + the next RMSNorm), the gate/up slabs by the SwiGLU finalize: 9 launches per layer instead of 13.
    if (dec->packed_only) {
        // LLMIE_DEC_PACKED_ONLY: the row-major matrices are gone -- every projection unpacks its tile-packed image (scales applied,
        // one fp16 rounding per weight: the numerics of the int4 prefill) into the workspace and runs the fp16 GEMM on it
        if (fp8) LLMIE_UNSUPPORTED("decoder_prefill: LLMIE_DEC_PACKED_ONLY engines prefill fp16 / int8 / int4 weights only");
        const size_t img_need = static_cast<size_t>(H > QKV ? H : QKV) * (I > H ? I : H);   // (an upper bound is carved: 2 I x H)
        (void)img_need;
        auto pproj = [&](const void *img, const void *scale, int swiglu_img, const half_t *x, half_t *y, int K, int N, int epi,
                         const half_t *residual) -> int {
            if (deq_bytes < static_cast<size_t>(N) * K * sizeof(half_t)) {
                set_error("decoder_prefill: workspace holds no room for the fp16 image of a %d x %d matrix", N, K);
                return LLMIE_ERR_WORKSPACE;
            }
            int rc2 = pk_unpack_f16(dec->pk_wf, img, static_cast<const half_t *>(scale), static_cast<half_t *>(deq), N, K, swiglu_img, st);
            if (rc2) return rc2;
            return linear_f16_nk(x, static_cast<const half_t *>(deq), y, T, K, N, epi, nullptr, residual, slabs, st);
        };
        for (int l = 0; l < c.num_layers; ++l) {
            const llmie_layer_weights &w = dec->layers[l];
            const llmie_decoder::PackedLayer &pw = dec->packed[l];
            TIMED(LLMIE_OP_ATTN_NORM, llmie_rmsnorm(h, resid, w.attn_norm_gamma, c.rms_eps, T, H, LLMIE_F16, stream));
            int fused = 0;
            if (rope_fusable && reinterpret_cast<uintptr_t>(w.qkv.bias) % 8 == 0 && deq_bytes >= static_cast<size_t>(QKV) * H * sizeof(half_t) &&
                gemm256_qkv_rope_eligible(0, T, QKV, H, h, deq, nullptr, qkv)) {
                // (the unpacked fp16 image as the operand of the QKV projection with the RoPE + append epilogue)
                fused = 1;
                TIMED(LLMIE_OP_QKV_GEMM, pk_unpack_f16(dec->pk_wf, pw.qkv, static_cast<const half_t *>(w.qkv.scale), static_cast<half_t *>(deq), QKV, H, 0, st));
                TIMED(LLMIE_OP_QKV_GEMM, qkv_rope(l, w.qkv, 0, h, nullptr, deq, nullptr));
            } else {
                TIMED(LLMIE_OP_QKV_GEMM, pproj(pw.qkv, w.qkv.scale, 0, h, qkv, H, QKV, EPI_NONE_, nullptr));
            }
            TIMED(LLMIE_OP_MHA, attention(l, w.qkv, fused));
            TIMED(LLMIE_OP_O_GEMM, pproj(pw.o, w.o.scale, 0, attn, h, H, H, EPI_NONE_, nullptr));
            TIMED(LLMIE_OP_FFN_NORM, llmie_fused_add_bias_residual_rmsnorm(resid, h, w.o.bias, w.ffn_norm_gamma, c.rms_eps, T, H, LLMIE_F16, stream));
            if (T <= 192 || gemm256_swiglu_fills(T, 2 * I)) {
                TIMED(LLMIE_OP_GATE_UP_SWIGLU, pproj(pw.gate_up, w.gate_up.scale, 1, h, act, H, 2 * I, EPI_SWIGLU_, nullptr));
            } else {
                TIMED(LLMIE_OP_GATE_UP_SWIGLU, pproj(pw.gate_up, w.gate_up.scale, 1, h, gu, H, 2 * I, EPI_NONE_, nullptr));
                TIMED(LLMIE_OP_GATE_UP_SWIGLU, llmie_silu_and_mul(gu, act, T, I, LLMIE_F16, stream));
            }
            TIMED(LLMIE_OP_DOWN_GEMM, pproj(pw.down, w.down.scale, 0, act, h, I, H, EPI_NONE_, resid));
        }
        return LLMIE_OK;
    }
    static const bool short_off = getenv("LLMIE_NO_FUSED_SHORT_PREFILL") != nullptr;
    const int sbits = wqbits ? wqbits : 16;   // split-K kernels' weight-format code
    const bool short_fmt_ok = !fp8 && (wqbits != 4 || (c.int4_group == 128 && T <= 64));   // int4 split-K form: 64 rows, group 128
    if (!short_off && short_fmt_ok && T <= 128 && H % 256 == 0 && I % 256 == 0 && H >= 512 && I >= 512 && splitk_rownorm_eligible(H)) {
        // int8: row scales applied by the slab consumers; int4: group scales applied inside the split-K kernel
        auto sc_of = [&](const llmie_matrix &m) { return SlabScale{wqbits == 8 ? static_cast<const half_t *>(m.scale) : nullptr, nullptr, nullptr}; };
        auto gs_of = [&](const llmie_matrix &m) { return wqbits == 4 ? static_cast<const half_t *>(m.scale) : nullptr; };
        TIMED(LLMIE_OP_ATTN_NORM, llmie_rmsnorm(h, resid, dec->layers[0].attn_norm_gamma, c.rms_eps, T, H, LLMIE_F16, stream));
        for (int l = 0; l < c.num_layers; ++l) {
            const llmie_layer_weights &w = dec->layers[l];
            const bool last = l + 1 == c.num_layers;
            SplitKSlabs sk;
            TIMED(LLMIE_OP_QKV_GEMM, linear_splitk_partial(sbits, h, w.qkv.data, T, H, QKV, st, &sk, slabs, gs_of(w.qkv)));
            // (the slab consumer of the QKV projection does RoPE + the cache append as well: one launch less per layer)
            const bool qfuse = !rope_fuse_off && splitk_finalize_qkv_rope_eligible(sk, c.head_size, qkv, w.qkv.bias);
            if (qfuse) {
                TIMED(LLMIE_OP_QKV_GEMM, splitk_finalize_qkv_rope(sk, sc_of(w.qkv), qkv, (const half_t *)w.qkv.bias, k_cache, v_cache, cum, history_lengths,
                                                                  dec->rope_table, l, batch, c.head_num, c.kv_head_num, c.max_seq_len, c.rotary_dim, st,
                                                                  kv8, ksc, vsc, dec->page_table, dec->max_pages, dec->num_pages));
            } else {
                TIMED(LLMIE_OP_QKV_GEMM, splitk_finalize(sk, sc_of(w.qkv), qkv, EPI_NONE_, nullptr, nullptr, st));
            }
            TIMED(LLMIE_OP_MHA, attention(l, w.qkv, qfuse ? 1 : 0));
            TIMED(LLMIE_OP_O_GEMM, linear_splitk_partial(sbits, attn, w.o.data, T, H, H, st, &sk, slabs, gs_of(w.o)));
            // context_decoder.cpp: h += resid; resid = h; h += o.bias; h = rmsnorm(h, ffn_gamma)
            TIMED(LLMIE_OP_FFN_NORM, splitk_rownorm(sk, sc_of(w.o), static_cast<const half_t *>(w.o.bias), resid,
                                                    static_cast<const half_t *>(w.ffn_norm_gamma), c.rms_eps, h, nullptr, nullptr, st));
            TIMED(LLMIE_OP_GATE_UP_SWIGLU, linear_splitk_partial(sbits, h, w.gate_up.data, T, H, 2 * I, st, &sk, slabs, gs_of(w.gate_up)));
            TIMED(LLMIE_OP_GATE_UP_SWIGLU, splitk_finalize(sk, sc_of(w.gate_up), act, EPI_SWIGLU_, nullptr, nullptr, st));
            TIMED(LLMIE_OP_DOWN_GEMM, linear_splitk_partial(sbits, act, w.down.data, T, I, H, st, &sk, slabs, gs_of(w.down)));
            // h = act.Wd^T + resid, then the next layer's resid = h; h = rmsnorm(h) (last layer: h stays un-normalised)
            const void *next_gamma = last ? nullptr : dec->layers[l + 1].attn_norm_gamma;
            TIMED(LLMIE_OP_ATTN_NORM, splitk_rownorm(sk, sc_of(w.down), nullptr, resid, static_cast<const half_t *>(next_gamma), c.rms_eps, h,
                                                     nullptr, nullptr, st));
        }
        return LLMIE_OK;
    }
    // fp8, prefill-sized: the two RMSNorms emit the e4m3 activations of the projection behind them (norm.hip
    // rmsnorm_quant_kernel; bit-identical to norm + quantize_rows) and the tiled fp8 GEMM takes them as they are
    const bool nq = fp8 && T > 8 && H % 128 == 0 && rmsnorm_quant_eligible(H);
    uint8_t *xqn = static_cast<uint8_t *>(f8ws);
    float *xsn = reinterpret_cast<float *>(xqn + ((static_cast<size_t>(T) * H + 255) & ~static_cast<size_t>(255)));
    auto tiled_fp8 = [&](const llmie_matrix &w, int N) {
        return nq && gemm256_fills(T, N) && N % 4 == 0 && (reinterpret_cast<uintptr_t>(w.data) | reinterpret_cast<uintptr_t>(w.scale)) % 16 == 0;
    };
    // Round 3, fp16 / int8 / int4 weights without an output-projection bias (Llama): the residual stream S lives un-normalised in
    // `h` for the whole pass -- the O and down projections add into it in their epilogues (y = S, residual = S: every element is read
    // and written by the same lane) and the two norms are out-of-place reads of S into `resid` (used as the projections' input N):
    //   N = norm(S) g1 -> qkv -> attention -> S += attn . Wo^T -> N = norm(S) g2 -> act = swiglu(N . Wgu^T) -> S += act . Wd^T
    // Same values as the sequence below up to where fp16 roundings fall (o + resid is rounded once instead of twice); each norm
    // moves 2 x |S| bytes instead of 3-4 x (context_decoder.cpp:70-199 order).
    // (interleaved A/B on one box, fp16: 1 x 2048 78.15k -> 78.55k tok/s, 8 x 512 89.56k -> 89.87k: the norms drop 13.5 + 15.0 ->
    // 9.7 + 9.7 us per layer, the O projection's residual epilogue costs 6.3 us of that back)
    bool lean = !fp8 && rmsnorm_oop_eligible(H);
    for (int l = 0; l < c.num_layers && lean; ++l) lean = dec->layers[l].o.bias == nullptr;
    if (lean) {
        half_t *S = h, *Nn = resid;
        for (int l = 0; l < c.num_layers; ++l) {
            const llmie_layer_weights &w = dec->layers[l];
            TIMED(LLMIE_OP_ATTN_NORM, rmsnorm_oop_f16(S, Nn, (const half_t *)w.attn_norm_gamma, c.rms_eps, T, H, st));
            int fused;
            TIMED(LLMIE_OP_QKV_GEMM, qkv_proj(l, w.qkv, Nn, &fused));
            TIMED(LLMIE_OP_MHA, attention(l, w.qkv, fused));
            TIMED(LLMIE_OP_O_GEMM, proj(attn, w.o, S, H, H, S));
            TIMED(LLMIE_OP_FFN_NORM, rmsnorm_oop_f16(S, Nn, (const half_t *)w.ffn_norm_gamma, c.rms_eps, T, H, st));
            if (wqbits && (T < kWqPrefillRows || gemm256_swiglu_fills(T, 2 * I))) {
                TIMED(LLMIE_OP_GATE_UP_SWIGLU, linear_wq(wqbits, Nn, w.gate_up.data, (const half_t *)w.gate_up.scale, act, T, H, 2 * I, c.int4_group,
                                                         EPI_SWIGLU_, nullptr, nullptr, nullptr, nullptr, 0.f, slabs, st, deq, deq_bytes));
            } else if (!wqbits && (T <= 192 || gemm256_swiglu_fills(T, 2 * I))) {
                TIMED(LLMIE_OP_GATE_UP_SWIGLU, linear_f16_nk(Nn, (const half_t *)w.gate_up.data, act, T, H, 2 * I, EPI_SWIGLU_, nullptr, nullptr, slabs, st));
            } else {
                TIMED(LLMIE_OP_GATE_UP_SWIGLU, proj(Nn, w.gate_up, gu, H, 2 * I, nullptr));
                TIMED(LLMIE_OP_GATE_UP_SWIGLU, llmie_silu_and_mul(gu, act, T, I, LLMIE_F16, stream));
            }
            TIMED(LLMIE_OP_DOWN_GEMM, proj(act, w.down, S, I, H, S));
        }
        return LLMIE_OK;
    }
    for (int l = 0; l < c.num_layers; ++l) {
        const llmie_layer_weights &w = dec->layers[l];
        int fused = 0;
        if (tiled_fp8(w.qkv, QKV)) {
            TIMED(LLMIE_OP_ATTN_NORM, rmsnorm_quant_f16(h, resid, nullptr, (const half_t *)w.attn_norm_gamma, c.rms_eps, T, H, false, xqn, xsn, st));
            if (rope_fusable && reinterpret_cast<uintptr_t>(w.qkv.bias) % 8 == 0 &&
                gemm256_qkv_rope_eligible(1, T, QKV, H, xqn, w.qkv.data, w.qkv.scale, qkv)) {
                fused = 1;
                TIMED(LLMIE_OP_QKV_GEMM, qkv_rope(l, w.qkv, 1, xqn, xsn, w.qkv.data, w.qkv.scale));
            } else {
                TIMED(LLMIE_OP_QKV_GEMM, (gemm256_launch(true, xqn, w.qkv.data, qkv, T, QKV, H, nullptr, nullptr, xsn, (const float *)w.qkv.scale, st),
                                          launch_status("decoder_prefill(qkv fp8)")));
            }
        } else {
            TIMED(LLMIE_OP_ATTN_NORM, llmie_rmsnorm(h, resid, w.attn_norm_gamma, c.rms_eps, T, H, LLMIE_F16, stream));
            TIMED(LLMIE_OP_QKV_GEMM, qkv_proj(l, w.qkv, h, &fused));
        }
        TIMED(LLMIE_OP_MHA, attention(l, w.qkv, fused));
        TIMED(LLMIE_OP_O_GEMM, proj(attn, w.o, h, H, H, nullptr));
        const bool gu_fused8 = fp8 && gemm256_swiglu_fills(T, 2 * I) && H % 128 == 0 && reinterpret_cast<uintptr_t>(w.gate_up.data) % 16 == 0;
        if (nq && gu_fused8 && w.ffn_norm_gamma && reinterpret_cast<uintptr_t>(w.gate_up.scale) % 16 == 0) {
            TIMED(LLMIE_OP_FFN_NORM, rmsnorm_quant_f16(h, resid, (const half_t *)w.o.bias, (const half_t *)w.ffn_norm_gamma, c.rms_eps, T, H, true,
                                                       xqn, xsn, st));
            TIMED(LLMIE_OP_GATE_UP_SWIGLU, (gemm256_swiglu_launch(true, xqn, w.gate_up.data, act, T, 2 * I, H, xsn, (const float *)w.gate_up.scale, st),
                                            launch_status("decoder_prefill(gate_up fp8)")));
            TIMED(LLMIE_OP_DOWN_GEMM, proj(act, w.down, h, I, H, resid));
            continue;
        }
        TIMED(LLMIE_OP_FFN_NORM, llmie_fused_add_bias_residual_rmsnorm(resid, h, w.o.bias, w.ffn_norm_gamma, c.rms_eps, T, H,
                                                                       LLMIE_F16, stream));
        // ffn.cpp:105-122: act = silu(h.Wg^T) * (h.Wu^T); SwiGLU fused into the projection's epilogue where a fused form exists
        if (wqbits && (T < kWqPrefillRows || gemm256_swiglu_fills(T, 2 * I))) {
            TIMED(LLMIE_OP_GATE_UP_SWIGLU, linear_wq(wqbits, h, w.gate_up.data, (const half_t *)w.gate_up.scale, act, T, H, 2 * I, c.int4_group,
                                                     EPI_SWIGLU_, nullptr, nullptr, nullptr, nullptr, 0.f, slabs, st, deq, deq_bytes));
        } else if (!wqbits && !fp8 && (T <= 192 || gemm256_swiglu_fills(T, 2 * I))) {
            TIMED(LLMIE_OP_GATE_UP_SWIGLU, linear_f16_nk(h, (const half_t *)w.gate_up.data, act, T, H, 2 * I, EPI_SWIGLU_, nullptr, nullptr, slabs, st));
        } else if (fp8 && gemm256_swiglu_fills(T, 2 * I) && H % 128 == 0 && reinterpret_cast<uintptr_t>(w.gate_up.data) % 16 == 0) {
            TIMED(LLMIE_OP_GATE_UP_SWIGLU, llmie_linear_fp8_swiglu(h, (const uint8_t *)w.gate_up.data, (const float *)w.gate_up.scale,
                                                                   act, T, H, 2 * I, f8ws, f8ws_bytes, stream));
        } else {
            TIMED(LLMIE_OP_GATE_UP_SWIGLU, proj(h, w.gate_up, gu, H, 2 * I, nullptr));
            TIMED(LLMIE_OP_GATE_UP_SWIGLU, llmie_silu_and_mul(gu, act, T, I, LLMIE_F16, stream));
        }
        TIMED(LLMIE_OP_DOWN_GEMM, proj(act, w.down, h, I, H, resid));
    }
    return LLMIE_OK;
}

extern "C" int llmie_decoder_prefill_paged(llmie_decoder *dec, const void *hidden_in, void *hidden_out, void *k_pool, void *v_pool,
                                           const int32_t *block_table, int max_pages, int num_pages, const int32_t *input_lengths,
                                           const int32_t *history_lengths, int batch, int num_tokens, int max_q_len,
                                           void *workspace, size_t workspace_bytes, llmie_stream stream) {
    LLMIE_REQUIRE(dec && block_table, "decoder_prefill_paged: NULL pointer");
    LLMIE_REQUIRE(max_pages > 0 && num_pages > 0 &&
                      static_cast<long long>(max_pages) * LLMIE_KV_PAGE_TOKENS >= dec->cfg.max_seq_len,
                  "decoder_prefill_paged: max_pages * %d must cover max_seq_len %d", LLMIE_KV_PAGE_TOKENS, dec->cfg.max_seq_len);
    dec->page_table = block_table;
    dec->max_pages = max_pages;
    dec->num_pages = num_pages;
    const int rc = llmie_decoder_prefill(dec, hidden_in, hidden_out, k_pool, v_pool, input_lengths, history_lengths, batch, num_tokens,
                                         max_q_len, workspace, workspace_bytes, stream);
    dec->page_table = nullptr;
    dec->max_pages = dec->num_pages = 0;
    return rc;
}

static int lm_head_sample_impl(llmie_decoder *dec, void *hidden, const void *final_norm_gamma, const llmie_matrix *lm_head,
                               llmie_weight_format lm_fmt, void *logits, int32_t *tmp_ids, void *tmp_vals, int32_t *topk_ids,
                               void *topk_vals, int K, int blocks_per_row, int32_t *seq_len, uint8_t *finished, int32_t *out_ids,
                               int batch, int step, int32_t *step_dev, int end_id, const void *embed_table, void *next_hidden,
                               int advance_step, bool fused_tail, llmie_stream stream) {
    LLMIE_REQUIRE(dec && hidden && final_norm_gamma && lm_head && lm_head->data && logits && topk_ids && topk_vals &&
                      seq_len && finished && out_ids, "lm_head_sample: NULL pointer");
    const llmie_decoder_config &c = dec->cfg;
    LLMIE_REQUIRE(batch >= 1 && batch <= c.max_batch, "lm_head_sample: batch %d outside [1,%d]", batch, c.max_batch);
    LLMIE_REQUIRE(c.vocab_size > 0, "lm_head_sample: vocab_size not set in the decoder config");
    int rc;
    if (lm_fmt == LLMIE_W_F16 && c.dtype == LLMIE_F16 && gemv_f16_eligible(batch, dec->H, hidden, lm_head->data) &&
        !getenv("LLMIE_NO_FUSED_DECODE")) {
        // final RMSNorm fused into the LM-head GEMV prologue (hidden itself is left un-normalised)
        TIMED(LLMIE_OP_LM_HEAD, linear_f16_nk_norm((const half_t *)hidden, (const half_t *)lm_head->data, (half_t *)logits, batch,
                                                   dec->H, c.vocab_size, EPI_NONE_, (const half_t *)lm_head->bias, nullptr,
                                                   (const half_t *)final_norm_gamma, nullptr, c.rms_eps, as_stream(stream)));
    } else {
    // llama.cpp:247  final RMSNorm (the residual copy is unused there: pass NULL)
    TIMED(LLMIE_OP_FINAL_NORM, llmie_rmsnorm(hidden, nullptr, final_norm_gamma, c.rms_eps, batch, dec->H, c.dtype, stream));
    // llama.cpp:282  logits = hidden . lm_head^T
    TIMED(LLMIE_OP_LM_HEAD, engine_linear(dec, lm_fmt, hidden, *lm_head, logits, batch, dec->H, c.vocab_size, false, nullptr,
                                          false, stream));
    }
    if (fused_tail) {
        // llama.cpp:293-318 (+ :219 of the next token): round 1 of the top-k, then ONE launch for round 2, the sampling, the next
        // step's input embedding and the step counter
        LLMIE_REQUIRE(K >= 1 && K <= 32 && K <= c.vocab_size && blocks_per_row >= 1 && blocks_per_row <= 64 && tmp_ids && tmp_vals,
                      "lm_head_sample_next: K=%d / blocks_per_row=%d outside [1, 32] / [1, 64], or tmp buffers missing", K, blocks_per_row);
        TIMED(LLMIE_OP_TOPK, topk_round1_only(logits, tmp_ids, tmp_vals, batch, c.vocab_size, K, blocks_per_row, c.dtype, as_stream(stream)));
        TIMED(LLMIE_OP_SAMPLING, decode_tail(tmp_ids, tmp_vals, topk_ids, topk_vals, K, blocks_per_row, seq_len, finished, out_ids, batch, step,
                                             step_dev, end_id, c.vocab_size, embed_table, next_hidden, dec->H, advance_step, dec->tail_ticket,
                                             c.dtype, as_stream(stream)));
        return LLMIE_OK;
    }
    // llama.cpp:293,304
    TIMED(LLMIE_OP_TOPK, llmie_topk(logits, tmp_ids, tmp_vals, topk_ids, topk_vals, batch, c.vocab_size, K, blocks_per_row,
                                    c.dtype, stream));
    TIMED(LLMIE_OP_SAMPLING, llmie_sampling(topk_ids, topk_vals, seq_len, finished, out_ids, batch, K, step, step_dev, end_id,
                                            c.vocab_size, c.dtype, stream));
    return LLMIE_OK;
}

extern "C" int llmie_lm_head_sample(llmie_decoder *dec, void *hidden, const void *final_norm_gamma,
                                    const llmie_matrix *lm_head, llmie_weight_format lm_fmt, void *logits,
                                    int32_t *tmp_ids, void *tmp_vals, int32_t *topk_ids, void *topk_vals, int K,
                                    int blocks_per_row, int32_t *seq_len, uint8_t *finished, int32_t *out_ids,
                                    int batch, int step, const int32_t *step_dev, int end_id, llmie_stream stream) {
    return lm_head_sample_impl(dec, hidden, final_norm_gamma, lm_head, lm_fmt, logits, tmp_ids, tmp_vals, topk_ids, topk_vals, K, blocks_per_row,
                               seq_len, finished, out_ids, batch, step, const_cast<int32_t *>(step_dev), end_id, nullptr, nullptr, 0, false, stream);
}

extern "C" int llmie_lm_head_sample_next(llmie_decoder *dec, void *hidden, const void *final_norm_gamma, const llmie_matrix *lm_head,
                                         llmie_weight_format lm_fmt, void *logits, int32_t *tmp_ids, void *tmp_vals, int32_t *topk_ids,
                                         void *topk_vals, int K, int blocks_per_row, int32_t *seq_len, uint8_t *finished, int32_t *out_ids,
                                         int batch, int step, int32_t *step_dev, int end_id, const void *embed_table, void *next_hidden,
                                         int advance_step, llmie_stream stream) {
    LLMIE_REQUIRE(!next_hidden || embed_table, "lm_head_sample_next: next_hidden without an embedding table");
    LLMIE_REQUIRE(!advance_step || step_dev, "lm_head_sample_next: advance_step needs the device-resident step");
    LLMIE_REQUIRE(tmp_ids && tmp_vals, "lm_head_sample_next: tmp buffers required");
    return lm_head_sample_impl(dec, hidden, final_norm_gamma, lm_head, lm_fmt, logits, tmp_ids, tmp_vals, topk_ids, topk_vals, K, blocks_per_row,
                               seq_len, finished, out_ids, batch, step, step_dev, end_id, embed_table, next_hidden, advance_step, true, stream);
}

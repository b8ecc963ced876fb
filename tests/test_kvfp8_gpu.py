"""fp8 (e4m3) KV cache (not in the reference, which stores T): cache bytes = e4m3(x / scale) with one static scale
per cache.  Oracle = the reference attention math (oracle/llmie_oracle.c orc_self_decoder) run on the DE-QUANTISED caches,
the e4m3 rounding emulated in numpy; decode (batch <= 8 GEMV path and the batch split-K path), prefill, and the
prefill(n+1)[-1] == prefill(n) -> decode(n+1) property on the quantised caches."""
import numpy as np
import pytest
import torch

import oracle as orc

pytestmark = pytest.mark.gpu
DEV, F16 = "cuda", torch.float16
KS, VS = 1.0 / 32, 1.0 / 16
# non-power-of-two scales: the e4m3 conversion instruction takes an E8M0 scale operand (exponent bits only), so the kernels split
# a scale into its power-of-two part (conversion) and the mantissa remainder (fp32); ADVICE round 2
KS_NP2, VS_NP2 = 0.037, 0.021


def _h(a):
    return a.astype(np.float16).astype(np.float32)


def _e4m3_table():
    vals = []
    for b in range(256):
        s, e, m = b >> 7, (b >> 3) & 0xF, b & 7
        v = np.nan if (e == 15 and m == 7) else ((m / 8.0) * 2.0 ** -6 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 7))
        vals.append(-v if s else v)
    return np.array(vals, np.float32)


TAB = _e4m3_table()


def _to_e4m3(x):
    pos = TAB[:127]
    a = np.minimum(np.abs(x.astype(np.float64)), 448.0)
    idx = np.clip(np.searchsorted(pos, a), 1, 126)
    lo, hi = pos[idx - 1].astype(np.float64), pos[idx].astype(np.float64)
    pick_hi = (a - lo > hi - a) | ((a - lo == hi - a) & (idx % 2 == 0))
    code = np.where(pick_hi, idx, idx - 1).astype(np.uint8)
    code = np.where(a == 0, 0, code).astype(np.uint8)
    return code | (np.signbit(x).astype(np.uint8) << 7)


def _layers(rng, nh, kvh, hs, I, L):
    H, QKV = nh * hs, (nh + 2 * kvh) * hs
    u = lambda shape, s: _h(rng.uniform(-1, 1, shape).astype(np.float32) * s)
    return [dict(attn_norm=_h(u((H,), 0.2) + 1), qkv=u((QKV, H), 2 / np.sqrt(H)), qkv_bias=None, o=u((H, H), 2 / np.sqrt(H)),
                 o_bias=None, ffn_norm=_h(u((H,), 0.2) + 1), gate_up=u((2 * I, H), 2 / np.sqrt(H)), down=u((H, I), 2 / np.sqrt(I)))
            for _ in range(L)]


def _engine(llmie, layers, nh, kvh, hs, I, max_seq, max_batch, kv_fmt, KS=KS, VS=VS):
    d = lambda a: torch.from_numpy(a).to(DEV).to(F16)
    eng = [dict(attn_norm=d(w["attn_norm"]), ffn_norm=d(w["ffn_norm"]), qkv=dict(data=d(w["qkv"])), o=dict(data=d(w["o"])),
                gate_up=dict(data=d(w["gate_up"])), down=dict(data=d(w["down"]))) for w in layers]
    cfg = dict(head_num=nh, kv_head_num=kvh, head_size=hs, inter_size=I, num_layers=len(layers), vocab_size=100,
               max_seq_len=max_seq, max_batch=max_batch, rotary_dim=hs, rotary_base=10000.0, rms_eps=1e-5, dtype=llmie.F16,
               wfmt=llmie.W_F16, int4_group=128, kv_fmt=kv_fmt, k_scale=KS, v_scale=VS)
    return llmie.Decoder(cfg, eng)


@pytest.mark.parametrize("name,nh,kvh,hs,I,L,bs,max_seq,step", [
    ("7Bgeom_b1", 32, 32, 128, 11008, 1, 1, 320, 300), ("7Bgeom_b4_L2", 32, 32, 128, 11008, 2, 4, 320, 300),
    ("gqa4_b2", 16, 4, 128, 1024, 2, 2, 600, 530), ("hs64_b3", 8, 8, 64, 768, 1, 3, 64, 33),
    ("batch20_splitk", 32, 32, 128, 11008, 2, 20, 288, 260)])
@pytest.mark.parametrize("KS,VS", [(KS, VS), (KS_NP2, VS_NP2)], ids=["pow2", "np2"])
def test_decode_with_fp8_kv_cache(llmie, name, nh, kvh, hs, I, L, bs, max_seq, step, KS, VS):
    rng = np.random.default_rng(61)
    H = nh * hs
    layers = _layers(rng, nh, kvh, hs, I, L)
    dec = _engine(llmie, layers, nh, kvh, hs, I, max_seq, bs, llmie.KV_FP8, KS, VS)
    x = _h(rng.standard_normal((bs, H)).astype(np.float32))
    kraw = rng.standard_normal((L, bs, kvh, max_seq, hs)).astype(np.float32) * 0.7
    vraw = rng.standard_normal((L, bs, kvh, max_seq, hs)).astype(np.float32) * 0.7
    kq, vq = _to_e4m3(kraw / KS), _to_e4m3(vraw / VS)
    kc, vc = TAB[kq] * np.float32(KS), TAB[vq] * np.float32(VS)  # what the device attends to
    kd, vd = torch.from_numpy(kq).to(DEV), torch.from_numpy(vq).to(DEV)
    xd = torch.from_numpy(x).to(DEV).to(F16)
    out = dec.forward(xd, torch.empty_like(xd), kd, vd, step)
    ocfg = dict(head_num=nh, kv_head_num=kvh, head_size=hs, inter_size=I, num_layers=L, vocab=100, max_seq_len=max_seq,
                rotary_dim=hs, rotary_base=10000.0, rms_eps=1e-5)
    if L == 1:
        # one layer: the oracle's kernels composed in numpy (self_decoder.cpp:69-119 order) with the device's treatment of the new
        # token -- its k / v rows are quantised to the cache format BEFORE they are attended to -- so short contexts, where that one
        # token carries real softmax weight, are held to the same bound
        w = layers[0]
        hn = _h(orc.rmsnorm(x, w["attn_norm"], 1e-5)[0])
        qkv = _h(orc.rope_decode(_h(orc.linear(hn, w["qkv"])), nh, kvh, hs, step, hs, 10000.0))
        ko, vo = nh * hs, (nh + kvh) * hs
        qkv[:, ko:vo] = TAB[_to_e4m3(qkv[:, ko:vo] / KS)] * np.float32(KS)
        qkv[:, vo:] = TAB[_to_e4m3(qkv[:, vo:] / VS)] * np.float32(VS)
        mha = _h(orc.decoder_mha(qkv, None, kc, vc, 0, nh, kvh, hs, step))
        h1 = _h(_h(orc.linear(mha.reshape(bs, H), w["o"])) + x)
        h2 = _h(orc.rmsnorm(h1, w["ffn_norm"], 1e-5)[0])
        act = _h(orc.silu_and_mul(orc.linear(h2, w["gate_up"]).reshape(bs, 2, I)))
        exp = _h(orc.linear(act.reshape(bs, I), w["down"])) + h1
    else:
        exp = orc.self_decoder(ocfg, layers, x, kc, vc, step)  # appends the new (unquantised) rows into kc / vc
    got = out.float().cpu().numpy()
    err = np.abs(got - exp)
    # (multi-layer cases: the oracle attends to the new token's un-quantised k/v, 1 of `step` tokens, hence their long contexts):
    # within the fp16 decoder tolerance
    assert (err <= 3e-2 + 3e-2 * np.abs(exp)).all(), "max err %g (|exp| max %g)" % (err.max(), np.abs(exp).max())
    # appended rows: e4m3(new k / scale); layer 0 sees identical inputs -> codes equal up to rare rounding ties
    new_k = kd[0, :, :, step - 1].cpu().numpy()
    want = _to_e4m3(_h(kc[0, :, :, step - 1]) / KS)
    assert (new_k != want).mean() < 0.02
    assert np.abs(TAB[new_k] - TAB[want]).max() * KS <= 0.13 * max(1.0, np.abs(kc[0, :, :, step - 1]).max())
    # rows of other positions untouched
    assert np.array_equal(kd[:, :, :, :step - 1].cpu().numpy(), kq[:, :, :, :step - 1])
    dec.close()


@pytest.mark.parametrize("KS,VS", [(KS, VS), (KS_NP2, VS_NP2)], ids=["pow2", "np2"])
def test_prefill_and_decode_agree_on_fp8_kv_cache(llmie, KS, VS):
    rng = np.random.default_rng(62)
    nh, hs, I, L, max_seq, n = 8, 128, 1536, 2, 256, 150
    H = nh * hs
    layers = _layers(rng, nh, nh, hs, I, L)
    d8 = _engine(llmie, layers, nh, nh, hs, I, max_seq, 1, llmie.KV_FP8, KS, VS)
    d16 = _engine(llmie, layers, nh, nh, hs, I, max_seq, 1, llmie.KV_NATIVE)
    xs = torch.from_numpy(_h(rng.standard_normal((n + 1, H)).astype(np.float32))).to(DEV).to(F16)
    z8 = lambda: torch.zeros((L, 1, nh, max_seq, hs), dtype=torch.uint8, device=DEV)
    z16 = lambda: torch.zeros((L, 1, nh, max_seq, hs), dtype=F16, device=DEV)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)
    k1, v1, k2, v2, k3, v3 = z8(), z8(), z8(), z8(), z16(), z16()
    full = d8.prefill(xs, torch.empty_like(xs), k1, v1, i32([n + 1]), i32([0]), n + 1).float()
    d8.prefill(xs[:n].contiguous(), torch.empty((n, H), dtype=F16, device=DEV), k2, v2, i32([n]), i32([0]), n)
    last = d8.forward(xs[n:n + 1].contiguous(), torch.empty((1, H), dtype=F16, device=DEV), k2, v2, n + 1).float()
    a, b = last.cpu().numpy(), full[n:n + 1].cpu().numpy()
    assert (np.abs(a - b) <= 3e-2 + 3e-2 * np.abs(b)).all(), np.abs(a - b).max()
    # both paths wrote the same cache bytes (a token row may differ by one code where its fp16 value sat on a tie)
    mism = (k1 != k2).float().mean().item()
    assert mism < 0.01, mism
    # and the fp8-cache model stays close to the fp16-cache model (e4m3: 3 mantissa bits on K and V)
    ref = d16.prefill(xs, torch.empty_like(xs), k3, v3, i32([n + 1]), i32([0]), n + 1).float()
    rel = ((full - ref).norm() / ref.norm()).item()
    print("fp8-KV prefill vs fp16-KV prefill rel %.4f, cache byte mismatch %.5f" % (rel, mism))
    assert rel < 0.05, rel
    # the cache holds e4m3(k / scale) of what the fp16 engine caches (layer 0: identical inputs)
    deq = torch.from_numpy(TAB).to(DEV)[k1[0].long()] * KS
    assert ((deq - k3[0].float()).abs() <= 0.07 * k3[0].float().abs() + 1e-2).float().mean().item() > 0.999
    d8.close()
    d16.close()


# the three forms of the flash prefill kernel on an e4m3 cache (prefill_attention_f16 picks by grid and by the longest sequence, see
# tests/test_prefill_gpu.py CASES): 64 query rows per workgroup, 4 waves x 2 x 16 rows, 8 waves x 16 rows
FORMS = [("bq64_np2", 8, 8, [150, 60], [30, 0], KS_NP2, VS_NP2),
         ("rt2_ragged_gqa", 16, 4, [256, 130, 200, 256, 129, 20, 255, 140], [0, 7, 0, 64, 0, 3, 100, 0], KS, VS),
         ("w8_long_np2", 16, 16, [1100, 600], [20, 0], KS_NP2, VS_NP2)]


@pytest.mark.parametrize("name,nh,kvh,lens,hist,ks,vs", FORMS, ids=[f[0] for f in FORMS])
def test_fp8_kv_prefill_tracks_fp16_kv_on_every_flash_form(llmie, name, nh, kvh, lens, hist, ks, vs):
    """Ragged prefills WITH history rows on an e4m3 cache against the same engine on an fp16 cache holding the de-quantised history
    (that path is held to the oracle by tests/test_prefill_gpu.py): the outputs differ by the e4m3 rounding of the new rows only,
    history bytes stay as they were, new rows are e4m3(k / scale)."""
    rng = np.random.default_rng(64)
    hs, I, L = 128, 512, 1
    max_seq = -(-max(h + l for h, l in zip(hist, lens)) // 128) * 128
    bs, T, H = len(lens), int(sum(lens)), nh * hs
    layers = _layers(rng, nh, kvh, hs, I, L)
    d8 = _engine(llmie, layers, nh, kvh, hs, I, max_seq, bs, llmie.KV_FP8, ks, vs)
    d16 = _engine(llmie, layers, nh, kvh, hs, I, max_seq, bs, llmie.KV_NATIVE)
    shape = (L, bs, kvh, max_seq, hs)
    codes = lambda: (rng.integers(0, 0x58, shape).astype(np.uint8) | (rng.integers(0, 2, shape).astype(np.uint8) << 7))
    ck, cv = codes(), codes()
    k8, v8 = torch.from_numpy(ck).to(DEV), torch.from_numpy(cv).to(DEV)
    k16, v16 = torch.from_numpy(TAB[ck] * np.float32(ks)).to(DEV).to(F16), torch.from_numpy(TAB[cv] * np.float32(vs)).to(DEV).to(F16)
    xs = torch.from_numpy(_h(rng.standard_normal((T, H)).astype(np.float32))).to(DEV).to(F16)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)
    out8 = d8.prefill(xs, torch.empty_like(xs), k8, v8, i32(lens), i32(hist), max(lens)).float()
    out16 = d16.prefill(xs, torch.empty_like(xs), k16, v16, i32(lens), i32(hist), max(lens)).float()
    assert torch.isfinite(out8).all()
    rel = ((out8 - out16).norm() / out16.norm()).item()
    print("%s: fp8-KV prefill vs fp16-KV prefill rel %.4f" % (name, rel))
    assert rel < 0.05, rel
    tab = torch.from_numpy(TAB).to(DEV)
    for b, (n, h0) in enumerate(zip(lens, hist)):
        # rows outside [history, history + len) keep their bytes
        assert torch.equal(k8[0, b, :, :h0].cpu(), torch.from_numpy(ck[0, b, :, :h0]))
        assert torch.equal(v8[0, b, :, h0 + n:].cpu(), torch.from_numpy(cv[0, b, :, h0 + n:]))
        # appended rows: e4m3(k / scale) of what the fp16 engine caches (same inputs: one layer)
        deq = tab[k8[0, b, :, h0:h0 + n].long()] * ks
        ref = k16[0, b, :, h0:h0 + n].float()
        assert ((deq - ref).abs() <= 0.07 * ref.abs() + 1e-2).float().mean().item() > 0.999
    d8.close()
    d16.close()

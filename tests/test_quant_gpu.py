"""Weight-only int8 / int4 linears and the quantised decoder engine (new work: the reference only plans
quantisation, README.md:36-39 -- no reference oracle exists, so the oracle is an fp32 GEMM over the de-quantised
weights, oracle/llmie_oracle.c orc_linear_w8 / orc_linear_w4, and the quantisers are checked bit-exactly against
their numpy definition)."""
import numpy as np
import pytest
import torch

import oracle as orc
from conftest import systematic_error

pytestmark = pytest.mark.gpu
DEV = "cuda"
F16 = torch.float16


def _h(a):
    return a.astype(np.float16).astype(np.float32)


def _quant8_ref(w):
    amax = np.abs(w).max(axis=1)
    s = (amax / np.float32(127.0)).astype(np.float16)
    s[s == 0] = np.float16(1.0)
    q = np.clip(np.rint(w / s.astype(np.float32)[:, None]), -127, 127).astype(np.int8)
    return q, s


def _quant4_ref(w, group):
    N, K = w.shape
    wg = w.reshape(N, K // group, group)
    amax = np.abs(wg).max(axis=2)
    s = (amax / np.float32(7.0)).astype(np.float16)
    s[s == 0] = np.float16(1.0)
    q = np.clip(np.rint(wg / s.astype(np.float32)[:, :, None]), -8, 7).astype(np.int32).reshape(N, K) + 8
    packed = (q[:, 0::2] | (q[:, 1::2] << 4)).astype(np.uint8)
    return packed, s


@pytest.mark.parametrize("N,K", [(256, 4096), (64, 11008), (5, 128)])
def test_quantizers_bit_exact(llmie, N, K):
    rng = np.random.default_rng(31)
    w = _h(rng.standard_normal((N, K)).astype(np.float32) * 0.05)
    wd = torch.from_numpy(w).to(DEV).to(F16)
    q8 = torch.empty((N, K), dtype=torch.int8, device=DEV)
    s8 = torch.empty(N, dtype=F16, device=DEV)
    llmie.quantize_w8(wd, q8, s8)
    eq, es = _quant8_ref(w)
    assert np.array_equal(s8.cpu().numpy(), es) and np.array_equal(q8.cpu().numpy(), eq)
    group = 128 if K % 128 == 0 else 32
    q4 = torch.empty((N, K // 2), dtype=torch.uint8, device=DEV)
    s4 = torch.empty((N, K // group), dtype=F16, device=DEV)
    llmie.quantize_w4(wd, q4, s4, group)
    eq4, es4 = _quant4_ref(w, group)
    assert np.array_equal(s4.cpu().numpy(), es4) and np.array_equal(q4.cpu().numpy(), eq4)


@pytest.mark.parametrize("M,K,N", [(1, 4096, 512), (2, 4096, 12288), (4, 4096, 1000), (8, 4096, 512), (1, 11008, 4096),
                                   (2, 11008, 256), (3, 11008, 256), (8, 11008, 128), (16, 4096, 512), (32, 4096, 1024),
                                   (33, 11008, 256), (64, 4096, 256), (1, 128, 384)])
def test_linear_w8a16(llmie, M, K, N):
    rng = np.random.default_rng(32)
    w = rng.standard_normal((N, K)).astype(np.float32) / np.sqrt(K)
    q, s = _quant8_ref(_h(w))
    x = _h(rng.standard_normal((M, K)).astype(np.float32))
    y = torch.full((M, N), 9.0, dtype=F16, device=DEV)
    llmie.linear_w8a16(torch.from_numpy(x).to(DEV).to(F16), torch.from_numpy(q).to(DEV), torch.from_numpy(s).to(DEV), y)
    exp = orc.linear_w8(x, q, s.astype(np.float32))
    err = np.abs(y.float().cpu().numpy() - exp)
    assert (err <= 2e-3 + 2e-3 * np.abs(exp)).all(), err.max()


@pytest.mark.parametrize("M,K,N,group", [(11, 4096, 512, 128), (37, 11008, 256, 128), (64, 4096, 1000, 128), (100, 11008, 4096, 128), (20, 4096, 22016, 128), (1, 4096, 512, 128), (2, 4096, 22016, 128), (4, 4096, 256, 128),
                                         (1, 11008, 4096, 128), (2, 11008, 128, 128), (1, 128, 64, 32),
                                         # rows of <= 2 KiB: two consecutive rows per workgroup instruction (odd N, a last group with
                                         # one row, rows shorter than the 128 threads of a half, three tokens)
                                         (1, 4096, 4097, 128), (3, 4096, 1003, 128), (2, 2048, 513, 128), (1, 1024, 77, 64),
                                         (1, 11008, 4097, 128), (1, 5120, 301, 128)])
def test_linear_w4a16(llmie, M, K, N, group):
    rng = np.random.default_rng(33)
    w = rng.standard_normal((N, K)).astype(np.float32) / np.sqrt(K)
    q, s = _quant4_ref(_h(w), group)
    x = _h(rng.standard_normal((M, K)).astype(np.float32))
    y = torch.empty((M, N), dtype=F16, device=DEV)
    llmie.linear_w4a16(torch.from_numpy(x).to(DEV).to(F16), torch.from_numpy(q).to(DEV), torch.from_numpy(s).to(DEV), y, group)
    exp = orc.linear_w4(x, q, s.astype(np.float32), group)
    err = np.abs(y.float().cpu().numpy() - exp)
    assert (err <= 2e-3 + 2e-3 * np.abs(exp)).all(), err.max()


def test_linear_w8_fused_bias_residual(llmie):
    rng = np.random.default_rng(34)
    M, K, N = 20, 4096, 256
    q, s = _quant8_ref(_h(rng.standard_normal((N, K)).astype(np.float32) / 64))
    x, b, r = _h(rng.standard_normal((M, K)).astype(np.float32)), _h(rng.standard_normal(N).astype(np.float32)), \
        _h(rng.standard_normal((M, N)).astype(np.float32))
    y = torch.from_numpy(r).to(DEV).to(F16)
    llmie.linear_w8a16(torch.from_numpy(x).to(DEV).to(F16), torch.from_numpy(q).to(DEV), torch.from_numpy(s).to(DEV), y,
                       bias=torch.from_numpy(b).to(DEV).to(F16), residual=y)
    exp = orc.linear_w8(x, q, s.astype(np.float32)) + b[None, :] + r
    assert np.abs(y.float().cpu().numpy() - exp).max() <= 8e-3


# prefill-sized M (round 3): int8 through the eight-phase GEMM's int8 B-operand form (gemm8p.cuh WQ = 8: plan "42" = 256-wide rounds +
# 128-wide rest, plan 2 = 128-wide, ragged M / N edges, K = 11008), shapes whose grid does not fill the chip and all int4 shapes
# through the fp16 image + fp16 GEMM.  Every output element against a plain torch fp32 GEMM over the de-quantised weights, and 96
# sampled rows against the oracle (orc_linear_w8 / _w4).
@pytest.mark.parametrize("bits,M,K,N", [(8, 2048, 4096, 12288), (8, 4000, 4096, 4096), (8, 2048, 11008, 4096), (8, 2048, 4096, 4100),
                                        (8, 200, 4096, 1024), (8, 300, 512, 33000), (8, 512, 320, 24576), (8, 512, 64, 24576), (8, 2048, 192, 6144),
                                        (4, 2048, 4096, 12288), (4, 4000, 11008, 4096),
                                        (4, 250, 4096, 768)])
def test_linear_wq_at_prefill_rows(llmie, bits, M, K, N):
    rng = np.random.default_rng(36)
    w = _h(rng.standard_normal((N, K)).astype(np.float32) / np.sqrt(K))
    x = _h(rng.standard_normal((M, K)).astype(np.float32))
    xd = torch.from_numpy(x).to(DEV).to(F16)
    y = torch.full((M, N), 9.0, dtype=F16, device=DEV)
    if bits == 8:
        q, sc = _quant8_ref(w)
        deq = q.astype(np.float32) * sc.astype(np.float32)[:, None]
        llmie.linear_w8a16(xd, torch.from_numpy(q).to(DEV), torch.from_numpy(sc).to(DEV), y)
    else:
        q, sc = _quant4_ref(w, 128)
        lo, hi = (q & 0xF).astype(np.float32) - 8, (q >> 4).astype(np.float32) - 8
        vals = np.empty((N, K), np.float32)
        vals[:, 0::2], vals[:, 1::2] = lo, hi
        deq = vals * np.repeat(sc.astype(np.float32), 128, axis=1)
        llmie.linear_w4a16(xd, torch.from_numpy(q).to(DEV), torch.from_numpy(sc).to(DEV), y, 128)
    got = y.float()
    ref = xd.float() @ torch.from_numpy(deq).to(DEV).t()
    err = (got - ref).abs()
    # int8: integer weights, fp32 accumulate, one scale, one rounding -> the fp16 kernel tolerance; int4 multiplies the fp16
    # ROUNDING of code * scale (one more 2^-11 per weight, random): same bound
    assert bool((err <= 2e-3 + 2e-3 * ref.abs()).all()), err.max().item()
    rows = np.unique(np.minimum(np.concatenate([[0, 1, 15, 16, 127, 128, 255, 256, M - 2, M - 1], rng.choice(M, 86, replace=False)]), M - 1))
    exp = orc.linear_w8(x[rows], q, sc.astype(np.float32)) if bits == 8 else orc.linear_w4(x[rows], q, sc.astype(np.float32), 128)
    oerr = np.abs(got[torch.from_numpy(rows).to(DEV)].cpu().numpy() - exp)
    assert (oerr <= 2e-3 + 2e-3 * np.abs(exp)).all(), oerr.max()


def test_linear_w8_at_prefill_rows_fused_bias_residual(llmie):
    rng = np.random.default_rng(37)
    M, K, N = 2048, 4096, 4096
    q, s = _quant8_ref(_h(rng.standard_normal((N, K)).astype(np.float32) / 64))
    x, b, r = _h(rng.standard_normal((M, K)).astype(np.float32)), _h(rng.standard_normal(N).astype(np.float32)), \
        _h(rng.standard_normal((M, N)).astype(np.float32))
    xd, y = torch.from_numpy(x).to(DEV).to(F16), torch.from_numpy(r).to(DEV).to(F16)
    llmie.linear_w8a16(xd, torch.from_numpy(q).to(DEV), torch.from_numpy(s).to(DEV), y, bias=torch.from_numpy(b).to(DEV).to(F16), residual=y)
    deq = torch.from_numpy(q.astype(np.float32) * s.astype(np.float32)[:, None]).to(DEV)
    ref = xd.float() @ deq.t() + torch.from_numpy(b).to(DEV)[None, :] + torch.from_numpy(r).to(DEV)
    assert (y.float() - ref).abs().max().item() <= 8e-3


@pytest.mark.parametrize("fmt,bs", [("int8", 1), ("int8", 4), ("int8", 9), ("int8", 20), ("int8", 32), ("int8", 72), ("int4", 1), ("int4", 2), ("int4", 6), ("int4", 19), ("int4", 40), ("int4", 70)])
def test_quantised_decoder_matches_oracle_on_dequantised_weights(llmie, fmt, bs):
    rng = np.random.default_rng(35)
    nh, hs, I, L, max_seq, step, group = 32, 128, 11008, 1, 96, 40, 128
    H, QKV = nh * hs, 3 * nh * hs

    def mk(n, k):
        w = _h(rng.uniform(-1, 1, (n, k)).astype(np.float32) * 2 / np.sqrt(k))
        if fmt == "int8":
            q, s = _quant8_ref(w)
            deq = q.astype(np.float32) * s.astype(np.float32)[:, None]
        else:
            q, s = _quant4_ref(w, group)
            lo, hi = (q & 0xF).astype(np.float32) - 8, (q >> 4).astype(np.float32) - 8
            vals = np.empty((n, k), np.float32)
            vals[:, 0::2], vals[:, 1::2] = lo, hi
            deq = vals * np.repeat(s.astype(np.float32), group, axis=1)
        return dict(data=torch.from_numpy(q).to(DEV), scale=torch.from_numpy(s).to(DEV)), deq

    mats = {k: mk(n, kk) for k, (n, kk) in dict(qkv=(QKV, H), o=(H, H), gate_up=(2 * I, H), down=(H, I)).items()}
    gam1, gam2 = _h(rng.uniform(0.8, 1.2, H).astype(np.float32)), _h(rng.uniform(0.8, 1.2, H).astype(np.float32))
    layer = dict(attn_norm=torch.from_numpy(gam1).to(DEV).to(F16), ffn_norm=torch.from_numpy(gam2).to(DEV).to(F16),
                 **{k: v[0] for k, v in mats.items()})
    cfg = dict(head_num=nh, kv_head_num=nh, head_size=hs, inter_size=I, num_layers=L, vocab_size=100, max_seq_len=max_seq,
               max_batch=bs, rotary_dim=hs, rotary_base=10000.0, rms_eps=1e-5, dtype=llmie.F16,
               wfmt=llmie.W_INT8 if fmt == "int8" else llmie.W_INT4, int4_group=group)
    dec = llmie.Decoder(cfg, [layer])
    x = _h(rng.standard_normal((bs, H)).astype(np.float32))
    kc = _h(rng.standard_normal((L, bs, nh, max_seq, hs)).astype(np.float32) * 0.5)
    vc = _h(rng.standard_normal((L, bs, nh, max_seq, hs)).astype(np.float32) * 0.5)
    xd = torch.from_numpy(x).to(DEV).to(F16)
    out = torch.empty_like(xd)
    dec.forward(xd, out, torch.from_numpy(kc).to(DEV).to(F16), torch.from_numpy(vc).to(DEV).to(F16), step)
    olayer = dict(attn_norm=gam1, ffn_norm=gam2, qkv_bias=None, o_bias=None, **{k: v[1] for k, v in mats.items()})
    ocfg = dict(head_num=nh, kv_head_num=nh, head_size=hs, inter_size=I, num_layers=L, vocab=100, max_seq_len=max_seq,
                rotary_dim=hs, rotary_base=10000.0, rms_eps=1e-5)
    exp = orc.self_decoder(ocfg, [olayer], x, kc, vc, step)
    err = np.abs(out.float().cpu().numpy() - exp)
    assert (err <= 2e-2 + 2e-2 * np.abs(exp)).all(), err.max()
    # the oracle runs on the DE-QUANTISED weights, so the layer error is the fp16 pipeline's, not the quantiser's: hold it to the
    # same systematic-error bounds as the fp16 decoder (tests/test_decoder_gpu.py)
    fro, proj = systematic_error(out.float().cpu().numpy(), exp)
    assert fro <= 3e-3 and proj <= 2e-4, "relative Frobenius error %.3g, projection on the signal %.3g" % (fro, proj)
    dec.close()


# --------------------------------------------------------------------------- fp8 (OCP e4m3fn)
def _e4m3_table():
    vals = []
    for b in range(256):
        s, e, m = b >> 7, (b >> 3) & 0xF, b & 7
        if e == 15 and m == 7:
            v = np.nan
        elif e == 0:
            v = (m / 8.0) * 2.0 ** -6
        else:
            v = (1 + m / 8.0) * 2.0 ** (e - 7)
        vals.append(-v if s else v)
    return np.array(vals, np.float32)


def _to_e4m3(x):
    """round-to-nearest-even onto the e4m3fn grid, saturating at +-448; returns (bytes, decoded values)"""
    tab = _e4m3_table()
    pos = tab[:127]  # 0 .. 448 ascending (codes 0..126)
    a = np.minimum(np.abs(x.astype(np.float64)), 448.0)
    idx = np.clip(np.searchsorted(pos, a), 1, 126)
    lo, hi = pos[idx - 1].astype(np.float64), pos[idx].astype(np.float64)
    pick_hi = (a - lo > hi - a) | ((a - lo == hi - a) & (idx % 2 == 0))  # tie -> even code
    code = np.where(pick_hi, idx, idx - 1).astype(np.uint8)
    code = np.where(a == 0, 0, code).astype(np.uint8)
    code = code | (np.signbit(x).astype(np.uint8) << 7)
    return code, tab[code]


def test_quantize_fp8_matches_numpy_e4m3(llmie):
    rng = np.random.default_rng(51)
    N, K = 64, 4096
    w = _h(rng.standard_normal((N, K)).astype(np.float32) * 0.05)
    q = torch.empty((N, K), dtype=torch.uint8, device=DEV)
    s = torch.empty(N, dtype=torch.float32, device=DEV)
    llmie.quantize_fp8(torch.from_numpy(w).to(DEV).to(F16), q, s)
    es = (np.abs(w).max(axis=1) / np.float32(448.0)).astype(np.float32)
    assert np.allclose(s.cpu().numpy(), es, rtol=1e-6)
    code, _ = _to_e4m3(w / es[:, None])
    got = q.cpu().numpy()
    mism = (got != code)
    assert mism.mean() < 1e-3, "fp8 codes differ from the numpy e4m3fn rounding in %.4f%% of elements" % (100 * mism.mean())
    tab = _e4m3_table()
    assert np.abs(tab[got] - tab[code]).max() <= 32.0  # a mismatch is at most one ulp at the top binade


@pytest.mark.parametrize("M,K,N", [(1, 4096, 512), (8, 4096, 1024), (32, 4096, 12288), (64, 11008, 256), (100, 4096, 256),
                                   (300, 512, 384), (4096, 256, 3072), (3990, 640, 3140), (4096, 384, 1600),
                                   (128, 4096, 12288), (70, 11008, 4096), (90, 1024, 8194), (2048, 256, 12288),
                                   (64, 4096, 12288), (40, 11008, 4096), (48, 1024, 8194),
                                   (4096, 896, 3072), (4090, 1024, 1664)])   # 7 / 8 k-tiles: the tails of the eight-phase loops
def test_linear_fp8(llmie, M, K, N):
    rng = np.random.default_rng(52)
    w = _h(rng.standard_normal((N, K)).astype(np.float32) / np.sqrt(K))
    x = _h(rng.standard_normal((M, K)).astype(np.float32))
    wd = torch.from_numpy(w).to(DEV).to(F16)
    wq = torch.empty((N, K), dtype=torch.uint8, device=DEV)
    ws = torch.empty(N, dtype=torch.float32, device=DEV)
    llmie.quantize_fp8(wd, wq, ws)
    work = torch.empty(llmie.linear_fp8_workspace_bytes(M, K, N), dtype=torch.uint8, device=DEV)
    y = torch.empty((M, N), dtype=F16, device=DEV)
    llmie.linear_fp8(torch.from_numpy(x).to(DEV).to(F16), wq, ws, y, work)
    # oracle: fp32 GEMM over the de-quantised operands (device weight codes; numpy e4m3 rounding of the activations)
    tab = _e4m3_table()
    wdeq = tab[wq.cpu().numpy()] * ws.cpu().numpy()[:, None]
    xs = (np.abs(x).max(axis=1) / np.float32(448.0)).astype(np.float32)
    _, xdeq = _to_e4m3(x / xs[:, None])
    exp = orc.linear(xdeq * xs[:, None], wdeq)
    err = np.abs(y.float().cpu().numpy() - exp)
    # tolerance: fp16 output rounding + rare 1-ulp differences in the on-device activation rounding (ties)
    assert (err <= 3e-3 + 3e-3 * np.abs(exp)).all(), err.max()
    # and the fp8 result is close to the fp16 GEMM it approximates (e4m3: 3 mantissa bits on both operands)
    ref = orc.linear(x, w)
    assert np.abs(y.float().cpu().numpy() - ref).max() <= 0.12 * np.abs(ref).max()


@pytest.mark.parametrize("M,K,I", [(4096, 256, 3072), (3990, 384, 3100), (3990, 384, 3000)])
def test_linear_fp8_swiglu(llmie, M, K, I):
    rng = np.random.default_rng(55)
    w = _h(rng.standard_normal((2 * I, K)).astype(np.float32) / np.sqrt(K))
    x = _h(rng.standard_normal((M, K)).astype(np.float32))
    wd = torch.from_numpy(w).to(DEV).to(F16)
    wq = torch.empty((2 * I, K), dtype=torch.uint8, device=DEV)
    ws = torch.empty(2 * I, dtype=torch.float32, device=DEV)
    llmie.quantize_fp8(wd, wq, ws)
    work = torch.empty(llmie.linear_fp8_workspace_bytes(M, K), dtype=torch.uint8, device=DEV)
    y = torch.empty((M, I), dtype=F16, device=DEV)
    llmie.linear_fp8_swiglu(torch.from_numpy(x).to(DEV).to(F16), wq, ws, y, work)
    tab = _e4m3_table()
    wdeq = tab[wq.cpu().numpy()] * ws.cpu().numpy()[:, None]
    gu = _h(_fp8_lin(x, wdeq))
    exp = orc.silu_and_mul(gu.reshape(M, 2, I))
    err = np.abs(y.float().cpu().numpy() - exp)
    assert (err <= 4e-3 + 4e-3 * np.abs(exp)).all(), err.max()


def _fp8_lin(x, wdeq):
    """fp8 projection as the engine defines it: per-token e4m3 activations (scale amax/448) x de-quantised e4m3 weights"""
    xs = (np.abs(x).max(axis=1) / np.float32(448.0)).astype(np.float32)
    xs[xs == 0] = 1.0
    _, xdeq = _to_e4m3(x / xs[:, None])
    return orc.linear(xdeq * xs[:, None], wdeq)


@pytest.mark.parametrize("bs", [1, 2, 4, 8, 16, 20, 32, 72, 128])
def test_fp8_decoder_matches_oracle_composition(llmie, bs):
    """one 7B-geometry layer, fp8 weights + per-token fp8 activations, against the oracle's kernels composed in numpy
    with the activation quantisation emulated, at the batch sizes of BASELINE configs[4]'s sweep (1 / 2: GEMV path; 4 / 8 / 16:
    packed e4m3 path up to its upper end; 20 / 32: skinny split-K; 72 / 128: the 128-row split-K form incl. its two-pass edge)"""
    rng = np.random.default_rng(54)
    nh, hs, I, L, max_seq, step = 32, 128, 11008, 1, 64, 33
    H, QKV = nh * hs, 3 * nh * hs
    tab = _e4m3_table()
    mats, deq = {}, {}
    for k, (n, kk) in dict(qkv=(QKV, H), o=(H, H), gate_up=(2 * I, H), down=(H, I)).items():
        w = torch.from_numpy(_h(rng.uniform(-1, 1, (n, kk)).astype(np.float32) * 2 / np.sqrt(kk))).to(DEV).to(F16)
        q = torch.empty((n, kk), dtype=torch.uint8, device=DEV)
        sc = torch.empty(n, dtype=torch.float32, device=DEV)
        llmie.quantize_fp8(w, q, sc)
        mats[k] = dict(data=q, scale=sc)
        deq[k] = tab[q.cpu().numpy()] * sc.cpu().numpy()[:, None]
    g1, g2 = _h(rng.uniform(0.8, 1.2, H).astype(np.float32)), _h(rng.uniform(0.8, 1.2, H).astype(np.float32))
    cfg = dict(head_num=nh, kv_head_num=nh, head_size=hs, inter_size=I, num_layers=L, vocab_size=100, max_seq_len=max_seq,
               max_batch=bs, rotary_dim=hs, rotary_base=10000.0, rms_eps=1e-5, dtype=llmie.F16, wfmt=llmie.W_FP8, int4_group=128)
    dec = llmie.Decoder(cfg, [dict(attn_norm=torch.from_numpy(g1).to(DEV).to(F16), ffn_norm=torch.from_numpy(g2).to(DEV).to(F16), **mats)])
    x = _h(rng.standard_normal((bs, H)).astype(np.float32))
    kc = _h(rng.standard_normal((L, bs, nh, max_seq, hs)).astype(np.float32) * 0.5)
    vc = _h(rng.standard_normal((L, bs, nh, max_seq, hs)).astype(np.float32) * 0.5)
    xd = torch.from_numpy(x).to(DEV).to(F16)
    out = dec.forward(xd, torch.empty_like(xd), torch.from_numpy(kc).to(DEV).to(F16), torch.from_numpy(vc).to(DEV).to(F16), step)
    # oracle composition (self_decoder.cpp:69-119 order), fp16 roundings where the engine stores fp16
    h = _h(orc.rmsnorm(x, g1, 1e-5)[0])
    qkv = _h(_fp8_lin(h, deq["qkv"]))
    qkv = _h(orc.rope_decode(qkv, nh, nh, hs, step, hs, 10000.0))
    mha = _h(orc.decoder_mha(qkv, None, kc, vc, 0, nh, nh, hs, step))
    h1 = _h(_h(_fp8_lin(mha.reshape(bs, H), deq["o"])) + x)
    h2 = _h(orc.rmsnorm(h1, g2, 1e-5)[0])
    gu = _fp8_lin(h2, deq["gate_up"])
    act = _h(orc.silu_and_mul(gu.reshape(bs, 2, I)))
    exp = _h(_fp8_lin(act.reshape(bs, I), deq["down"])) + h1
    got = out.float().cpu().numpy()
    # Per-token dynamic fp8 is chaotic at the element level: an fp16-ulp difference in an activation row (summation
    # order) moves some elements across an e4m3 rounding boundary (a 6% step), and two quantisation stages amplify
    # that to ~2% of the output scale.  The projections themselves are pinned to 3e-3 by test_linear_fp8 on identical
    # inputs; the layer is held to 3% in the Frobenius norm (the fp16 layer it approximates is ~8% away).
    rel = np.linalg.norm(got - exp) / np.linalg.norm(exp)
    print("fp8 layer vs oracle composition: bs %d rel %.4f max %.3f" % (bs, rel, np.abs(got - exp).max()))
    assert rel < 0.03, rel
    assert np.abs(got - exp).max() <= 0.3
    dec.close()


def test_fp8_decoder_runs_and_tracks_fp16(llmie):
    rng = np.random.default_rng(53)
    nh, hs, I, L, max_seq, step, bs = 32, 128, 11008, 1, 64, 20, 4
    H, QKV = nh * hs, 3 * nh * hs
    mats16, mats8 = {}, {}
    for k, (n, kk) in dict(qkv=(QKV, H), o=(H, H), gate_up=(2 * I, H), down=(H, I)).items():
        w = torch.from_numpy(_h(rng.uniform(-1, 1, (n, kk)).astype(np.float32) * 2 / np.sqrt(kk))).to(DEV).to(F16)
        q = torch.empty((n, kk), dtype=torch.uint8, device=DEV)
        s = torch.empty(n, dtype=torch.float32, device=DEV)
        llmie.quantize_fp8(w, q, s)
        mats16[k], mats8[k] = w, dict(data=q, scale=s)
    ones = torch.ones(H, dtype=F16, device=DEV)
    base = dict(head_num=nh, kv_head_num=nh, head_size=hs, inter_size=I, num_layers=L, vocab_size=100, max_seq_len=max_seq,
                max_batch=bs, rotary_dim=hs, rotary_base=10000.0, rms_eps=1e-5, dtype=llmie.F16, int4_group=128)
    d16 = llmie.Decoder(dict(base, wfmt=llmie.W_F16), [dict(attn_norm=ones, ffn_norm=ones, **mats16)])
    d8 = llmie.Decoder(dict(base, wfmt=llmie.W_FP8), [dict(attn_norm=ones, ffn_norm=ones, **mats8)])
    x = torch.randn((bs, H), device=DEV).to(F16)
    kc = (torch.randn((L, bs, nh, max_seq, hs), device=DEV) * 0.5).to(F16)
    vc = (torch.randn((L, bs, nh, max_seq, hs), device=DEV) * 0.5).to(F16)
    o16 = d16.forward(x, torch.empty_like(x), kc.clone(), vc.clone(), step)
    o8 = d8.forward(x, torch.empty_like(x), kc.clone(), vc.clone(), step)
    rel = (o8.float() - o16.float()).norm() / o16.float().norm()
    assert rel.item() < 0.08, rel.item()
    d16.close()
    d8.close()

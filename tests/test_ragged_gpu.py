"""Ragged batch decode (continuous batching; the reference steps a whole batch at one position): per-sequence context
lengths on the device.  The attention op on a ragged batch equals, bit for bit, one batch-1 call per sequence (same
kernels, same chunking) -- dense and paged caches; the engine's ragged step equals, row by row and bit for bit, the uniform
step at that row's position, and three genuinely separate batch-1 engine steps within fp16 tolerance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV, F16 = "cuda", torch.float16


def _rope_table(max_pos, hs, rot, base=10000.0):
    j = np.arange(hs // 2, dtype=np.float32)
    inv = np.power(np.float32(base), (2 * j) / np.float32(rot)).astype(np.float32)
    ang = (np.arange(max_pos, dtype=np.float32)[:, None] / inv[None, :]).astype(np.float32)
    tab = np.stack([np.cos(ang), np.sin(ang)], axis=-1).astype(np.float32)
    return torch.from_numpy(tab).to(DEV)


def _pages(rng, bs, max_seq, L, kvh, hs, kd, vd, ctx_prev, llmie):
    max_pages = (max_seq + 127) // 128
    num_pages = bs * max_pages + 2
    perm = torch.from_numpy(rng.permutation(num_pages)[:bs * max_pages].astype(np.int32)).reshape(bs, max_pages).to(DEV)
    kp = torch.zeros((L, num_pages, kvh, 128, hs), dtype=kd.dtype, device=DEV)
    vp = torch.zeros_like(kp)
    llmie.kv_pages_copy(kd, kp, perm, ctx_prev, True)
    llmie.kv_pages_copy(vd, vp, perm, ctx_prev, True)
    return kp, vp, perm


@pytest.mark.parametrize("nh,kvh,hs,ctx,max_seq,bias", [
    (32, 32, 128, [5, 130, 700], 768, False), (16, 4, 128, [700, 5, 130, 257, 1], 768, True), (8, 8, 64, [33, 600, 128, 129], 640, False)])
def test_ragged_attention_equals_one_call_per_sequence(llmie, nh, kvh, hs, ctx, max_seq, bias):
    rng = np.random.default_rng(91)
    bs, H = len(ctx), nh * hs
    qkv = (torch.randn((bs, nh + 2 * kvh, hs), device=DEV)).to(F16)
    b = (torch.randn(((nh + 2 * kvh) * hs,), device=DEV) * 0.3).to(F16) if bias else None
    kd = (torch.randn((1, bs, kvh, max_seq, hs), device=DEV) * 0.5).to(F16)
    vd = (torch.randn((1, bs, kvh, max_seq, hs), device=DEV) * 0.5).to(F16)
    tab = _rope_table(max_seq, hs, hs)
    ctx_dev = torch.tensor(ctx, dtype=torch.int32, device=DEV)
    ws = torch.empty(llmie.decoder_mha_workspace_bytes(bs, nh, hs, max_seq) // 4, device=DEV)
    # (a) dense ragged call
    k1, v1 = kd.clone(), vd.clone()
    out = torch.zeros((bs, H), dtype=F16, device=DEV)
    llmie.decoder_mha_ragged(qkv, b, k1, v1, out, 0, nh, kvh, ctx_dev, ws, tab, hs, max_seq)
    # (b) one batch-1 call per sequence on that sequence's cache alone
    for i, c in enumerate(ctx):
        ki, vi = kd[:, i:i + 1].clone(), vd[:, i:i + 1].clone()
        oi = torch.zeros((1, H), dtype=F16, device=DEV)
        llmie.decoder_mha_rope(qkv[i:i + 1].contiguous(), b, ki, vi, oi, 0, nh, kvh, c, ws, tab, hs, None)
        assert torch.equal(out[i:i + 1], oi), "sequence %d (context %d)" % (i, c)
        assert torch.equal(k1[:, i:i + 1], ki) and torch.equal(v1[:, i:i + 1], vi)
        # only slot c - 1 of this sequence changed
        changed = (k1[0, i] != kd[0, i]).any(dim=-1).any(dim=0).nonzero().flatten().tolist()
        assert changed in ([c - 1], []), changed
    # (c) paged ragged call: bit-identical to the dense one, appended rows in the right page slots
    ctx_prev = torch.tensor([c - 1 for c in ctx], dtype=torch.int32, device=DEV)
    kp, vp, perm = _pages(rng, bs, max_seq, 1, kvh, hs, kd, vd, ctx_prev, llmie)
    outp = torch.zeros((bs, H), dtype=F16, device=DEV)
    llmie.decoder_mha_ragged(qkv, b, kp, vp, outp, 0, nh, kvh, ctx_dev, ws, tab, hs, max_seq, block_table=perm)
    assert torch.equal(outp, out)
    kback = torch.zeros_like(kd)
    llmie.kv_pages_copy(kback, kp, perm, ctx_dev, False)
    for i, c in enumerate(ctx):
        assert torch.equal(kback[:, i, :, :c], k1[:, i, :, :c])


def test_ragged_attention_ignores_a_sequence_with_an_invalid_length(llmie):
    nh, hs, max_seq, bs = 8, 128, 256, 3
    qkv = torch.randn((bs, 3 * nh, hs), device=DEV).to(F16)
    kd = (torch.randn((1, bs, nh, max_seq, hs), device=DEV) * 0.5).to(F16)
    vd = kd.clone()
    k1, v1 = kd.clone(), vd.clone()
    tab = _rope_table(max_seq, hs, hs)
    ws = torch.empty(llmie.decoder_mha_workspace_bytes(bs, nh, hs, max_seq) // 4, device=DEV)
    out = torch.full((bs, nh * hs), 7.0, dtype=F16, device=DEV)
    llmie.decoder_mha_ragged(qkv, None, k1, v1, out, 0, nh, nh, torch.tensor([0, 100, max_seq + 1], dtype=torch.int32, device=DEV),
                             ws, tab, hs, max_seq)
    assert bool((out[0] == 7.0).all()) and bool((out[2] == 7.0).all()) and not bool((out[1] == 7.0).all())
    assert torch.equal(k1[:, 0], kd[:, 0]) and torch.equal(k1[:, 2], kd[:, 2]) and not torch.equal(k1[:, 1], kd[:, 1])


def _engine(llmie, rng, nh, kvh, hs, I, L, max_seq, bs):
    H, QKV = nh * hs, (nh + 2 * kvh) * hs
    u = lambda shape, s: torch.from_numpy((rng.uniform(-1, 1, shape) * s).astype(np.float32)).to(DEV).to(F16)
    layers = [dict(attn_norm=u((H,), 0.2) + 1, qkv=dict(data=u((QKV, H), 2 / np.sqrt(H)), bias=u((QKV,), 0.1)),
                   o=dict(data=u((H, H), 2 / np.sqrt(H)), bias=u((H,), 0.1)), ffn_norm=u((H,), 0.2) + 1,
                   gate_up=dict(data=u((2 * I, H), 2 / np.sqrt(H))), down=dict(data=u((H, I), 2 / np.sqrt(I)))) for _ in range(L)]
    cfg = dict(head_num=nh, kv_head_num=kvh, head_size=hs, inter_size=I, num_layers=L, vocab_size=100, max_seq_len=max_seq,
               max_batch=bs, rotary_dim=hs, rotary_base=10000.0, rms_eps=1e-5, dtype=llmie.F16, wfmt=llmie.W_F16, int4_group=128)
    return llmie.Decoder(cfg, layers), layers, cfg


@pytest.mark.parametrize("name,nh,kvh,I,L,ctx", [
    ("gemv_b3", 8, 8, 1024, 2, [5, 130, 700]), ("packed_b7_gqa", 16, 4, 1024, 3, [5, 130, 700, 64, 129, 257, 700]),
    ("splitk_b20", 8, 8, 1024, 2, [5, 130, 700, 1] * 5)])
def test_ragged_decoder_step_rows_equal_the_uniform_step_at_their_position(llmie, name, nh, kvh, I, L, ctx):
    rng = np.random.default_rng(92)
    hs, max_seq, bs = 128, 768, len(ctx)
    H = nh * hs
    dec, layers, cfg = _engine(llmie, rng, nh, kvh, hs, I, L, max_seq, bs)
    kd = (torch.randn((L, bs, kvh, max_seq, hs), device=DEV) * 0.5).to(F16)
    vd = (torch.randn((L, bs, kvh, max_seq, hs), device=DEV) * 0.5).to(F16)
    x = torch.randn((bs, H), device=DEV).to(F16)
    ctx_dev = torch.tensor(ctx, dtype=torch.int32, device=DEV)
    k1, v1 = kd.clone(), vd.clone()
    out = dec.forward_ragged(x, torch.empty_like(x), k1, v1, ctx_dev).clone()
    assert bool(torch.isfinite(out.float()).all())
    for c in sorted(set(ctx)):
        k2, v2 = kd.clone(), vd.clone()
        uni = dec.forward(x, torch.empty_like(x), k2, v2, c)
        for i in [i for i, ci in enumerate(ctx) if ci == c]:
            assert torch.equal(out[i], uni[i]), "%s: row %d at context %d" % (name, i, c)
            assert torch.equal(k1[:, i], k2[:, i]) and torch.equal(v1[:, i], v2[:, i])
    # paged ragged step: bit-identical to the dense ragged step
    ctx_prev = torch.tensor([c - 1 for c in ctx], dtype=torch.int32, device=DEV)
    kp, vp, perm = _pages(rng, bs, max_seq, L, kvh, hs, kd, vd, ctx_prev, llmie)
    outp = dec.forward_paged_ragged(x, torch.empty_like(x), kp, vp, perm, ctx_dev)
    assert torch.equal(outp, out)
    kback = torch.zeros_like(kd)
    llmie.kv_pages_copy(kback, kp, perm, ctx_dev, False)
    for i, c in enumerate(ctx):
        assert torch.equal(kback[:, i, :, :c], k1[:, i, :, :c])
    dec.close()
    # three genuinely separate batch-1 engines' steps (different projection kernels: fp16 rounding differs) stay within tolerance
    if bs == 3:
        dec1, _, _ = _engine(llmie, np.random.default_rng(92), nh, kvh, hs, I, L, max_seq, 1)
        for i, c in enumerate(ctx):
            ki, vi = kd[:, i:i + 1].clone(), vd[:, i:i + 1].clone()
            oi = dec1.forward(x[i:i + 1].contiguous(), torch.empty((1, H), dtype=F16, device=DEV), ki, vi, c)
            err = (oi.float() - out[i:i + 1].float()).abs()
            assert bool((err <= 2e-2 + 2e-2 * out[i:i + 1].float().abs()).all()), float(err.max())
        dec1.close()

"""Paged KV cache (not in the reference): the decode engine on 128-token pages addressed through a block table
must be BIT-identical to the same engine on the dense reference layout (which the other tests pin to the oracle), for
fp16 and e4m3 caches, batch <= 8 (GEMV path) and the batch split-K path, shuffled page assignment; llmie_kv_pages_copy
(dense <-> pages) is a bit-exact round trip."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV, F16 = "cuda", torch.float16


def _engine(llmie, rng, nh, kvh, hs, I, L, max_seq, bs, kv8, ks=1 / 32, vs=1 / 16):
    H, QKV = nh * hs, (nh + 2 * kvh) * hs
    u = lambda shape, s: torch.from_numpy((rng.uniform(-1, 1, shape) * s).astype(np.float32)).to(DEV).to(F16)
    layers = [dict(attn_norm=u((H,), 0.2) + 1, qkv=dict(data=u((QKV, H), 2 / np.sqrt(H))), o=dict(data=u((H, H), 2 / np.sqrt(H))),
                   ffn_norm=u((H,), 0.2) + 1, gate_up=dict(data=u((2 * I, H), 2 / np.sqrt(H))), down=dict(data=u((H, I), 2 / np.sqrt(I))))
              for _ in range(L)]
    cfg = dict(head_num=nh, kv_head_num=kvh, head_size=hs, inter_size=I, num_layers=L, vocab_size=100, max_seq_len=max_seq,
               max_batch=bs, rotary_dim=hs, rotary_base=10000.0, rms_eps=1e-5, dtype=llmie.F16, wfmt=llmie.W_F16, int4_group=128,
               kv_fmt=llmie.KV_FP8 if kv8 else llmie.KV_NATIVE, k_scale=ks, v_scale=vs)
    return llmie.Decoder(cfg, layers)


@pytest.mark.parametrize("name,nh,kvh,hs,I,L,bs,max_seq,step,kv8", [
    ("b1", 8, 8, 128, 1024, 2, 1, 600, 530, False), ("b3_gqa", 16, 4, 128, 1024, 2, 3, 384, 300, False),
    ("b20_splitk", 8, 8, 128, 1024, 2, 20, 300, 257, False), ("b2_fp8kv", 8, 8, 128, 1024, 2, 2, 700, 640, True),
    ("b12_fp8kv_splitk", 8, 8, 128, 768, 1, 12, 256, 129, True), ("first_page", 8, 8, 128, 512, 1, 2, 256, 5, False),
    ("b2_fp8kv_np2scales", 8, 8, 128, 1024, 2, 2, 700, 640, True)])
def test_paged_decode_is_bit_identical_to_dense(llmie, name, nh, kvh, hs, I, L, bs, max_seq, step, kv8):
    rng = np.random.default_rng(71)
    ks, vs = (0.037, 0.021) if "np2" in name else (1 / 32, 1 / 16)
    dec = _engine(llmie, rng, nh, kvh, hs, I, L, max_seq, bs, kv8, ks, vs)
    H = nh * hs
    if kv8:
        kd = torch.randint(0, 0x58, (L, bs, kvh, max_seq, hs), device=DEV, dtype=torch.uint8)
        vd = torch.randint(0, 0x58, (L, bs, kvh, max_seq, hs), device=DEV, dtype=torch.uint8)
    else:
        kd = (torch.randn((L, bs, kvh, max_seq, hs), device=DEV) * 0.5).to(F16)
        vd = (torch.randn((L, bs, kvh, max_seq, hs), device=DEV) * 0.5).to(F16)
    max_pages = (max_seq + 127) // 128
    num_pages = bs * max_pages + 3
    perm = torch.from_numpy(rng.permutation(num_pages)[:bs * max_pages].astype(np.int32)).reshape(bs, max_pages).to(DEV)
    kp = torch.zeros((L, num_pages, kvh, 128, hs), dtype=kd.dtype, device=DEV)
    vp = torch.zeros_like(kp)
    ctx = torch.full((bs,), step - 1, dtype=torch.int32, device=DEV)
    llmie.kv_pages_copy(kd, kp, perm, ctx, True)
    llmie.kv_pages_copy(vd, vp, perm, ctx, True)
    x = torch.randn((bs, H), device=DEV).to(F16)
    dense_out = dec.forward(x, torch.empty_like(x), kd, vd, step).clone()
    paged_out = dec.forward_paged(x, torch.empty_like(x), kp, vp, perm, step)
    assert torch.equal(paged_out, dense_out)
    # the appended token landed in the right page slots: gather back and compare with the dense caches
    kback, vback = torch.zeros_like(kd), torch.zeros_like(vd)
    ctx1 = torch.full((bs,), step, dtype=torch.int32, device=DEV)
    llmie.kv_pages_copy(kback, kp, perm, ctx1, False)
    llmie.kv_pages_copy(vback, vp, perm, ctx1, False)
    assert torch.equal(kback[:, :, :, :step], kd[:, :, :, :step]) and torch.equal(vback[:, :, :, :step], vd[:, :, :, :step])
    # pages outside the block table were never touched
    used = torch.zeros(num_pages, dtype=torch.bool, device=DEV)
    used[perm.flatten().long()] = True
    assert int(kp[:, ~used].abs().sum().item() if not kv8 else kp[:, ~used].sum().item()) == 0
    # a device-resident step (graph-friendly form) gives the same result
    step_dev = torch.tensor([step], dtype=torch.int32, device=DEV)
    again = dec.forward_paged(x, torch.empty_like(x), kp, vp, perm, -1, step_dev=step_dev)
    assert torch.equal(again, dense_out)
    dec.close()


@pytest.mark.parametrize("name,nh,kvh,hs,I,L,lens,hist,max_seq,kv8", [
    ("b1_s300", 8, 8, 128, 1024, 2, [300], [0], 384, False), ("ragged_b3_gqa", 16, 4, 128, 1024, 2, [70, 257, 129], [0, 0, 0], 384, False),
    ("history_chunks", 8, 8, 128, 768, 2, [100, 64], [150, 200], 384, False), ("fp8kv_b2", 8, 8, 128, 768, 2, [200, 131], [0, 0], 256, True),
    ("equal_b2_then_decode", 8, 8, 128, 512, 1, [140, 140], [0, 0], 256, False),
    # the cases above run the flash kernel's 64-row form (fewer 128-row query tiles than CUs); these select the other two
    ("rt2_b8_gqa_history", 16, 4, 128, 512, 1, [256, 130, 200, 256, 129, 20, 255, 140], [10, 7, 5, 64, 9, 3, 100, 12], 384, False),
    ("rt2_b8_fp8kv", 16, 4, 128, 512, 1, [256, 250, 129, 256, 200, 131, 256, 140], [0] * 8, 256, True),
    ("w8_long_fp8kv", 16, 16, 128, 512, 1, [1100, 600], [0, 0], 1152, True)])
def test_paged_prefill_is_bit_identical_to_dense(llmie, name, nh, kvh, hs, I, L, lens, hist, max_seq, kv8):
    """llmie_decoder_prefill_paged == llmie_decoder_prefill (hidden states and every cache row), then one decode step on
    the pages it wrote == the dense decode step."""
    rng = np.random.default_rng(72)
    bs, T, H = len(lens), sum(lens), nh * hs
    dec = _engine(llmie, rng, nh, kvh, hs, I, L, max_seq, bs, kv8)
    cdt = torch.uint8 if kv8 else F16
    kd = torch.zeros((L, bs, kvh, max_seq, hs), dtype=cdt, device=DEV)
    vd = torch.zeros_like(kd)
    max_pages = (max_seq + 127) // 128
    num_pages = bs * max_pages + 2
    perm = torch.from_numpy(rng.permutation(num_pages)[:bs * max_pages].astype(np.int32)).reshape(bs, max_pages).to(DEV)
    kp = torch.zeros((L, num_pages, kvh, 128, hs), dtype=cdt, device=DEV)
    vp = torch.zeros_like(kp)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)
    if any(hist):  # history: prefill the earlier chunk on both layouts first (dense call for dense, paged call for pages)
        xh = torch.randn((sum(hist), H), device=DEV).to(F16)
        a = dec.prefill(xh, torch.empty_like(xh), kd, vd, i32(hist), i32([0] * bs), max(hist)).clone()
        b = dec.prefill_paged(xh, torch.empty_like(xh), kp, vp, perm, i32(hist), i32([0] * bs), max(hist))
        assert torch.equal(a, b)
    x = torch.randn((T, H), device=DEV).to(F16)
    dense_out = dec.prefill(x, torch.empty_like(x), kd, vd, i32(lens), i32(hist), max(lens)).clone()
    paged_out = dec.prefill_paged(x, torch.empty_like(x), kp, vp, perm, i32(lens), i32(hist), max(lens))
    assert torch.equal(paged_out, dense_out)
    tot = [a + b for a, b in zip(lens, hist)]
    kback, vback = torch.zeros_like(kd), torch.zeros_like(vd)
    llmie.kv_pages_copy(kback, kp, perm, i32(tot), False)
    llmie.kv_pages_copy(vback, vp, perm, i32(tot), False)
    for b in range(bs):
        assert torch.equal(kback[:, b, :, :tot[b]], kd[:, b, :, :tot[b]]) and torch.equal(vback[:, b, :, :tot[b]], vd[:, b, :, :tot[b]])
    used = torch.zeros(num_pages, dtype=torch.bool, device=DEV)
    used[perm.flatten().long()] = True
    assert int(kp[:, ~used].float().abs().sum().item()) == 0
    if len(set(tot)) == 1:  # the decode engine takes one step index for the batch
        xs = torch.randn((bs, H), device=DEV).to(F16)
        d = dec.forward(xs, torch.empty_like(xs), kd, vd, tot[0] + 1).clone()
        p = dec.forward_paged(xs, torch.empty_like(xs), kp, vp, perm, tot[0] + 1)
        assert torch.equal(d, p)
    dec.close()

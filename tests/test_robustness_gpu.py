"""Failure containment on the device side (round-1 advisor findings): a corrupt device-resident position must not become an
out-of-bounds append, and NaN logits must not become token -1."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV, F16 = "cuda", torch.float16


def _rope_table(max_pos, hs):
    j = np.arange(hs // 2, dtype=np.float32)
    inv = np.power(np.float32(10000.0), (2 * j) / np.float32(hs)).astype(np.float32)
    ang = (np.arange(max_pos, dtype=np.float32)[:, None] / inv[None, :]).astype(np.float32)
    return torch.from_numpy(np.stack([np.cos(ang), np.sin(ang)], axis=-1).astype(np.float32)).to(DEV)


@pytest.mark.parametrize("bad_step", [0, -3, 161, 100000])
def test_device_step_outside_the_cache_touches_nothing(llmie, bad_step):
    """graph-replay form (step read on the device): a value outside [1, max_seq_len] -- e.g. a replay loop that ran one step too
    far -- leaves the caches, the output and the guard bytes around them untouched"""
    nh, hs, max_seq, bs = 8, 128, 160, 2   # max_seq is not a multiple of the 128-token chunk: an append at slot 160 would hit the next head
    qkv = torch.randn((bs, 3 * nh, hs), device=DEV).to(F16)
    guard = 4096
    flat = (torch.randn(guard + bs * nh * max_seq * hs + guard, device=DEV) * 0.5).to(F16)
    vflat = flat.clone()
    k0, v0 = flat.clone(), vflat.clone()
    kc = flat[guard:-guard].view(1, bs, nh, max_seq, hs)
    vc = vflat[guard:-guard].view(1, bs, nh, max_seq, hs)
    out = torch.full((bs, nh * hs), 3.0, dtype=F16, device=DEV)
    ws = torch.empty(llmie.decoder_mha_workspace_bytes(bs, nh, hs, max_seq) // 4, device=DEV)
    step_dev = torch.tensor([bad_step], dtype=torch.int32, device=DEV)
    llmie.decoder_mha_rope(qkv, None, kc, vc, out, 0, nh, nh, -1, ws, _rope_table(max_seq, hs), hs, None, step_dev=step_dev)
    torch.cuda.synchronize()
    assert torch.equal(flat, k0) and torch.equal(vflat, v0)
    assert bool((out == 3.0).all())
    # and a valid step right after still works on the same buffers
    step_dev.fill_(max_seq)
    llmie.decoder_mha_rope(qkv, None, kc, vc, out, 0, nh, nh, -1, ws, _rope_table(max_seq, hs), hs, None, step_dev=step_dev)
    assert not bool((out == 3.0).all()) and bool(torch.isfinite(out.float()).all())
    assert torch.equal(flat[:guard], k0[:guard]) and torch.equal(flat[-guard:], k0[-guard:])


def test_nan_logits_do_not_become_token_minus_one(llmie):
    """top-k never inserts a NaN (nothing is 'better' than it), so a NaN row leaves unfilled candidates (id -1): the sampler skips
    them; a row with no valid candidate ends its sequence (end_id, finished) instead of emitting -1"""
    bs, V, K, end_id = 3, 1000, 4, 2
    logits = torch.randn((bs, V), device=DEV)
    logits[1] = float("nan")                       # a whole row lost to an overflow upstream
    logits[2, : V - 2] = float("nan")              # only two finite candidates left
    tid = torch.empty((bs, 8, K), dtype=torch.int32, device=DEV)
    tv = torch.empty((bs, 8, K), device=DEV)
    ids = torch.empty((bs, K), dtype=torch.int32, device=DEV)
    vals = torch.empty((bs, K), device=DEV)
    llmie.topk(logits, tid, tv, ids, vals)
    assert bool((ids[1] == -1).all()) and int((ids[2] >= 0).sum()) == 2
    seq = torch.zeros(bs, dtype=torch.int32, device=DEV)
    fin = torch.zeros(bs, dtype=torch.uint8, device=DEV)
    out = torch.empty(bs, dtype=torch.int32, device=DEV)
    llmie.sampling(ids, vals, seq, fin, out, 5, end_id, V)
    o = out.cpu().numpy()
    assert 0 <= o[0] < V and fin[0].item() == int(o[0] == end_id)
    assert o[1] == end_id and fin[1].item() == 1
    assert o[2] in (V - 2, V - 1)

"""Nothing on the compute path allocates: every scratch byte of the split-K projections is the caller's (round-1 review:
`splitk_scratch()` grew a library-owned buffer with hipMalloc on first use, which a first call under hipGraph capture cannot do)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_first_call_at_batch_128_is_captured_into_a_graph():
    """fresh process: its first split-K launches (fp16 / int8 / fp8 linear at 128 rows, decoder step at batch 128) are recorded
    into a hipGraph, replayed, and equal the eager launches bit for bit"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "first_call_capture.py")], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["equal"] == [True, True, True, True], out
    assert out["nonzero"] == [True, True, True, True], out
    assert out["kv_equal"], out
    assert out["f16_err"] < 0.05 and out["no_ws_f16_err"] < 0.05, out
    assert "workspace" in out["no_ws_int8"], out       # only a split-K form at 128 rows: says what it needs
    assert "workspace too small" in out["small_ws"], out


def test_workspace_queries_cover_the_plans(llmie):
    # zero where no split-K form exists, positive for decode batches; the fp8 query adds its activation part
    assert llmie.linear_workspace_bytes(llmie.W_F16, 1000, 4096, 4096) == 0
    assert llmie.linear_workspace_bytes(llmie.W_F16, 128, 4096, 12288) >= 128 * 12288 * 4
    assert llmie.linear_workspace_bytes(llmie.W_INT8, 32, 11008, 4096) >= 32 * 4096 * 4
    assert llmie.linear_workspace_bytes(llmie.W_INT4, 32, 4096, 4096) >= 32 * 4096 * 4
    assert llmie.linear_workspace_bytes(llmie.W_F16, 64, 500, 4096) == 0       # K not a multiple of the sub-block
    a = llmie.linear_fp8_workspace_bytes(64, 4096)
    assert a >= 64 * 4096 + 64 * 4
    assert llmie.linear_fp8_workspace_bytes(64, 4096, 4096) >= a + 64 * 4096 * 4


@pytest.mark.parametrize("bs", [32, 128])
def test_lm_head_at_batch_sizes_uses_the_decoder_slab_area(llmie, bs):
    """the LM head ([vocab, H] fp16, the largest N of a step) runs split-K over the decoder's own slab area at decode batches:
    sized at create time for that shape too (a 32000-row vocabulary beside small layers)"""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    DEV, F16 = "cuda", torch.float16
    rng = np.random.default_rng(3)
    nh, hs, I, V, K = 8, 64, 768, 32000, 4
    H = nh * hs
    u = lambda shape, s: torch.from_numpy((rng.uniform(-1, 1, shape) * s).astype(np.float32)).to(DEV).to(F16)
    layers = [dict(attn_norm=u((H,), 0.2) + 1, qkv=dict(data=u((3 * H, H), 0.1)), o=dict(data=u((H, H), 0.1)), ffn_norm=u((H,), 0.2) + 1,
                   gate_up=dict(data=u((2 * I, H), 0.1)), down=dict(data=u((H, I), 0.1)))]
    cfg = dict(head_num=nh, kv_head_num=nh, head_size=hs, inter_size=I, num_layers=1, vocab_size=V, max_seq_len=16, max_batch=bs,
               rotary_dim=hs, rotary_base=10000.0, rms_eps=1e-5, dtype=llmie.F16, wfmt=llmie.W_F16, int4_group=128)
    dec = llmie.Decoder(cfg, layers)
    x, gam, lm = u((bs, H), 1.0), u((H,), 0.1) + 1, u((V, H), 0.2)
    logits = torch.empty((bs, V), dtype=F16, device=DEV)
    tid = torch.empty((bs, 8, K), dtype=torch.int32, device=DEV)
    tv = torch.empty((bs, 8, K), dtype=F16, device=DEV)
    fid = torch.empty((bs, K), dtype=torch.int32, device=DEV)
    fv = torch.empty((bs, K), dtype=F16, device=DEV)
    seq = torch.full((bs,), 7, dtype=torch.int32, device=DEV)
    fin = torch.zeros(bs, dtype=torch.uint8, device=DEV)
    oid = torch.empty(bs, dtype=torch.int32, device=DEV)
    dec.lm_head_sample(x.clone(), gam, lm, llmie.W_F16, logits, tid, tv, fid, fv, seq, fin, oid, step=9, end_id=2)
    xn, _ = orc.rmsnorm(x.float().cpu().numpy(), gam.float().cpu().numpy(), 1e-5)
    exp = orc.linear(xn.astype(np.float16).astype(np.float32), lm.float().cpu().numpy())
    err = np.abs(logits.float().cpu().numpy() - exp)
    assert (err <= 2e-2 + 1e-2 * np.abs(exp)).all(), err.max()
    dec.close()

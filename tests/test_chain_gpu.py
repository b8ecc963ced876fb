"""Persistent chain launch of the packed batch-decode path (round 3, opt-in by LLMIE_CHAIN=1 -- measured slower than the launch
sequence; pk_gemm.cuh pk_chain_kernel): O -> gate/up -> down (-> slab reduce) -> next layer's QKV as phases of
ONE launch per layer with in-kernel grid barriers.  The phases run the launches' own code
(pk_phase), so the chain must reproduce the launch sequence BIT FOR BIT (decoder outputs and KV caches), at Llama-2-7B geometry
(K-split down projection + reduce phase), for every packed weight format, eager and replayed from a hipGraph, with a clean device
status word (no barrier timed out).  The launch sequence itself is pinned to the oracle by tests/test_quant_gpu.py /
tests/test_packed_gpu.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
from tests.conftest import load_llmie
llmie = load_llmie()
DEV, F16 = "cuda", torch.float16
out_path, wfmt = sys.argv[1], sys.argv[2]
rng = np.random.default_rng(23)
nh, hs, I, L, max_seq = 32, 128, 11008, 3, 160
H, QKV = nh * hs, 3 * nh * hs
u = lambda shape, s: torch.from_numpy((rng.uniform(-1, 1, shape) * s).astype(np.float32)).to(DEV).to(F16)
def quantised(w):
    n, k = w.shape
    if wfmt == "f16":
        return dict(data=w)
    if wfmt == "int8":
        q, sc = torch.empty((n, k), dtype=torch.int8, device=DEV), torch.empty(n, dtype=F16, device=DEV)
        llmie.quantize_w8(w, q, sc)
    elif wfmt == "int4":
        q, sc = torch.empty((n, k // 2), dtype=torch.uint8, device=DEV), torch.empty((n, k // 128), dtype=F16, device=DEV)
        llmie.quantize_w4(w, q, sc, 128)
    else:
        q, sc = torch.empty((n, k), dtype=torch.uint8, device=DEV), torch.empty(n, dtype=torch.float32, device=DEV)
        llmie.quantize_fp8(w, q, sc)
    return dict(data=q, scale=sc)
layers = [dict(attn_norm=u((H,), 0.2) + 1, ffn_norm=u((H,), 0.2) + 1, qkv=quantised(u((QKV, H), 2 / np.sqrt(H))), o=quantised(u((H, H), 2 / np.sqrt(H))),
               gate_up=quantised(u((2 * I, H), 2 / np.sqrt(H))), down=quantised(u((H, I), 2 / np.sqrt(I)))) for _ in range(L)]
fmt = {"f16": llmie.W_F16, "int8": llmie.W_INT8, "int4": llmie.W_INT4, "fp8": llmie.W_FP8}[wfmt]
res = {}
for bs in ((9, 16) if wfmt == "fp8" else ((5, 16) if wfmt == "int4" else (5, 17, 32))):
    cfg = dict(head_num=nh, kv_head_num=nh, head_size=hs, inter_size=I, num_layers=L, vocab_size=100, max_seq_len=max_seq, max_batch=bs,
               rotary_dim=hs, rotary_base=10000.0, rms_eps=1e-5, dtype=llmie.F16, wfmt=fmt, int4_group=128)
    dec = llmie.Decoder(cfg, layers)
    g = torch.Generator(device="cpu").manual_seed(bs)
    kc = (torch.randn((L, bs, nh, max_seq, hs), generator=g) * 0.5).to(DEV).to(F16)
    vc = (torch.randn((L, bs, nh, max_seq, hs), generator=g) * 0.5).to(DEV).to(F16)
    x = torch.randn((bs, H), generator=g).to(DEV).to(F16)
    k0, v0 = kc.clone(), vc.clone()
    out = dec.forward(x, torch.empty_like(x), kc, vc, 130)
    dec.status()
    res["out_b%%d" %% bs] = out.float().cpu().numpy()
    res["k_b%%d" %% bs] = kc[:, :, :, 129].float().cpu().numpy()
    # the same step replayed from a graph, three times (the barrier counters re-zero themselves at the end of every launch)
    step_dev = torch.tensor([130], dtype=torch.int32, device=DEV)
    y = torch.empty_like(x)
    s = torch.cuda.Stream()
    kc.copy_(k0); vc.copy_(v0)
    with torch.cuda.stream(s):
        dec.forward(x, y, kc, vc, -1, step_dev=step_dev)
    s.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        dec.forward(x, y, kc, vc, -1, step_dev=step_dev)
    for _ in range(3):
        kc.copy_(k0); vc.copy_(v0); y.zero_()
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, out), "graph replay differs from the eager step (batch %%d)" %% bs
    dec.status()
    dec.close()
np.savez(out_path, **res)
""" % dict(root=ROOT)


def _run(tmp_path, name, wfmt, env_extra):
    out = os.path.join(str(tmp_path), name + ".npz")
    r = subprocess.run([sys.executable, "-c", PROBE, out, wfmt], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(os.environ, **env_extra))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return dict(np.load(out))


@pytest.mark.parametrize("wfmt", ["int8", "f16", "int4", "fp8"])
def test_chain_launch_is_bit_identical_to_the_launch_sequence(tmp_path, wfmt):
    chain = _run(tmp_path, "chain", wfmt, {"LLMIE_CHAIN": "1"})
    seq = _run(tmp_path, "seq", wfmt, {})
    assert set(chain) == set(seq) and chain
    for k in chain:
        assert np.isfinite(chain[k]).all()
        assert np.array_equal(chain[k], seq[k]), "%s %s: chain launch differs from the launch sequence (max diff %g)" % (
            wfmt, k, np.abs(chain[k] - seq[k]).max())

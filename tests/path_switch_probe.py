"""Run by tests/test_path_switches_gpu.py in a fresh process (the switches are read once per process): one decode step at the
given batch and one short prefill through a small 2-layer engine; writes the outputs to argv[1] (.npz)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import load_llmie  # noqa: E402

llmie = load_llmie()
DEV, F16 = "cuda", torch.float16
out_path, wfmt = sys.argv[1], sys.argv[2]
rng = np.random.default_rng(17)
nh, hs, I, L, max_seq = 8, 128, 1024, 2, 384
H, QKV = nh * hs, 3 * nh * hs
u = lambda shape, s: torch.from_numpy((rng.uniform(-1, 1, shape) * s).astype(np.float32)).to(DEV).to(F16)
raw = [dict(attn_norm=u((H,), 0.2) + 1, ffn_norm=u((H,), 0.2) + 1, qkv=u((QKV, H), 2 / np.sqrt(H)), o=u((H, H), 2 / np.sqrt(H)),
            gate_up=u((2 * I, H), 2 / np.sqrt(H)), down=u((H, I), 2 / np.sqrt(I))) for _ in range(L)]


def quantised(w):
    if wfmt == "f16":
        return dict(data=w)
    n, k = w.shape
    if wfmt == "int8":
        q, sc = torch.empty((n, k), dtype=torch.int8, device=DEV), torch.empty(n, dtype=F16, device=DEV)
        llmie.quantize_w8(w, q, sc)
    else:
        q, sc = torch.empty((n, k), dtype=torch.uint8, device=DEV), torch.empty(n, dtype=torch.float32, device=DEV)
        llmie.quantize_fp8(w, q, sc)
    return dict(data=q, scale=sc)


layers = [dict(attn_norm=r["attn_norm"], ffn_norm=r["ffn_norm"], qkv=quantised(r["qkv"]), o=quantised(r["o"]),
               gate_up=quantised(r["gate_up"]), down=quantised(r["down"])) for r in raw]
fmt = {"f16": llmie.W_F16, "int8": llmie.W_INT8, "fp8": llmie.W_FP8}[wfmt]
res = {}
for bs in (2, 20):
    cfg = dict(head_num=nh, kv_head_num=nh, head_size=hs, inter_size=I, num_layers=L, vocab_size=100, max_seq_len=max_seq, max_batch=bs,
               rotary_dim=hs, rotary_base=10000.0, rms_eps=1e-5, dtype=llmie.F16, wfmt=fmt, int4_group=128)
    dec = llmie.Decoder(cfg, layers)
    g = torch.Generator(device="cpu").manual_seed(bs)
    kc = (torch.randn((L, bs, nh, max_seq, hs), generator=g) * 0.5).to(DEV).to(F16)
    vc = (torch.randn((L, bs, nh, max_seq, hs), generator=g) * 0.5).to(DEV).to(F16)
    x = torch.randn((bs, H), generator=g).to(DEV).to(F16)
    res["decode_b%d" % bs] = dec.forward(x, torch.empty_like(x), kc, vc, 300).float().cpu().numpy()
    dec.status()   # (a grid barrier of the persistent chain launch that timed out raises here)
    if bs == 2:
        lens = torch.tensor([70, 40], dtype=torch.int32, device=DEV)
        xs = torch.randn((110, H), generator=g).to(DEV).to(F16)
        kz, vz = torch.zeros_like(kc), torch.zeros_like(vc)
        res["prefill"] = dec.prefill(xs, torch.empty_like(xs), kz, vz, lens, torch.zeros(2, dtype=torch.int32, device=DEV), 70).float().cpu().numpy()
    dec.close()
np.savez(out_path, **res)

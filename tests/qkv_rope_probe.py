"""Run by tests/test_qkv_rope_fusion_gpu.py in a fresh process (LLMIE_NO_QKV_ROPE_FUSION is read once per process): prefill-sized
passes through 2-layer engines in every weight / cache format on both cache layouts; writes the hidden states and every cache byte
to argv[1] (.npz).  Sizes: the fused epilogue runs on the eight-phase kernels, which want a grid of >= 192 workgroups
(tokens / 256 x QKV columns / 128), so the cases are 1-2k tokens on 16-head models."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import load_llmie  # noqa: E402

llmie = load_llmie()
DEV, F16 = "cuda", torch.float16
out_path = sys.argv[1]
only = sys.argv[2] if len(sys.argv) > 2 else ""   # (debugging aid: one case)
res = {}

# name, weight format, heads, kv heads, inter, lengths, histories, max_seq, e4m3 cache, paged, qkv bias, rotary_dim
CASES = [
    ("f16_b1_7b_heads", "f16", 32, 32, 1024, [2048], [0], 2048, False, False, False, 128),          # 256-wide + 128-wide launches (the bench's plan)
    ("f16_ragged_gqa_bias", "f16", 32, 8, 1024, [700, 257, 129], [0, 0, 0], 768, False, False, True, 128),
    ("f16_history_paged", "f16", 16, 16, 768, [1200, 848], [150, 200], 1408, False, True, False, 128),   # all 256-wide tiles
    ("f16_kv8_np2", "f16", 16, 16, 768, [1000, 131], [0, 17], 1024, True, False, True, 64),          # e4m3 cache, partial rotary
    ("f16_kv8_paged", "f16", 16, 16, 768, [900, 300], [0, 0], 1024, True, True, False, 128),
    ("int8_b2", "int8", 16, 16, 1024, [800, 333], [0, 0], 896, False, False, True, 128),
    ("int4_b1", "int4", 16, 16, 1024, [1025], [0], 1152, False, False, False, 128),
    ("fp8_b2", "fp8", 16, 16, 1024, [640, 512], [0, 64], 768, False, False, True, 128),
    ("fp8_kv8_paged", "fp8", 16, 16, 1024, [1200], [0], 1280, True, True, False, 128),
    ("po_int8_b2", "int8", 16, 16, 1024, [700, 420], [0, 0], 768, False, False, True, 128),        # LLMIE_DEC_PACKED_ONLY: unpacked image + fused epilogue
    # short prefills (<= 128 tokens): the QKV projection's split-K slab consumer does the RoPE + append (splitk_finalize_qkv_rope)
    ("short_f16_ragged_bias", "f16", 16, 16, 1024, [70, 40, 18], [0, 5, 0], 256, False, False, True, 128),
    ("short_f16_gqa_paged_kv8", "f16", 16, 4, 1024, [128], [100], 384, True, True, False, 64),
    ("short_int8_b2", "int8", 16, 16, 1024, [64, 57], [0, 0], 128, False, False, True, 128),
    ("short_int4_b1", "int4", 16, 16, 1024, [50], [3], 128, False, False, False, 128),
]


def quantised(w, wfmt):
    if wfmt == "f16":
        return dict(data=w)
    n, k = w.shape
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


for name, wfmt, nh, kvh, I, lens, hist, max_seq, kv8, paged, bias, rot in CASES:
    if only and name != only:
        continue
    rng = np.random.default_rng(sum(map(ord, name)))
    hs, L = 128, 2
    H, QKV, bs, T = nh * hs, (nh + 2 * kvh) * hs, len(lens), sum(lens)
    u = lambda shape, s: torch.from_numpy((rng.uniform(-1, 1, shape) * s).astype(np.float32)).to(DEV).to(F16)
    layers = []
    for _ in range(L):
        qkv = quantised(u((QKV, H), 2 / np.sqrt(H)), wfmt)
        if bias:
            qkv["bias"] = u((QKV,), 0.3)
        layers.append(dict(attn_norm=u((H,), 0.2) + 1, ffn_norm=u((H,), 0.2) + 1, qkv=qkv, o=quantised(u((H, H), 2 / np.sqrt(H)), wfmt),
                           gate_up=quantised(u((2 * I, H), 2 / np.sqrt(H)), wfmt), down=quantised(u((H, I), 2 / np.sqrt(I)), wfmt)))
    cfg = dict(head_num=nh, kv_head_num=kvh, head_size=hs, inter_size=I, num_layers=L, vocab_size=100, max_seq_len=max_seq, max_batch=bs,
               rotary_dim=rot, rotary_base=10000.0, rms_eps=1e-5, dtype=llmie.F16,
               wfmt={"f16": llmie.W_F16, "int8": llmie.W_INT8, "int4": llmie.W_INT4, "fp8": llmie.W_FP8}[wfmt], int4_group=128,
               kv_fmt=llmie.KV_FP8 if kv8 else llmie.KV_NATIVE, k_scale=0.037 if "np2" in name else 1 / 32, v_scale=0.021 if "np2" in name else 1 / 16,
               flags=llmie.DEC_PACKED_ONLY if name.startswith("po_") else 0)
    dec = llmie.Decoder(cfg, layers)
    g = torch.Generator(device="cpu").manual_seed(len(name))
    x = torch.randn((T, H), generator=g).to(DEV).to(F16)
    cdt = torch.uint8 if kv8 else F16
    lens_d = torch.tensor(lens, dtype=torch.int32, device=DEV)
    hist_d = torch.tensor(hist, dtype=torch.int32, device=DEV)
    if paged:
        max_pages = (max_seq + 127) // 128
        num_pages = bs * max_pages + 2
        perm = torch.from_numpy(rng.permutation(num_pages)[:bs * max_pages].astype(np.int32)).reshape(bs, max_pages).to(DEV)
        shape = (L, num_pages, kvh, 128, hs)
    else:
        shape = (L, bs, kvh, max_seq, hs)
    # history rows hold data the pass must read (and leave alone); everything else starts as a marker the pass must leave where it
    # does not write
    if kv8:
        kc = torch.randint(0, 0x58, shape, generator=g, dtype=torch.uint8).to(DEV)
        vc = torch.randint(0, 0x58, shape, generator=g, dtype=torch.uint8).to(DEV)
    else:
        kc = (torch.randn(shape, generator=g) * 0.5).to(DEV).to(F16)
        vc = (torch.randn(shape, generator=g) * 0.5).to(DEV).to(F16)
    out = torch.empty_like(x)
    if paged:
        dec.prefill_paged(x, out, kc, vc, perm, lens_d, hist_d, max(lens))
    else:
        dec.prefill(x, out, kc, vc, lens_d, hist_d, max(lens))
    torch.cuda.synchronize()
    res[name + "/hidden"] = out.view(torch.int16).cpu().numpy()
    res[name + "/k"] = (kc if kv8 else kc.view(torch.int16)).cpu().numpy()
    res[name + "/v"] = (vc if kv8 else vc.view(torch.int16)).cpu().numpy()
    dec.close()
np.savez(out_path, **res)

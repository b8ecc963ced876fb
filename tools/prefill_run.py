#!/usr/bin/env python3
"""Runs a few prefill passes (development tool, for rocprofv3): python tools/prefill_run.py BATCH SEQ [LAYERS] [fp8]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

b, s = int(sys.argv[1]), int(sys.argv[2])
cfg = dict(bench.LLAMA2_7B)
cfg["num_layers"] = int(sys.argv[3]) if len(sys.argv) > 3 else 2
wfmt = "fp8" if "fp8" in sys.argv else "f16"
llmie = bench.load_llmie()
weights = bench.build_weights(torch, cfg, 1234)
layers = weights["layers"] if wfmt == "f16" else bench.quantize_layers(torch, llmie, weights["layers"], wfmt)
dec, kc, vc = bench.make_decoder(torch, llmie, cfg, weights, layers, wfmt, b, s)
H, T = cfg["head_num"] * cfg["head_size"], b * s
hid = torch.randn((T, H), device="cuda").half()
out = torch.empty_like(hid)
lens = torch.full((b,), s, dtype=torch.int32, device="cuda")
hist = torch.zeros(b, dtype=torch.int32, device="cuda")
for _ in range(2):
    dec.prefill(hid, out, kc, vc, lens, hist, s)
torch.cuda.synchronize()
P = 4
dec.profile_begin(P * (cfg["num_layers"] * 12 + 8))
for _ in range(P):
    dec.prefill(hid, out, kc, vc, lens, hist, s)
for op, (ms, n) in dec.profile_end().items():
    if n:
        print("%-16s %9.1f us/launch  x%3d per pass" % (op, ms / n * 1e3, n // P))

#!/usr/bin/env python3
"""Per-op decode breakdown (development tool): python tools/opprofile.py WFMT BATCH CTX [LAYERS] [kvfp8]
Uses the engine's hipEvent op timers (llmie_decoder_profile_begin/end) on eager launches of full decode steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

wfmt, B, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = dict(bench.LLAMA2_7B)
if len(sys.argv) > 4:
    cfg["num_layers"] = int(sys.argv[4])
kv8 = len(sys.argv) > 5 and sys.argv[5] == "kvfp8"
llmie = bench.load_llmie()
weights = bench.build_weights(torch, cfg, 1234)
layers = weights["layers"] if wfmt == "f16" else bench.quantize_layers(torch, llmie, weights["layers"], wfmt)
dec, kc, vc = bench.make_decoder(torch, llmie, cfg, weights, layers, wfmt, B, S, kv8)
H = cfg["head_num"] * cfg["head_size"]
hidden = torch.randn((B, H), device="cuda").half()
out = torch.empty_like(hidden)
P = 4
step_dev = torch.tensor([S - P], dtype=torch.int32, device="cuda")
for _ in range(2):
    dec.forward(hidden, out, kc, vc, -1, step_dev=step_dev)
torch.cuda.synchronize()
dec.profile_begin(P * (cfg["num_layers"] * 12 + 8))
for _ in range(P):
    dec.forward(hidden, out, kc, vc, -1, step_dev=step_dev)
    llmie.advance_step(step_dev)
prof = dec.profile_end()
tot = 0.0
for op, (ms, n) in prof.items():
    if n:
        print("%-16s %8.2f us/launch  x%3d/step  %9.1f us/step" % (op, ms / n * 1e3, n // P, ms / P * 1e3))
        tot += ms / P * 1e3
wb = {"f16": 2.0, "int8": 1.0, "fp8": 1.0, "int4": 0.5 + 2.0 / 128}[wfmt]
nbytes = bench.decode_bytes_per_step(cfg, B, S, wb, 1 if kv8 else 2)
print("total %.1f us/step (sum of timed ops, eager)  algorithmic %.3f GB -> %.2f TB/s" % (tot, nbytes / 1e9, nbytes / tot / 1e6))

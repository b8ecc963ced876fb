"""dev: phase-edge timestamps of the persistent chain launch (int8, batch 32, 7B geometry, 2 layers; last layer's chain has no QKV slot,
so run 3 layers and read the stamps of... every layer overwrites: we arm only around a 1-layer-visible window by using L layers and
reading after the step -> the LAST layer's chain (O, gate/up, down, reduce).  For the full 5-slot chain use L>=2 and LLMIE_STAMP_LAYER."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
llmie = bench.load_llmie()
cfg = dict(bench.LLAMA2_7B)
cfg["num_layers"] = 2
weights = bench.build_weights(torch, cfg, seed=1)
q8 = bench.quantize_layers(torch, llmie, weights["layers"], "int8")
B, S = 32, 128
dec, kc, vc = bench.make_decoder(torch, llmie, cfg, weights, q8, "int8", B, S)
H = 4096
x = torch.randn((B, H), device="cuda").half()
stamps = torch.zeros((256, 16), dtype=torch.int64, device="cuda")
junk = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
for rep in range(3):
    junk.fill_(rep)
    dec.debug_stamps(stamps)
    dec.forward(x, torch.empty_like(x), kc, vc, 100)
    dec.status()
    st = stamps.cpu().numpy().astype(np.int64)
    n = int((st[0] != 0).sum())
    t0 = st[:, 0].min()
    rel = (st[:, :n] - t0) * 0.01
    names = ["start", "O done", "bar1 out", "gate/up done", "bar2 out", "down done", "bar3 out", "reduce done", "bar4 out", "end"]
    print("rep %d: %d stamps; median [min .. max] us over the 256 workgroups" % (rep, n))
    for i in range(n):
        v = rel[:, i]
        print("  %-14s %7.2f  [%7.2f .. %7.2f]" % (names[i] if i < len(names) else str(i), np.median(v), v.min(), v.max()))

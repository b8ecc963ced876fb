"""A/B: decode attention (split + merge launches) with the position as a host argument against the device-resident position:
what the dependent load of step_dev costs per launch.  Interleaved blocks of launches, hipEvents, an L2-sized scrub in between the
blocks is NOT used (the engine's own launches follow each other back to back as well)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.conftest import load_llmie
llmie = load_llmie()
dev = "cuda"
nh, hs, S, L = 32, 128, 2048, 4
qkv = torch.randn((1, 3 * nh, hs), device=dev).half()
kc = torch.randn((L, 1, nh, S + 128, hs), device=dev).half()
vc = torch.randn((L, 1, nh, S + 128, hs), device=dev).half()
out = torch.empty((1, nh * hs), device=dev).half()
ws = torch.empty(llmie.decoder_mha_workspace_bytes(1, nh, hs, S + 128), dtype=torch.uint8, device=dev)
step = S + 1
sd = torch.tensor([step], dtype=torch.int32, device=dev)
junk = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
def run(n, use_dev, scrub):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for i in range(n):
        if scrub:
            junk.add_(1)   # other traffic between the launches (as the projections are in the engine)
        e0.record()
        llmie.decoder_mha_rope(qkv, None, kc, vc, out, i % L, nh, nh, -1 if use_dev else step, ws, None, 0, None, step_dev=sd if use_dev else None)
        e1.record()
        e1.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3
for scrub in (False, True):
    for rep in range(3):
        a = run(100, False, scrub)
        b = run(100, True, scrub)
        print("scrub=%d  host step %.2f us   device step %.2f us   (split + merge, hipEvents around one call)" % (scrub, a, b))

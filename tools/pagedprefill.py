#!/usr/bin/env python3
"""Development tool: dense vs paged prefill of one sequence (Llama-2-7B geometry, a few layers) -- LLMIE_LIB selects another build."""
import importlib.util, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
llmie = bench.load_llmie()
if os.environ.get("LLMIE_LIB"):
    llmie.LIB_PATH = os.environ["LLMIE_LIB"]
cfg = dict(bench.LLAMA2_7B, num_layers=4)
w = bench.build_weights(torch, cfg, 0)
L, kvh, hs, H = 4, 32, 128, 4096
for T in (2048, 512):
    dec, kc, vc = bench.make_decoder(torch, llmie, cfg, w, w["layers"], "f16", 1, T)
    x = torch.randn((T, H), device="cuda").half()
    y = torch.empty_like(x)
    lens = torch.tensor([T], dtype=torch.int32, device="cuda"); hist = torch.zeros(1, dtype=torch.int32, device="cuda")
    pages = (T + 127) // 128
    kp = torch.zeros((L, pages + 2, kvh, 128, hs), dtype=torch.float16, device="cuda"); vp = torch.zeros_like(kp)
    table = torch.from_numpy(np.random.default_rng(0).permutation(pages + 2)[:pages].astype(np.int32)).reshape(1, pages).cuda()

    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    d = sorted(timed(lambda: dec.prefill(x, y, kc, vc, lens, hist, T)) for _ in range(3))[1]
    p = sorted(timed(lambda: dec.prefill_paged(x, y, kp, vp, table, lens, hist, T)) for _ in range(3))[1]
    print("%s T=%d: dense %.3f ms  paged %.3f ms (4 layers)" % (os.environ.get("LLMIE_LIB", "default")[-12:], T, d, p), flush=True)
    dec.close()

#!/usr/bin/env python3
"""Fixed vs per-tile cost of the packed linear (development tool): N = 4096 j at K = 4096 -> j tiles per workgroup."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
llmie = bench.load_llmie()
lib = llmie.lib()
dev = "cuda"
X32 = int(os.environ.get("PK_X32", "0"))   # x32-layout flags of the timed calls (buffers always hold 32 rows)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
fmt = llmie.W_INT8
K = 4096


def timeit(fn, n, reps=5):
    best = 1e9
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(i)
        e1.record(); e1.synchronize()
        if r: best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


x = torch.randn((32, K), device=dev).half()
gamma = torch.ones(K, device=dev).half()
for mode in ("plain", "norm", "resid"):
    for j in (1, 2, 3, 4, 6, 8):
        N = 4096 * j
        NS = max(2, 600 // (N * K // 1000000))
        sets = []
        for _ in range(NS):
            w = torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev)
            sc = torch.full((N,), 0.01, dtype=torch.float16, device=dev)
            p, _ = llmie.pack_weight(fmt, w, sc, False)
            sets.append((p, sc))
            del w
        y = torch.zeros((32, N), device=dev, dtype=torch.float16)
        st = llmie._st()

        def call(i):
            p, s = sets[i % NS]
            rc = lib.llmie_linear_packed(fmt, llmie._p(x), llmie._p(p), llmie._p(s), llmie._p(y), M, K, N, 0, X32,
                                         llmie._p(y) if mode == "resid" else None, llmie._p(gamma) if mode == "norm" else None, None, 1e-5, None, 0, st)
            assert rc == 0, lib.llmie_last_error()
        t = timeit(call, 2 * NS)
        print("M=%d %-5s tiles/WG=%d  %7.2f us  %5.2f TB/s" % (M, mode, j, t, N * K / t / 1e6), flush=True)
        del sets
        torch.cuda.empty_cache()

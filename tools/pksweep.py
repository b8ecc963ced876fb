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
FMT = sys.argv[2] if len(sys.argv) > 2 else "int8"
fmt = {"int8": llmie.W_INT8, "int4": llmie.W_INT4, "f16": llmie.W_F16}[FMT]
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
            if FMT == "int4":
                w = torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev)
                sc = torch.full((N, K // 128), 0.01, dtype=torch.float16, device=dev)
                p, ps = llmie.pack_weight(fmt, w, sc, False)
                sets.append((p, ps))
            elif FMT == "f16":
                w = (torch.randn((N, K), device=dev) * 0.02).half()
                p, _ = llmie.pack_weight(fmt, w, None, False)
                sets.append((p, None))
            else:
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
        print("%s M=%d %-5s tiles/WG=%d  %7.2f us  %5.2f TB/s" % (FMT, M, mode, j, t, N * K * {"int8": 1, "int4": 0.5, "f16": 2}[FMT] / t / 1e6), flush=True)
        del sets
        torch.cuda.empty_cache()

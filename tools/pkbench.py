#!/usr/bin/env python3
"""Micro-benchmark of the packed-weight batch-decode linear (development tool): python tools/pkbench.py [fmt ...] [M=32]
Cycles over NSETS weight copies so every launch streams from HBM; prints us per launch (hipEvents, incl. launch gap) and
effective TB/s of the weight stream.  Run under rocprofv3 --kernel-trace for kernel-only durations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
llmie = bench.load_llmie()
lib = llmie.lib()
dev = "cuda"
X32 = int(os.environ.get("PK_X32", "0"))   # x32-layout flags of the timed calls (buffers always hold 32 rows)
fmts = [a for a in sys.argv[1:] if not a.startswith("M=")] or ["int8"]
Ms = [int(a[2:]) for a in sys.argv[1:] if a.startswith("M=")] or [32]
NSETS = 8
code = {"f16": llmie.W_F16, "int8": llmie.W_INT8, "int4": llmie.W_INT4, "fp8": llmie.W_FP8}
bpe = {"f16": 2.0, "int8": 1.0, "int4": 0.5, "fp8": 1.0}
shapes = [("qkv", 12288, 4096, False, True), ("o", 4096, 4096, False, False), ("gate_up", 22016, 4096, True, True), ("down", 4096, 11008, False, False)]


def timeit(fn, n, reps=5):
    best = 1e9
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(i)
        e1.record(); e1.synchronize()
        if r: best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


for fmt in fmts:
    for M in Ms:
        for name, N, K, swiglu, norm in shapes:
            x = torch.randn((32, K), device=dev).half()
            gamma = torch.ones(K, device=dev).half()
            outN = N // 2 if swiglu else N
            y = torch.empty((32, outN), device=dev, dtype=torch.float16)
            res = torch.zeros((32, outN), device=dev, dtype=torch.float16)
            sets = []
            for _ in range(NSETS):
                w = (torch.randn((N, K), device=dev) / K ** 0.5).half()
                if fmt == "f16":
                    store, sc = w, None
                elif fmt == "int8":
                    store = torch.empty((N, K), dtype=torch.int8, device=dev); sc = torch.empty(N, dtype=torch.float16, device=dev)
                    llmie.quantize_w8(w, store, sc)
                elif fmt == "int4":
                    store = torch.empty((N, K // 2), dtype=torch.uint8, device=dev); sc = torch.empty((N, K // 128), dtype=torch.float16, device=dev)
                    llmie.quantize_w4(w, store, sc, 128)
                else:
                    store = torch.empty((N, K), dtype=torch.uint8, device=dev); sc = torch.empty(N, dtype=torch.float32, device=dev)
                    llmie.quantize_fp8(w, store, sc)
                packed, ps = llmie.pack_weight(code[fmt], store, sc, swiglu)
                sets.append((packed, ps if fmt == "int4" else sc))
                del w, store
            ws_bytes = lib.llmie_linear_packed_workspace_bytes(code[fmt], M, K, N)
            ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
            st = llmie._st()

            def call(i):
                p, s = sets[i % NSETS]
                rc = lib.llmie_linear_packed(code[fmt], llmie._p(x), llmie._p(p), llmie._p(s), llmie._p(y), M, K, N, int(swiglu), X32,
                                             llmie._p(res) if not swiglu and not norm else None, llmie._p(gamma) if norm else None, None, 1e-5,
                                             llmie._p(ws), ws_bytes, st)
                assert rc == 0, lib.llmie_last_error()
            t = timeit(call, 2 * NSETS)
            print("%-5s M=%-3d %-8s %7.2f us  %5.2f TB/s (weights %6.1f MB)" % (fmt, M, name, t, N * K * bpe[fmt] / t / 1e6, N * K * bpe[fmt] / 1e6), flush=True)
            del sets
            torch.cuda.empty_cache()

#!/usr/bin/env python3
"""Per-kernel micro-benchmark through the C ABI (development tool): cycles over NSETS weight copies so every
launch streams from HBM; reports us per launch (incl. ~1 us launch gap) and effective TB/s."""
import importlib.util, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("llmie_amd", os.path.join(ROOT, "llm-inference-engine_amd", "__init__.py"))
llmie = importlib.util.module_from_spec(spec); sys.modules["llmie_amd"] = llmie; spec.loader.exec_module(llmie)
dev = "cuda"
NSETS = 12

def timeit(fn, n=NSETS, reps=5):
    best = 1e9
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(i)
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        if r: best = min(best, us)
    return best

shapes = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)]
if "prefill" in sys.argv:
    for M in (512, 2048, 4096):
        for name, N, K in shapes:
            x = torch.randn((M, K), device=dev).half()
            y = torch.empty((M, N), device=dev, dtype=torch.float16)
            W = [(torch.randn((N, K), device=dev) / K ** 0.5).half() for _ in range(4)]
            t = timeit(lambda i: llmie.linear(x, W[i % 4], y), n=8)
            print("M=%d %-8s f16 %8.1f us  %6.1f TFLOP/s" % (M, name, t, 2.0 * M * N * K / t / 1e6))
            if name == "gate_up":
                y2 = torch.empty((M, N // 2), device=dev, dtype=torch.float16)
                t2 = timeit(lambda i: llmie.silu_and_mul(y.view(M, 2, N // 2), y2), n=8)
                try:
                    t3 = timeit(lambda i: llmie.linear_swiglu(x, W[i % 4], y2), n=8)
                    print("M=%d %-8s f16 SiluAndMul alone %6.1f us; fused projection+SwiGLU %8.1f us  %6.1f TFLOP/s" % (M, name, t2, t3, 2.0 * M * N * K / t3 / 1e6))
                except llmie.LlmieError:
                    print("M=%d %-8s f16 SiluAndMul alone %6.1f us; no fused form at this size" % (M, name, t2))
            Q = []
            for w in W:
                q = torch.empty((N, K), dtype=torch.uint8, device=dev); sc = torch.empty(N, dtype=torch.float32, device=dev)
                llmie.quantize_fp8(w, q, sc); Q.append((q, sc))
            work = torch.empty(llmie.linear_fp8_workspace_bytes(M, K, N), dtype=torch.uint8, device=dev)
            t = timeit(lambda i: llmie.linear_fp8(x, Q[i % 4][0], Q[i % 4][1], y, work), n=8)
            print("M=%d %-8s fp8 %8.1f us  %6.1f TFLOP/s (incl. activation quantisation)" % (M, name, t, 2.0 * M * N * K / t / 1e6))
            del W, Q
    sys.exit(0)
Ms = [int(a) for a in sys.argv[1:]] or [1]
for M in Ms:
    for name, N, K in shapes:
        x = torch.randn((M, K), device=dev).half()
        y = torch.empty((M, N), device=dev, dtype=torch.float16)
        W = [(torch.randn((N, K), device=dev) / K ** 0.5).half() for _ in range(NSETS)]
        t = timeit(lambda i: llmie.linear(x, W[i], y))
        print("M=%d %-8s f16  %7.2f us  %5.2f TB/s" % (M, name, t, N * K * 2 / t / 1e6))
        Q = []
        for w in W:
            q = torch.empty((N, K), dtype=torch.int8, device=dev); s = torch.empty(N, dtype=torch.float16, device=dev)
            llmie.quantize_w8(w, q, s); Q.append((q, s))
        t = timeit(lambda i: llmie.linear_w8a16(x, Q[i][0], Q[i][1], y))
        print("M=%d %-8s int8 %7.2f us  %5.2f TB/s" % (M, name, t, N * K / t / 1e6))
        if M <= 2:
            Q4 = []
            for w in W:
                q = torch.empty((N, K // 2), dtype=torch.uint8, device=dev); s = torch.empty((N, K // 128), dtype=torch.float16, device=dev)
                llmie.quantize_w4(w, q, s, 128); Q4.append((q, s))
            t = timeit(lambda i: llmie.linear_w4a16(x, Q4[i][0], Q4[i][1], y, 128))
            print("M=%d %-8s int4 %7.2f us  %5.2f TB/s" % (M, name, t, N * K / 2 / t / 1e6))
            del Q4
        del W, Q
        torch.cuda.empty_cache()

# ---- prefill GEMM (MFMA-bound): TFLOP/s ----
if "prefill" in sys.argv:
    pass

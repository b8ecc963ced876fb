#!/usr/bin/env python3
"""Prefill GEMM timing (development tool): this library's linear / fused SwiGLU projection against torch.matmul (vendor GEMM) at
the Llama-2-7B projection shapes:  python tools/gemmbench.py [M ...]"""
import importlib.util, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("llmie_amd", os.path.join(ROOT, "llm-inference-engine_amd", "__init__.py"))
llmie = importlib.util.module_from_spec(spec); sys.modules["llmie_amd"] = llmie; spec.loader.exec_module(llmie)
if os.environ.get("LLMIE_LIB"):   # another build of the library (A/B of one kernel change in one gpurun call)
    llmie.LIB_PATH = os.environ["LLMIE_LIB"]


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


for M in [int(a) for a in sys.argv[1:]] or [4096, 2048]:
    for name, N, K in [("qkv", 12288, 4096), ("o", 4096, 4096), ("down", 4096, 11008)]:
        x = torch.randn((M, K), device="cuda").half()
        W = (torch.randn((N, K), device="cuda") / K ** 0.5).half()
        y = torch.empty((M, N), device="cuda", dtype=torch.float16)
        fl = 2.0 * M * N * K
        t = timed(lambda: llmie.linear(x, W, y))
        tv = timed(lambda: torch.matmul(x, W.t(), out=y))
        ref = (x[:64].float() @ W.float().t())
        llmie.linear(x, W, y)
        err = (y[:64].float() - ref).abs().max().item()
        print("M=%d %-8s llmie %7.1f us %7.1f TF | vendor %7.1f us %7.1f TF | max err %.3g" % (M, name, t * 1e6, fl / t / 1e12, tv * 1e6, fl / tv / 1e12, err))
    I, K = 11008, 4096
    x = torch.randn((M, K), device="cuda").half()
    W = (torch.randn((2 * I, K), device="cuda") / K ** 0.5).half()
    y = torch.empty((M, I), device="cuda", dtype=torch.float16)
    t = timed(lambda: llmie.linear_swiglu(x, W, y))
    print("M=%d %-8s llmie %7.1f us %7.1f TF (fused SwiGLU)" % (M, "gate_up", t * 1e6, 2.0 * M * 2 * I * K / t / 1e12))

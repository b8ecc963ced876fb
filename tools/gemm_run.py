#!/usr/bin/env python3
"""Runs the prefill projections a few times (development tool, for rocprofv3): python tools/gemm_run.py M"""
import importlib.util, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("llmie_amd", os.path.join(ROOT, "llm-inference-engine_amd", "__init__.py"))
llmie = importlib.util.module_from_spec(spec); sys.modules["llmie_amd"] = llmie; spec.loader.exec_module(llmie)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name, N, K in [("qkv", 12288, 4096), ("down", 4096, 11008)]:
    x = torch.randn((M, K), device="cuda").half()
    W = (torch.randn((N, K), device="cuda") / K ** 0.5).half()
    y = torch.empty((M, N), device="cuda", dtype=torch.float16)
    for _ in range(4):
        llmie.linear(x, W, y)
torch.cuda.synchronize()
print("done")

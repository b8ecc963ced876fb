#!/usr/bin/env python3
"""Diagnostic: what a batch-1 GEMV gains when its weights are already in the 256 MiB Infinity Cache.
States: cold (512 MiB written to another buffer before each launch), replay (the same launch back to back; the kernel's own loads are
non-temporal), warmed (a default-policy read pass over the weights -- torch sum -- before each launch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
llmie = bench.load_llmie()
dev = "cuda"
junk = torch.empty(1 << 29, dtype=torch.uint8, device=dev)


def timed(fn, pre, n=20):
    ts = []
    for i in range(n):
        pre(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for name, K, N in (("o", 4096, 4096), ("qkv", 4096, 12288), ("down", 11008, 4096), ("gate_up", 4096, 22016)):
    w = (torch.randn((N, K), device=dev) * 0.02).half()
    x = torch.randn((1, K), device=dev).half()
    y = torch.empty((1, N), device=dev, dtype=torch.float16)
    fn = lambda: llmie.linear(x, w, y, workspace=None)
    mb = N * K * 2 / 1e6
    cold = timed(fn, lambda i: junk.fill_(i & 255))
    replay = timed(fn, lambda i: None)
    def warm(i):
        junk.fill_(i & 255)
        w.view(torch.int16).sum()
    warmed = timed(fn, warm)
    print("%-8s %6.1f MB  cold %6.2f us (%.2f TB/s)  replay %6.2f us  warmed-by-read %6.2f us (%.2f TB/s)" %
          (name, mb, cold, mb / cold, replay, warmed, mb / warmed), flush=True)

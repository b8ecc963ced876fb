#!/usr/bin/env python3
"""Development tool: long race screen of the eight-phase GEMM kernels (gemm8p.cuh) -- every launch of a shape must reproduce the first
one bit for bit (fixed accumulation order); fp16 plain / residual / SwiGLU and e4m3, both tile widths, with a cache-thrashing fill and
a concurrent copy stream perturbing the DMA timing.   python tools/gemm8p_soak.py [launches per shape]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
llmie = bench.load_llmie()
N_LAUNCH = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda"
junk = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()
src, dst = torch.empty(1 << 27, dtype=torch.uint8, device=dev), torch.empty(1 << 27, dtype=torch.uint8, device=dev)
bad = 0


def screen(name, fn, y):
    global bad
    fn(); torch.cuda.synchronize()
    y0 = y.clone()
    for it in range(N_LAUNCH):
        if it % 7 == 0:
            junk.fill_(it & 255)
        if it % 3 == 0:
            with torch.cuda.stream(side):
                dst.copy_(src, non_blocking=True)   # runs beside the next launch
        y.fill_(3.0)
        fn()
        if not torch.equal(y, y0):
            bad += 1
            print("MISMATCH %s launch %d: %d elements" % (name, it, (y != y0).sum().item()), flush=True)
            break
    torch.cuda.synchronize()
    print("%-34s %d launches identical" % (name, N_LAUNCH) if not bad else "%-34s FAILED" % name, flush=True)


g = torch.Generator(device=dev).manual_seed(1)
for M, K, N in ((2048, 4096, 8192), (2048, 11008, 4096), (4096, 4096, 4096), (2048, 4096, 4096), (4000, 2112, 3100)):
    x = torch.randn((M, K), device=dev, generator=g).half()
    w = (torch.randn((N, K), device=dev, generator=g) / K ** 0.5).half()
    y = torch.empty((M, N), device=dev, dtype=torch.float16)
    screen("fp16 plain %dx%dx%d" % (M, N, K), lambda: llmie.linear(x, w, y), y)
    r = torch.randn((M, N), device=dev, generator=g).half()
    b = torch.randn((N,), device=dev, generator=g).half()
    screen("fp16 bias+res %dx%dx%d" % (M, N, K), lambda: llmie.linear(x, w, y, bias=b, residual=r), y)
for M, K, I in ((2048, 4096, 11008), (4096, 4096, 11008), (1024, 4096, 11008)):
    x = torch.randn((M, K), device=dev, generator=g).half()
    w = (torch.randn((2 * I, K), device=dev, generator=g) / K ** 0.5).half()
    y = torch.empty((M, I), device=dev, dtype=torch.float16)
    screen("fp16 SwiGLU %dx%dx%d" % (M, I, K), lambda: llmie.linear_swiglu(x, w, y), y)
    wq = torch.empty((2 * I, K), dtype=torch.uint8, device=dev); ws = torch.empty(2 * I, dtype=torch.float32, device=dev)
    llmie.quantize_fp8(w, wq, ws)
    work = torch.empty(llmie.linear_fp8_workspace_bytes(M, K), dtype=torch.uint8, device=dev)
    screen("e4m3 SwiGLU %dx%dx%d" % (M, I, K), lambda: llmie.linear_fp8_swiglu(x, wq, ws, y, work), y)
for M, K, N in ((2048, 4096, 12288), (2048, 11008, 4096), (4096, 4096, 4096)):
    x = torch.randn((M, K), device=dev, generator=g).half()
    w = (torch.randn((N, K), device=dev, generator=g) / K ** 0.5).half()
    wq = torch.empty((N, K), dtype=torch.uint8, device=dev); ws = torch.empty(N, dtype=torch.float32, device=dev)
    llmie.quantize_fp8(w, wq, ws)
    work = torch.empty(llmie.linear_fp8_workspace_bytes(M, K, N), dtype=torch.uint8, device=dev)
    y = torch.empty((M, N), device=dev, dtype=torch.float16)
    screen("e4m3 plain %dx%dx%d" % (M, N, K), lambda: llmie.linear_fp8(x, wq, ws, y, work), y)
print("RESULT:", "FAILED" if bad else "all identical")
sys.exit(1 if bad else 0)

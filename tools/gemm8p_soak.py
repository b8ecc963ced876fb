#!/usr/bin/env python3
"""Development tool: long race screen of the eight-phase GEMM kernels (gemm8p.cuh) -- every launch of a shape must reproduce the first
one bit for bit (fixed accumulation order); fp16 plain / residual / SwiGLU and e4m3, both tile widths, with a cache-thrashing fill and
a concurrent copy stream perturbing the DMA timing; round 3: the ROPE forms through whole prefill passes of one 7B-geometry
layer (hidden states and both caches).   python tools/gemm8p_soak.py [launches per shape] [--no-rope]"""
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
# ---- round 3: the ROPE forms (QKV projection with RoPE + KV-cache append as its epilogue) through one 7B-geometry engine layer:
#      hidden states AND the K cache of every repeated prefill must reproduce the first one bit for bit
if "--no-rope" not in sys.argv:
    import numpy as np
    nh, hs, I = 32, 128, 11008
    H, QKV = nh * hs, 3 * nh * hs
    mk = lambda n, k: (torch.randn((n, k), device=dev, generator=g) / k ** 0.5).half()
    raw = dict(qkv=mk(QKV, H), o=mk(H, H), gate_up=mk(2 * I, H), down=mk(H, I))
    gam = lambda: (torch.rand((H,), device=dev, generator=g) * 0.4 + 0.8).half()
    for wfmt in ("f16", "int8", "fp8"):
        def quant(w):
            n, k = w.shape
            if wfmt == "f16":
                return dict(data=w)
            if wfmt == "int8":
                q, sc = torch.empty((n, k), dtype=torch.int8, device=dev), torch.empty(n, dtype=torch.float16, device=dev)
                llmie.quantize_w8(w, q, sc)
            else:
                q, sc = torch.empty((n, k), dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
                llmie.quantize_fp8(w, q, sc)
            return dict(data=q, scale=sc)
        layers = [dict(attn_norm=gam(), ffn_norm=gam(), qkv=quant(raw["qkv"]), o=quant(raw["o"]), gate_up=quant(raw["gate_up"]), down=quant(raw["down"]))]
        for lens in ([2048], [512] * 8, [700, 300, 1048]):
            bs, T, max_seq = len(lens), sum(lens), max(lens)
            cfg = dict(head_num=nh, kv_head_num=nh, head_size=hs, inter_size=I, num_layers=1, vocab_size=100, max_seq_len=max_seq, max_batch=bs,
                       rotary_dim=hs, rotary_base=10000.0, rms_eps=1e-5, dtype=llmie.F16,
                       wfmt={"f16": llmie.W_F16, "int8": llmie.W_INT8, "fp8": llmie.W_FP8}[wfmt], int4_group=128)
            dec = llmie.Decoder(cfg, layers)
            x = torch.randn((T, H), device=dev, generator=g).half()
            out = torch.empty_like(x)
            kc = torch.zeros((1, bs, nh, max_seq, hs), device=dev, dtype=torch.float16)
            vc = torch.zeros_like(kc)
            ld = torch.tensor(lens, dtype=torch.int32, device=dev)
            hd = torch.zeros(bs, dtype=torch.int32, device=dev)
            both = torch.empty(out.numel() + kc.numel() + vc.numel(), device=dev, dtype=torch.float16)

            def fn():
                kc.zero_(); vc.zero_()
                dec.prefill(x, out, kc, vc, ld, hd, max_seq)
                both[:out.numel()] = out.flatten()
                both[out.numel():out.numel() + kc.numel()] = kc.flatten()
                both[out.numel() + kc.numel():] = vc.flatten()
            screen("prefill %s %s tokens (QKV + RoPE epilogue)" % (wfmt, "x".join(map(str, lens)) if len(set(lens)) > 1 else "%dx%d" % (bs, lens[0])), fn, both)
            dec.close()
print("RESULT:", "FAILED" if bad else "all identical")
sys.exit(1 if bad else 0)

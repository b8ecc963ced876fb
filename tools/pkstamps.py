#!/usr/bin/env python3
"""Diagnostic (needs a PK_STAMPS build of pk_linear.hip in /tmp/libllmie_stamps.so): s_memrealtime stamps of one launch."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
llmie = bench.load_llmie()
llmie.LIB_PATH = os.environ["LLMIE_STAMPS_LIB"]
lib = llmie.lib()
X32 = int(os.environ.get("PK_X32", "0"))
lib.llmie_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
M, K = 32, 4096
fmt = llmie.W_INT8
x = torch.randn((32, K), device="cuda").half()
gamma = torch.ones(K, device="cuda").half()
for mode, tiles in (("plain", 1), ("plain", 3), ("norm", 3)):
    N = 4096 * tiles
    w = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda")
    sc = torch.full((N,), 0.01, dtype=torch.float16, device="cuda")
    p, _ = llmie.pack_weight(fmt, w, sc, False)
    y = torch.zeros((32, N), device="cuda", dtype=torch.float16)
    junk = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    for rep in range(3):
        junk.fill_(rep)   # cold caches
        torch.cuda.synchronize()
        rc = lib.llmie_linear_packed(fmt, llmie._p(x), llmie._p(p), llmie._p(sc), llmie._p(y), M, K, N, 0, X32, None,
                                     llmie._p(gamma) if mode == "norm" else None, None, 1e-5, None, 0, llmie._st())
        assert rc == 0
        torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * 16, dtype=np.uint64)
    assert lib.llmie_debug_stamps(buf.ctypes.data, buf.nbytes) == 0
    st = buf.reshape(256, 8, 16)[:, :, :10].astype(np.int64)
    t0 = st[:, :, 0].min()
    rel = (st - t0) * 0.01   # us (100 MHz)
    names = ["start", "loads issued", "first wait", "staged+barrier", "prologue end", "tile0 blocks", "tile0 done", "end", "pre issued", "x issued"]
    print("== %s, %d tile(s)/WG: median over waves [min .. max] us since the first wave started" % (mode, tiles))
    for i, n in enumerate(names):
        v = rel[:, :, i].reshape(-1)
        print("  %-14s %6.2f  [%6.2f .. %6.2f]" % (n, np.median(v), v.min(), v.max()))

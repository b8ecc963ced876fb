"""GPU parity tests: every C-ABI kernel entry point (include/llmie.h section 1) against the
CPU oracle on the same seeded inputs.  Shapes follow the reference's own unit tests
(tests/unit_tests/*.cu, cited per test), config A (hidden 128, 4 heads, seq 32) and the
Llama-2-7B shapes, plus ragged / empty / maximum-size edge cases.

Bar: bit-exact for integer / byte / index work (embedding, padding offset, causal mask,
KV append / broadcast, transpose, top-k ids and values); stated tolerance for floating point.
fp16 tolerances: the oracle computes in fp32 from the same fp16-rounded inputs, the kernel
accumulates in fp32 and rounds once to fp16 => |err| <= 2^-10 * |value| + accumulated fp32
noise; tolerances below are written as (rtol, atol) per test.
"""
import numpy as np
import pytest
import torch

import oracle as orc

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.float16]
DEV = "cuda"


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t.to(dtype) if dtype is not None else t


def host(t):
    return t.detach().float().cpu().numpy() if t.is_floating_point() else t.detach().cpu().numpy()


def rnd(rng, shape, scale=1.0, dtype=torch.float32):
    """random values already rounded to `dtype` (returned as fp32 numpy)"""
    a = (rng.standard_normal(shape) * scale).astype(np.float32)
    if dtype == torch.float16:
        a = a.astype(np.float16).astype(np.float32)
    return a


def tol(dtype, f32=(1e-5, 1e-5), f16=(2e-3, 2e-3)):
    return f32 if dtype == torch.float32 else f16


def close(got, exp, rtol, atol):
    got, exp = np.asarray(got, np.float32), np.asarray(exp, np.float32)
    err = np.abs(got - exp)
    bound = atol + rtol * np.abs(exp)
    assert (err <= bound).all(), "max err %g at %s (exp %g got %g)" % (
        err.max(), np.unravel_index(np.argmax(err - bound), err.shape),
        exp.reshape(-1)[np.argmax(err - bound)], got.reshape(-1)[np.argmax(err - bound)])


# --------------------------------------------------------------------------- embedding
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,H,V", [(64, 4096, 32000), (1, 128, 50), (7, 100, 33)])
def test_input_embedding(llmie, dtype, T, H, V):
    # tests/unit_tests/test_input_embedding.cu: T=64, H=4096, V=32000
    rng = np.random.default_rng(42)
    ids = rng.integers(0, V, T).astype(np.int32)
    table = rnd(rng, (V, H), dtype=dtype)
    out = torch.full((T, H), -5.0, dtype=dtype, device=DEV)
    llmie.input_embedding(dev(ids), dev(table, dtype), out)
    assert np.array_equal(host(out), orc.input_embedding(ids, table))  # bit-exact gather


def test_input_embedding_out_of_range_ids_are_skipped(llmie):
    table = torch.arange(40, dtype=torch.float32, device=DEV).reshape(10, 4)
    out = torch.full((3, 4), -1.0, device=DEV)
    llmie.input_embedding(dev(np.array([3, 10, -1], np.int32)), table, out)
    assert host(out).tolist() == [[12, 13, 14, 15], [-1] * 4, [-1] * 4]


# --------------------------------------------------------------------------- padding offset
@pytest.mark.parametrize("lens,max_q", [([4, 3, 5], 5), ([4, 3, 3, 4], 5), ([3, 3, 3], 5), ([1], 1),
                                        ([0, 2, 0, 7], 7), (list(range(1, 301)), 300)])
def test_cal_padding_offset(llmie, lens, max_q):
    # cal_padding_offset.cuh:8-15 doc example; test_cal_padding_offset.cu lens 4-(i*i%3)
    lens = np.array(lens, np.int32)
    off = torch.full((len(lens), max_q), -7, dtype=torch.int32, device=DEV)
    cum = torch.empty(len(lens) + 1, dtype=torch.int32, device=DEV)
    llmie.cal_padding_offset(off, cum, dev(lens))
    eoff, ecum = orc.cal_padding_offset(lens, max_q, fill=-7)
    assert np.array_equal(host(cum), ecum)
    assert np.array_equal(host(off), eoff)  # including the untouched tail


# --------------------------------------------------------------------------- causal mask
@pytest.mark.parametrize("dtype", DTYPES)
def test_build_causal_mask_reference_case(llmie, golden, dtype):
    g = golden["causal_mask_rand"]  # test_build_causal_mask.cu: bs=64, q=128, k=512, rand() lens
    mask = torch.empty((g["batch"], g["max_q_len"], g["max_k_len"]), dtype=dtype, device=DEV)
    llmie.build_causal_mask(mask, dev(np.array(g["q_lens"], np.int32)), dev(np.array(g["k_lens"], np.int32)))
    exp = orc.build_causal_mask(g["q_lens"], g["k_lens"], g["max_q_len"], g["max_k_len"])
    assert np.array_equal(host(mask), exp)
    assert int(host(mask).sum()) == g["ones"]


def test_build_causal_mask_edges(llmie):
    ql, kl = np.array([0, 3, 5, 2], np.int32), np.array([0, 3, 9, 1], np.int32)  # empty, square, history, q>k
    mask = torch.empty((4, 5, 9), device=DEV)
    llmie.build_causal_mask(mask, dev(ql), dev(kl))
    assert np.array_equal(host(mask), orc.build_causal_mask(ql, kl, 5, 9))


# --------------------------------------------------------------------------- norms
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,H", [(64, 4096), (1, 128), (32, 128), (3, 11008), (2, 100), (1, 40000)])
def test_rmsnorm(llmie, dtype, T, H):
    # test_rmsnorm.cu: T=64, H=4096, eps=1e-6, tolerance 1e-3
    rng = np.random.default_rng(1)
    x, gam = rnd(rng, (T, H), dtype=dtype), rnd(rng, (H,), 0.5, dtype) + 1.0
    if dtype == torch.float16:
        gam = gam.astype(np.float16).astype(np.float32)
    xd, rd = dev(x, dtype), torch.empty((T, H), dtype=dtype, device=DEV)
    llmie.rmsnorm(xd, rd, dev(gam, dtype), 1e-6)
    ey, er = orc.rmsnorm(x, gam, 1e-6)
    assert np.array_equal(host(rd), er)  # residual copy is bit-exact
    close(host(xd), ey, *tol(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_rmsnorm_reference_patterns(llmie, golden, dtype):
    g = golden["rmsnorm_fp32_pattern"]
    T, H = g["tokens"], g["hidden"]
    idx = np.arange(T * H, dtype=np.int64)
    x = ((idx * idx) % 3 + 1).astype(np.float32).reshape(T, H)
    gam = (np.arange(H) % 3 + 1).astype(np.float32)
    xd = dev(x, dtype)
    llmie.rmsnorm(xd, None, dev(gam, dtype), g["eps"])  # residual is optional
    flat = host(xd).reshape(-1)
    for i, e in zip(g["sample_index"], g["sample_expected"]):
        assert abs(flat[i] - e) <= (g["tol"] if dtype == torch.float32 else 2e-3 * abs(e))
    ones = torch.ones((64, 4096), dtype=dtype, device=DEV)
    llmie.rmsnorm(ones, None, torch.ones(4096, dtype=dtype, device=DEV), 1e-6)
    assert np.abs(host(ones) - 1.0).max() <= 1e-3  # golden rmsnorm_ones


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,H,bias,resid", [(2048, 128, True, True), (4, 4096, False, True), (5, 4096, True, True),
                                            (3, 344, True, False), (2, 100, False, True)])
def test_fused_add_bias_residual_rmsnorm(llmie, dtype, T, H, bias, resid):
    # test_add_residual_and_rmsnorm.cu: T=2048, H=128, eps=0.5 ; spec = fp32 kernel :43-121
    rng = np.random.default_rng(2)
    o, r = rnd(rng, (T, H), dtype=dtype), rnd(rng, (T, H), dtype=dtype)
    b = rnd(rng, (H,), 0.1, dtype) if bias else None
    gam = rnd(rng, (H,), 0.2, dtype) + 1.0
    if dtype == torch.float16:
        gam = gam.astype(np.float16).astype(np.float32)
    od, rd = dev(o, dtype), (dev(r, dtype) if resid else None)
    llmie.fused_add_bias_residual_rmsnorm(rd, od, None if b is None else dev(b, dtype), dev(gam, dtype), 0.5)
    eo, er = orc.fused_add_bias_residual_rmsnorm(r if resid else None, o, b, gam, 0.5) if resid else \
        orc.fused_add_bias_residual_rmsnorm(np.zeros_like(o), o, b, gam, 0.5)
    if resid:
        # new residual = out + residual, rounded once to the storage dtype
        er_round = er.astype(np.float16).astype(np.float32) if dtype == torch.float16 else er
        close(host(rd), er_round, 0, 0 if dtype == torch.float32 else 1e-3)
    close(host(od), eo, *tol(dtype, f16=(4e-3, 4e-3)))


@pytest.mark.parametrize("dtype", DTYPES)
def test_fused_norm_reference_recipe(llmie, golden, dtype):
    g = golden["fused_norm_ones"]  # test_add_residual_and_rmsnorm.cu:60-80: out 1, residual 0, bias 0, gamma 1, eps 0.5
    T, H = g["tokens"], g["hidden"]
    od = torch.full((T, H), g["out_fill"], dtype=dtype, device=DEV)
    rd = torch.full((T, H), g["residual_fill"], dtype=dtype, device=DEV)
    llmie.fused_add_bias_residual_rmsnorm(rd, od, torch.full((H,), g["bias_fill"], dtype=dtype, device=DEV),
                                          torch.full((H,), g["gamma_fill"], dtype=dtype, device=DEV), g["eps"])
    assert np.abs(host(od) - g["expected"]).max() <= (1e-6 if dtype == torch.float32 else 5e-4)  # the reference's bar: 1e-3
    assert (host(rd) == g["expected_residual"]).all()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,H", [(16, 4096), (1, 128), (3, 33)])
def test_add_residual(llmie, dtype, T, H):
    # test_add_residual.cu: T=16, H=4096, both (i%2)+1  -> exact
    n = T * H
    a = (np.arange(n) % 2 + 1).astype(np.float32).reshape(T, H)
    od = dev(a, dtype)
    llmie.add_residual(dev(a, dtype), od)
    assert np.array_equal(host(od), orc.add_residual(a, a))
    rng = np.random.default_rng(3)
    x, y = rnd(rng, (T, H), dtype=dtype), rnd(rng, (T, H), dtype=dtype)
    od = dev(y, dtype)
    llmie.add_residual(dev(x, dtype), od)
    close(host(od), orc.add_residual(x, y), *tol(dtype, f32=(0, 0), f16=(1e-3, 1e-6)))


# --------------------------------------------------------------------------- linear
def test_linear_reference_case_exact(llmie, golden):
    # test_linear.cu: srand(233), rand()%3, M=64, K=N=4096, y = x.W^T -- small ints, exact in fp32
    import ctypes
    import hashlib
    g = golden["linear_srand233"]
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(g["srand"])
    w = np.array([libc.rand() % 3 for _ in range(g["N"] * g["K"])], np.float32).reshape(g["N"], g["K"])
    x = np.array([libc.rand() % 3 for _ in range(g["M"] * g["K"])], np.float32).reshape(g["M"], g["K"])
    y = torch.empty((g["M"], g["N"]), device=DEV)
    llmie.linear(dev(x), dev(w), y, trans_b=True)
    yi = np.rint(host(y)).astype(np.int32)
    assert np.abs(host(y) - yi).max() == 0.0
    assert hashlib.sha256(yi.tobytes()).hexdigest() == g["y_sha256_int32"]
    # fp16 path (skinny MFMA, M=64): integers up to 2*2*4096 -> exact in the fp32 accumulator,
    # rounded to fp16 on store (values <= 16384 need 15 bits: compare against the fp16-rounded truth)
    y16 = torch.empty((g["M"], g["N"]), dtype=torch.float16, device=DEV)
    llmie.linear(dev(x, torch.float16), dev(w, torch.float16), y16, trans_b=True)
    assert np.array_equal(host(y16), yi.astype(np.float32).astype(np.float16).astype(np.float32))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K,N,trans_b", [
    (1, 4096, 4096, True), (1, 4096, 12288, True), (1, 11008, 4096, True), (1, 4096, 22016, True),
    (2, 4096, 4096, True), (3, 4096, 1024, True), (4, 11008, 512, True), (5, 4096, 512, True),
    (8, 4096, 1024, True), (8, 11008, 256, True),
    (13, 4096, 1024, True), (16, 4096, 1024, True), (17, 4096, 1000, True), (32, 4096, 2048, True),
    (33, 11008, 512, True), (64, 4096, 512, True),
    (65, 4096, 256, True), (128, 128, 384, True), (200, 4096, 384, True), (130, 11008, 128, True), (256, 128, 130, True),
    (2048, 4096, 512, True), (32, 128, 688, True), (32, 344, 128, True),
    (1, 128, 384, True), (7, 100, 37, True), (1, 4096, 4095, True), (3, 4104, 77, True),
    (4, 128, 96, False), (33, 100, 37, False), (128, 512, 256, False)])
def test_linear(llmie, dtype, M, K, N, trans_b):
    rng = np.random.default_rng(M * 7 + N)
    x = rnd(rng, (M, K), 1.0, dtype)
    w = rnd(rng, (N, K) if trans_b else (K, N), 1.0 / np.sqrt(K), dtype)
    y = torch.full((M, N), 99.0, dtype=dtype, device=DEV)
    llmie.linear(dev(x, dtype), dev(w, dtype), y, trans_b=trans_b)
    exp = orc.linear(x, w, trans_b=trans_b)
    # K-term fp32 accumulation in a different order than the oracle: |err| <~ sqrt(K)*2^-24*|x||w| ~ 1e-5
    close(host(y), exp, *tol(dtype, f32=(1e-4, 2e-5), f16=(2e-3, 2e-3)))


@pytest.mark.parametrize("M,K,N", [(128, 4096, 12288), (100, 11008, 4096), (96, 4096, 8200), (77, 1024, 8194), (128, 512, 130),
                                   (128, 4096, 22016), (90, 1024, 22010),   # 192-row tiles (115 x 2 workgroups), ragged last tile
                                   (65, 4096, 22016), (200, 2048, 8448),
                                   (64, 4096, 12288), (33, 11008, 4096), (50, 1024, 8194), (40, 4096, 22016), (64, 512, 130)])
def test_linear_128_row_splitk(llmie, M, K, N):
    """32 < M <= 128 rows per pass take the LDS-DMA split-K kernel (gemm_mid.cuh; 64- and 128-row activation tiles): 256- and 128-row weight tiles, ragged K
    slices (64 k-tiles over 5 slices), ragged N tiles, N % 4 != 0 (scalar slab stores), clamped activation rows, two passes"""
    rng = np.random.default_rng(M + N)
    x, w = rnd(rng, (M, K), 1.0, torch.float16), rnd(rng, (N, K), 1.0 / np.sqrt(K), torch.float16)
    y = torch.full((M, N), 99.0, dtype=torch.float16, device=DEV)
    llmie.linear(dev(x, torch.float16), dev(w, torch.float16), y)
    close(host(y), orc.linear(x, w), 2e-3, 2e-3)


@pytest.mark.parametrize("M,K,N,epi", [(4096, 128, 3072, False), (4000, 192, 3100, False), (3900, 256, 3330, True),
                                       (8192, 64, 2048, True), (4096, 192, 1664, False), (4090, 128, 1602, True),
                                       (2048, 128, 12288, True), (2000, 192, 12200, False),
                                       # k-tile counts that end the eight-phase loops in each of their tails: 5 and 8 k-tiles on
                                       # the 256-wide kernel (two per loop body), 7 and 8 on the 128-wide one (three per body)
                                       (4096, 320, 3072, True), (4000, 512, 3100, False), (4096, 448, 1600, False),
                                       (4090, 512, 1664, True)])
def test_linear_gemm256(llmie, M, K, N, epi):
    """shapes whose 256 x 256 grid fills the chip (>= 192 tiles) take the LDS-DMA kernel (gemm256.cuh): full tiles,
    ragged M and N edges, odd k-tile counts, bias + in-place residual epilogue; the last two shapes take the two-launch plan
    (whole rounds of 256-wide tiles + the remaining columns 128-wide)"""
    rng = np.random.default_rng(M + N)
    x, w = rnd(rng, (M, K), 1.0, torch.float16), rnd(rng, (N, K), 1.0 / np.sqrt(K), torch.float16)
    if epi:
        b, r = rnd(rng, (N,), 1.0, torch.float16), rnd(rng, (M, N), 1.0, torch.float16)
        y = dev(r, torch.float16)
        llmie.linear(dev(x, torch.float16), dev(w, torch.float16), y, bias=dev(b, torch.float16), residual=y)
        close(host(y), orc.linear(x, w) + b[None, :] + r, 2e-3, 4e-3)
    else:
        y = torch.full((M, N), 99.0, dtype=torch.float16, device=DEV)
        llmie.linear(dev(x, torch.float16), dev(w, torch.float16), y)
        close(host(y), orc.linear(x, w), 2e-3, 2e-3)


@pytest.mark.parametrize("M,K,N", [(2048, 4096, 8192), (2048, 4096, 4096), (4096, 2112, 3072), (2000, 11008, 4000), (512, 704, 16384)])
def test_linear_eight_phase_schedule_race_screen(llmie, M, K, N):
    """gemm8p.cuh orders its LDS-DMA against its fragment reads by counted vmcnt waits and raw barriers only -- a misplaced wait
    would show as a rare wrong tile, not as a steady error.  The accumulation order is fixed, so every launch must reproduce the
    first one bit for bit: 24 launches per shape (256 x 256 and 256 x 128 tiles, even / odd / multiple-of-three k-tile counts,
    ragged edges), a 512 MiB fill between some of them (cold L2 / Infinity Cache changes the DMA landing order), and the first
    result checked against the fp32 reference GEMM of the same operands."""
    g = torch.Generator(device=DEV).manual_seed(M + K + N)
    x = torch.randn((M, K), device=DEV, generator=g).half()
    w = (torch.randn((N, K), device=DEV, generator=g) / K ** 0.5).half()
    y0 = torch.empty((M, N), device=DEV, dtype=torch.float16)
    llmie.linear(x, w, y0)
    ref = x[:192].float() @ w.float().t()
    assert (y0[:192].float() - ref).abs().max().item() < 2e-2
    ref = x[-64:].float() @ w.float().t()
    assert (y0[-64:].float() - ref).abs().max().item() < 2e-2
    junk = torch.empty(1 << 29, dtype=torch.uint8, device=DEV)
    y = torch.empty_like(y0)
    for it in range(24):
        if it % 3 == 0:
            junk.fill_(it)
        y.fill_(7.0)
        llmie.linear(x, w, y)
        assert torch.equal(y, y0), "launch %d differs from the first in %d elements" % (it, (y != y0).sum().item())


@pytest.mark.parametrize("M,K,I", [(4096, 128, 3072), (4000, 192, 3100), (300, 256, 344),
                                   (4000, 192, 3000), (1024, 320, 11008)])   # two-launch plans: 128-column tiles for whole rounds + 64-column tiles (ragged last tile)
def test_linear_swiglu_large_m(llmie, M, K, I):
    """ffn.cpp:105-122 in one launch at prefill sizes: the 256-token tile multiplies 128 gate and the matching 128 up rows
    and forms silu(gate) * up in registers (gemm256.cuh SWIGLU form); the small case falls back to ... an error unless a
    fused form exists, so it is skipped when unsupported"""
    rng = np.random.default_rng(M + I)
    x, w = rnd(rng, (M, K), 1.0, torch.float16), rnd(rng, (2 * I, K), 1.0 / np.sqrt(K), torch.float16)
    y = torch.full((M, I), 99.0, dtype=torch.float16, device=DEV)
    try:
        llmie.linear_swiglu(dev(x, torch.float16), dev(w, torch.float16), y)
    except llmie.LlmieError:
        assert M < 1000  # only the shapes without a fused form may refuse
        return
    gu = np.float16(orc.linear(x, w)).astype(np.float32)  # the projection output is rounded to fp16 before SiluAndMul
    close(host(y), orc.silu_and_mul(gu.reshape(M, 2, I)), 3e-3, 3e-3)


@pytest.mark.parametrize("M", [1, 4, 20, 150])
def test_linear_fused_bias_residual(llmie, M):
    rng = np.random.default_rng(9)
    K, N = 4096, 512
    x, w = rnd(rng, (M, K), 1.0, torch.float16), rnd(rng, (N, K), 1 / 64., torch.float16)
    b, r = rnd(rng, (N,), 1.0, torch.float16), rnd(rng, (M, N), 1.0, torch.float16)
    y = dev(r, torch.float16)  # residual aliases the output (in-place add)
    llmie.linear(dev(x, torch.float16), dev(w, torch.float16), y, bias=dev(b, torch.float16), residual=y)
    close(host(y), orc.linear(x, w) + b[None, :] + r, 2e-3, 4e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("bs,nh,m,n,k,trans_b", [(1, 4, 32, 32, 32, True), (1, 4, 32, 32, 32, False),
                                                 (2, 3, 17, 40, 128, True), (2, 3, 17, 128, 40, False),
                                                 (1, 32, 128, 128, 128, True)])
def test_batched_gemm(llmie, dtype, bs, nh, m, n, k, trans_b):
    # context_attention.cpp:240-271: q.k^T (trans_b) then p.v ; parity unpinned in the reference
    rng = np.random.default_rng(5)
    a = rnd(rng, (bs, nh, m, k), 1.0, dtype)
    b = rnd(rng, (bs, nh, n, k) if trans_b else (bs, nh, k, n), 1.0 / np.sqrt(k), dtype)
    c = torch.empty((bs, nh, m, n), dtype=dtype, device=DEV)
    llmie.batched_gemm(dev(a, dtype), dev(b, dtype), c, trans_b)
    close(host(c), orc.batched_gemm(a, b, trans_b), *tol(dtype, f32=(1e-4, 2e-5), f16=(2e-3, 2e-3)))


# --------------------------------------------------------------------------- RoPE
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("bs,nh,kvh,hs,step,rot", [(1, 32, 32, 128, 129, 128), (3, 32, 32, 128, 2048, 128),
                                                   (2, 4, 4, 32, 33, 32), (2, 8, 2, 64, 7, 64), (1, 4, 4, 32, 5, 16)])
def test_rope_decode(llmie, dtype, bs, nh, kvh, hs, step, rot):
    rng = np.random.default_rng(6)
    qkv = rnd(rng, (bs, nh + 2 * kvh, hs), 1.0, dtype)
    d = dev(qkv, dtype)
    llmie.rope_decode(d, nh, kvh, step, rot, 10000.0)
    exp = orc.rope_decode(qkv, nh, kvh, hs, step, rot, 10000.0)
    # device powf/sinf/cosf vs glibc at angles up to `step` rad: a 1-ulp angle error is ~step*6e-8
    close(host(d), exp, *tol(dtype, f32=(0, 3e-4 * max(1, step / 256)), f16=(2e-3, 2e-3)))
    assert np.array_equal(host(d)[:, nh + kvh:], qkv[:, nh + kvh:])  # v untouched, bit-exact
    # device-resident step (graph replay form)
    d2 = dev(qkv, dtype)
    llmie.rope_decode(d2, nh, kvh, -1, rot, 10000.0, step_dev=dev(np.array([step], np.int32)))
    assert torch.equal(d, d2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("bs,S,nh,kvh,hs,lens,hist,use_bias", [
    (1, 32, 32, 32, 128, [32], [0], False),            # test_qkv_bias_and_rope.cu shape
    (3, 8, 4, 4, 32, [8, 3, 5], [0, 4, 100], False),   # ragged + history
    (2, 6, 8, 2, 64, [6, 1], [7, 0], True)])           # GQA + bias (extension: bias added before rotation)
def test_qkv_bias_transpose_rope(llmie, dtype, bs, S, nh, kvh, hs, lens, hist, use_bias):
    rng = np.random.default_rng(7)
    T = int(sum(lens))
    off, _ = orc.cal_padding_offset(lens, S, fill=0)
    qkv = rnd(rng, (T, nh + 2 * kvh, hs), 1.0, dtype)
    bias = rnd(rng, ((nh + 2 * kvh) * hs,), 0.5, dtype) if use_bias else None
    q = torch.full((bs, nh, S, hs), 7.0, dtype=dtype, device=DEV)
    k = torch.full((bs, kvh, S, hs), 7.0, dtype=dtype, device=DEV)
    v = torch.full((bs, kvh, S, hs), 7.0, dtype=dtype, device=DEV)
    llmie.qkv_bias_transpose_rope(q, k, v, dev(qkv, dtype), None if bias is None else dev(bias, dtype),
                                  dev(off.reshape(-1)[:T].copy()), dev(np.array(hist, np.int32)), hs, 10000.0)
    eq, ek, ev = orc.qkv_bias_transpose_rope(qkv, bias, off.reshape(-1)[:T], hist, bs, S, nh, kvh, hs, hs,
                                             10000.0, fill=7.0)
    t = tol(dtype, f32=(0, 1e-4), f16=(2e-3, 2e-3))
    close(host(q), eq, *t)
    close(host(k), ek, *t)
    if not use_bias:
        assert np.array_equal(host(v), ev)  # pure copy + padding left untouched
    else:
        close(host(v), ev, *t)


# --------------------------------------------------------------------------- KV cache
@pytest.mark.parametrize("dtype", DTYPES)
def test_concat_and_repeat_kv(llmie, dtype):
    # test_concat_past_kv.cu (bs=1,q=16,max_seq=32,hs=8,kvh=2,history=1) + test_repeat_kv.cu, plus GQA/ragged
    rng = np.random.default_rng(8)
    for (L, bs, nh, kvh, max_q, max_seq, hs, cur, hist, layer) in [
            (1, 1, 2, 2, 16, 32, 8, [16], [1], 0),
            (3, 2, 8, 2, 5, 16, 32, [5, 2], [3, 0], 2),
            (2, 2, 4, 4, 4, 8, 128, [0, 4], [0, 4], 1)]:
        src = rnd(rng, (bs, kvh, max_q, hs), 1.0, dtype)
        cache0 = rnd(rng, (L, bs, kvh, max_seq, hs), 1.0, dtype)
        cd = dev(cache0, dtype)
        llmie.concat_kv(dev(src, dtype), cd, dev(np.array(cur, np.int32)), dev(np.array(hist, np.int32)), layer)
        exp = orc.concat_kv(src, cache0.copy(), cur, hist, layer)
        assert np.array_equal(host(cd), exp)  # bit-exact, untouched slots included
        ctx = np.array([c + h for c, h in zip(cur, hist)], np.int32)
        max_k = int(max(1, ctx.max()))
        dst = torch.full((bs, nh, max_k, hs), -3.0, dtype=dtype, device=DEV)
        llmie.repeat_kv(cd, dst, dev(ctx), layer)
        assert np.array_equal(host(dst), orc.repeat_kv(exp, ctx, layer, nh, max_k, fill=-3.0))


def test_concat_and_repeat_kv_reference_recipes(llmie, golden):
    g = golden["concat_kv_ones"]  # test_concat_past_kv.cu: ones, history 1, 16 new rows into a 32-row cache
    src = torch.full((g["batch"], g["kv_head_num"], g["max_q_len"], g["head_size"]), g["src_fill"], device=DEV)
    cache = torch.full((1, g["batch"], g["kv_head_num"], g["max_seq_len"], g["head_size"]), -7.0, device=DEV)
    llmie.concat_kv(src, cache, dev(np.array(g["cur_query_length"], np.int32)), dev(np.array(g["history_length"], np.int32)),
                    g["layer"])
    c = host(cache)
    lo, hi = g["written_rows"]
    assert (c[0, :, :, lo:hi + 1] == 1.0).all() and (c[0, :, :, :lo] == -7.0).all() and (c[0, :, :, hi + 1:] == -7.0).all()
    g = golden["repeat_kv_ramp"]  # test_repeat_kv.cu: cache[i] = i, ctx 2, layer 0
    ramp = torch.arange(int(np.prod(g["cache_shape"])), dtype=torch.float32, device=DEV).reshape(g["cache_shape"])
    dst = torch.zeros((1, g["head_num"], g["max_k_len"], g["cache_shape"][-1]), device=DEV)
    llmie.repeat_kv(ramp, dst, dev(np.array(g["ctx_len"], np.int32)), g["layer"])
    assert host(dst).reshape(-1).tolist() == g["expected"]


# --------------------------------------------------------------------------- softmax / transpose / swiglu
@pytest.mark.parametrize("dtype", DTYPES)
def test_scale_mask_softmax_reference_recipe(llmie, golden, dtype):
    g = golden["softmax_mod8"]  # test_scale_and_mask_and_softmax.cu: qk = i % 8, mask ones, scale 0.5
    bs, nh, ql, kl = g["shape"]
    qk = (torch.arange(bs * nh * ql * kl, device=DEV) % 8).to(dtype).reshape(bs, nh, ql, kl)
    out = torch.empty_like(qk)
    llmie.scale_mask_softmax(qk, torch.ones((bs, ql, kl), dtype=dtype, device=DEV), out, g["scale"])
    exp = np.broadcast_to(np.array(g["row"], np.float32), (bs, nh, ql, kl))
    assert np.abs(host(out) - exp).max() <= (g["tol"] if dtype == torch.float32 else 5e-4)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("bs,nh,ql,kl", [(1, 2, 8, 8), (2, 4, 32, 32), (1, 3, 5, 300), (1, 2, 3, 2048), (1, 1, 2, 5000)])
def test_scale_mask_softmax(llmie, dtype, bs, nh, ql, kl):
    # test_scale_and_mask_and_softmax.cu: bs=1,nh=2,q=k=8, scale=rsqrt(4)
    rng = np.random.default_rng(10)
    qk = rnd(rng, (bs, nh, ql, kl), 3.0, dtype)
    qlen = np.full(bs, ql, np.int32)
    klen = np.full(bs, kl, np.int32)
    qlen[-1] = max(1, ql - 1)
    mask = orc.build_causal_mask(qlen, klen, ql, kl)
    out = torch.empty((bs, nh, ql, kl), dtype=dtype, device=DEV)
    llmie.scale_mask_softmax(dev(qk, dtype), dev(mask, dtype), out, 0.5)
    exp = orc.scale_mask_softmax(qk, mask, 0.5)
    close(host(out), exp, *tol(dtype, f32=(1e-5, 1e-7), f16=(2e-3, 1e-6)))
    # in place
    qd = dev(qk, dtype)
    llmie.scale_mask_softmax(qd, dev(mask, dtype), qd, 0.5)
    assert torch.equal(qd, out)


@pytest.mark.parametrize("dtype", DTYPES)
def test_transpose_remove_padding(llmie, golden, dtype):
    g = golden["transpose_remove_padding"]
    src = np.arange(np.prod(g["shape"]), dtype=np.float32).reshape(g["shape"])
    dst = torch.empty((g["num_tokens"], g["shape"][1], g["shape"][3]), dtype=dtype, device=DEV)
    llmie.transpose_remove_padding(dev(src, dtype), dev(np.array(g["padding_offset"], np.int32)), dst)
    assert host(dst).reshape(-1).tolist() == g["expected"]
    rng = np.random.default_rng(11)
    lens, S, nh, hs = [5, 1, 8], 8, 4, 128
    off, _ = orc.cal_padding_offset(lens, S, fill=0)
    T = sum(lens)
    src = rnd(rng, (3, nh, S, hs), 1.0, dtype)
    dst = torch.empty((T, nh, hs), dtype=dtype, device=DEV)
    llmie.transpose_remove_padding(dev(src, dtype), dev(off.reshape(-1)[:T].copy()), dst)
    assert np.array_equal(host(dst), orc.transpose_remove_padding(src, off.reshape(-1)[:T], T))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,I", [(128, 11008), (1, 344), (3, 37)])
def test_silu_and_mul(llmie, golden, dtype, T, I):
    ones = torch.ones((T, 2, I), dtype=dtype, device=DEV)
    out = torch.empty((T, I), dtype=dtype, device=DEV)
    llmie.silu_and_mul(ones, out)
    g = golden["swiglu_ones"]  # test_silu_and_mul.cu: ones -> 0.7310586 (tol 1e-6 in fp32)
    assert np.abs(host(out) - g["expected"]).max() <= (g["tol"] if dtype == torch.float32 else 5e-4)
    rng = np.random.default_rng(12)
    x = rnd(rng, (T, 2, I), 2.0, dtype)
    llmie.silu_and_mul(dev(x, dtype), out)
    close(host(out), orc.silu_and_mul(x), *tol(dtype, f32=(2e-6, 1e-6), f16=(2e-3, 1e-4)))


# --------------------------------------------------------------------------- decode attention
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("L,layer,bs,nh,kvh,hs,max_seq,step,bias", [
    (1, 0, 1, 2, 2, 4, 4, 4, False),          # test_decoder_self_attention.cu shape (generic path)
    (1, 0, 1, 2, 2, 4, 8, 1, True),           # first token ever
    (2, 1, 2, 4, 4, 32, 64, 33, False),       # config A (hs=32)
    (2, 1, 1, 32, 32, 128, 256, 129, False),  # config B decode step at 129
    (1, 0, 2, 32, 32, 128, 2048, 2048, False),  # full context, max size
    (1, 0, 3, 8, 2, 128, 300, 257, True),     # GQA rep=4 + bias, ragged chunk tail
    (1, 0, 1, 16, 2, 64, 128, 1, False),      # rep=8, single token
    (1, 0, 2, 6, 2, 64, 64, 40, False)])      # rep=3 -> generic path
def test_decoder_mha(llmie, dtype, L, layer, bs, nh, kvh, hs, max_seq, step, bias):
    rng = np.random.default_rng(13)
    qkv = rnd(rng, (bs, nh + 2 * kvh, hs), 1.0, dtype)
    b = rnd(rng, ((nh + 2 * kvh) * hs,), 0.3, dtype) if bias else None
    kc = rnd(rng, (L, bs, kvh, max_seq, hs), 0.5, dtype)
    vc = rnd(rng, (L, bs, kvh, max_seq, hs), 0.5, dtype)
    kd, vd = dev(kc, dtype), dev(vc, dtype)
    out = torch.full((bs, nh * hs), 9.0, dtype=dtype, device=DEV)
    ws = torch.empty(max(1, llmie.decoder_mha_workspace_bytes(bs, nh, hs, max_seq) // 4), device=DEV)
    llmie.decoder_mha(dev(qkv, dtype), None if b is None else dev(b, dtype), kd, vd, out, layer, nh, kvh, step, ws)
    eo = orc.decoder_mha(qkv, b, kc, vc, layer, nh, kvh, hs, step)  # updates kc/vc in place
    # appended slot: k,v (+bias) rounded to the cache dtype; everything else bit-identical
    ekc = kc.astype(np.float16).astype(np.float32) if dtype == torch.float16 else kc
    evc = vc.astype(np.float16).astype(np.float32) if dtype == torch.float16 else vc
    close(host(kd), ekc, 0, 0 if (dtype == torch.float32 or not bias) else 1e-3)
    close(host(vd), evc, 0, 0 if (dtype == torch.float32 or not bias) else 1e-3)
    close(host(out), eo, *tol(dtype, f32=(1e-4, 1e-5), f16=(3e-3, 2e-3)))


def test_decoder_mha_device_step_matches_host_step(llmie):
    rng = np.random.default_rng(14)
    bs, nh, hs, max_seq, step = 2, 32, 128, 512, 300
    qkv = dev(rnd(rng, (bs, 3 * nh, hs), 1.0, torch.float16), torch.float16)
    kc = dev(rnd(rng, (1, bs, nh, max_seq, hs), 0.5, torch.float16), torch.float16)
    vc = dev(rnd(rng, (1, bs, nh, max_seq, hs), 0.5, torch.float16), torch.float16)
    ws = torch.empty(llmie.decoder_mha_workspace_bytes(bs, nh, hs, max_seq) // 4, device=DEV)
    o1 = torch.empty((bs, nh * hs), dtype=torch.float16, device=DEV)
    o2 = torch.empty_like(o1)
    k2, v2 = kc.clone(), vc.clone()
    llmie.decoder_mha(qkv, None, kc, vc, o1, 0, nh, nh, step, ws)
    llmie.decoder_mha(qkv, None, k2, v2, o2, 0, nh, nh, -1, ws, step_dev=dev(np.array([step], np.int32)))
    assert torch.equal(o1, o2) and torch.equal(kc, k2) and torch.equal(vc, v2)


def test_decoder_mha_workspace_too_small_is_an_error(llmie):
    qkv = torch.zeros((1, 96, 128), dtype=torch.float16, device=DEV)
    kc = torch.zeros((1, 1, 32, 256, 128), dtype=torch.float16, device=DEV)
    out = torch.zeros((1, 4096), dtype=torch.float16, device=DEV)
    with pytest.raises(llmie.LlmieError):
        llmie.decoder_mha(qkv, None, kc, kc.clone(), out, 0, 32, 32, 200, torch.empty(16, device=DEV))


# --------------------------------------------------------------------------- top-k / sampling
@pytest.mark.parametrize("dtype", DTYPES)
def test_topk_reference_case(llmie, golden, dtype):
    g = golden["topk_ramp"]  # test_topk.cu: probs[i]=i over [2,32000], K=5, 8 blocks per row
    rows, V, K = g["rows"], g["vocab"], g["K"]
    if dtype == torch.float16:
        probs = (np.arange(rows * V) % 2048).astype(np.float32).reshape(rows, V)  # fp16-exact integers, many ties
    else:
        probs = np.arange(rows * V, dtype=np.float32).reshape(rows, V)
    ids = torch.empty((rows, K), dtype=torch.int32, device=DEV)
    vals = torch.empty((rows, K), dtype=dtype, device=DEV)
    tid = torch.empty((rows, 8, K), dtype=torch.int32, device=DEV)
    tv = torch.empty((rows, 8, K), dtype=dtype, device=DEV)
    llmie.topk(dev(probs, dtype), tid, tv, ids, vals, blocks_per_row=8)
    eids, evals = orc.topk(probs, K)
    assert np.array_equal(host(ids), eids) and np.array_equal(host(vals), evals)
    if dtype == torch.float32:
        assert host(ids).tolist() == g["ids"] and host(vals).tolist() == g["vals"]


@pytest.mark.parametrize("rows,V,K,bpr", [(1, 32000, 4, 8), (3, 1000, 3, 1), (5, 777, 8, 3), (2, 32000, 20, 8),
                                          (1, 9, 5, 8), (130, 64, 1, 2)])
def test_topk_random_with_ties_and_negatives(llmie, rows, V, K, bpr):
    rng = np.random.default_rng(15)
    probs = np.round(rng.standard_normal((rows, V)) * 4).astype(np.float32)  # heavy ties, negative logits
    ids = torch.empty((rows, K), dtype=torch.int32, device=DEV)
    vals = torch.empty((rows, K), device=DEV)
    tid = torch.empty((rows, bpr, K), dtype=torch.int32, device=DEV)
    tv = torch.empty((rows, bpr, K), device=DEV)
    llmie.topk(dev(probs), tid, tv, ids, vals, blocks_per_row=bpr)
    eids, evals = orc.topk(probs, K)
    assert np.array_equal(host(ids), eids) and np.array_equal(host(vals), evals)


@pytest.mark.parametrize("dtype", DTYPES)
def test_sampling(llmie, dtype):
    # test_sampling.cu: bs=3, K=3, V=1000, step=6, end=10, val=K-1-(i%K), id=i
    bs, K, V, step, end = 3, 3, 1000, 6, 10
    tid = np.arange(bs * K, dtype=np.int32).reshape(bs, K)
    tv = (K - 1 - (np.arange(bs * K) % K)).astype(np.float32).reshape(bs, K)
    seq = np.full(bs, 4, np.int32)
    fin = np.zeros(bs, np.uint8)
    sd, fd, od = dev(seq), dev(fin), torch.empty(bs, dtype=torch.int32, device=DEV)
    llmie.sampling(dev(tid), dev(tv, dtype), sd, fd, od, step, end, V)
    eo, es, ef = orc.sampling(tid, tv, seq, fin, step, end, V)
    assert np.array_equal(host(od), eo) and np.array_equal(host(sd), es) and np.array_equal(host(fd).astype(bool), ef)
    # many rows / steps: the Philox stream and the pick agree with the oracle everywhere
    rng = np.random.default_rng(16)
    bs, K, V = 500, 5, 32000
    tid = rng.integers(0, 3 * V, (bs, K)).astype(np.int32)
    tv = -np.sort(-rnd(rng, (bs, K), 2.0, dtype), axis=1)
    fin = (rng.random(bs) < 0.3).astype(np.uint8)
    seq = rng.integers(1, 100, bs).astype(np.int32)
    for st in (1, 77, 2048):
        sd, fd, od = dev(seq), dev(fin), torch.empty(bs, dtype=torch.int32, device=DEV)
        llmie.sampling(dev(tid), dev(tv, dtype), sd, fd, od, st, 2, V)
        eo, es, ef = orc.sampling(tid, tv, seq, fin, st, 2, V)
        got = host(od)
        mism = (got != eo)
        assert mism.mean() <= 0.004, "device expf vs libm may flip a pick only on a threshold tie"
        assert np.array_equal(host(sd), es)
        assert np.array_equal(host(fd).astype(bool)[~mism], ef[~mism])
        # ... and every pick that differs must BE such a tie: with the oracle's own uniform number u
        # (the Philox stream is bit-identical on both sides) and the candidates' cumulative probabilities in float64, the device's
        # and the oracle's candidates are neighbours and u sits within a few float32 ulps of the boundary between them
        tvh = tv.astype(np.float16).astype(np.float64) if dtype == torch.float16 else tv.astype(np.float64)
        e = np.exp(tvh - tvh[:, :1])
        cum = np.cumsum(e, axis=1) / e.sum(axis=1, keepdims=True)
        for b in np.nonzero(mism)[0]:
            u = float(orc.lib().orc_uniform_philox(st, int(b)))
            cand = tid[b] % V
            gi, oi = np.nonzero(cand == got[b])[0], np.nonzero(cand == eo[b])[0]
            assert gi.size and oi.size, "row %d: the device picked %d, not one of its candidates" % (b, got[b])
            lo = min(gi.min(), oi.min())
            assert abs(int(gi.min()) - int(oi.min())) == 1 and abs(u - cum[b, lo]) <= 4e-6, (
                "row %d step %d: picks %d / %d are not a threshold tie (u %.9f, boundary %.9f)" % (b, st, got[b], eo[b], u, cum[b, lo]))


def test_sampling_distribution(llmie):
    # distribution-level parity (the reference's cuRAND stream is unpinned): frequencies ~ softmax(vals)
    bs, K, V = 20000, 4, 100
    tv = np.tile(np.array([2.0, 1.0, 0.0, -1.0], np.float32), (bs, 1))
    tid = np.tile(np.arange(K, dtype=np.int32), (bs, 1))
    od = torch.empty(bs, dtype=torch.int32, device=DEV)
    llmie.sampling(dev(tid), dev(tv), dev(np.zeros(bs, np.int32)), dev(np.zeros(bs, np.uint8)), od, 3, 99, V)
    freq = np.bincount(host(od), minlength=K)[:K] / bs
    p = np.exp(tv[0] - tv[0].max())
    p /= p.sum()
    assert np.abs(freq - p).max() < 0.015


# --------------------------------------------------------------------------- fused RoPE + attention + in-launch merge
def _rope_table(max_pos, hs, rot, base=10000.0):
    j = np.arange(hs // 2, dtype=np.float32)
    inv = np.power(np.float32(base), (2 * j) / np.float32(rot)).astype(np.float32)
    ang = (np.arange(max_pos, dtype=np.float32)[:, None] / inv[None, :]).astype(np.float32)
    tab = np.stack([np.cos(ang), np.sin(ang)], axis=-1).astype(np.float32)
    tab[:, rot // 2:, 0], tab[:, rot // 2:, 1] = 1.0, 0.0
    return tab


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("bs,nh,kvh,hs,max_seq,step,bias,rot", [
    (1, 32, 32, 128, 256, 129, False, 128), (2, 32, 32, 128, 2048, 2048, False, 128),
    (3, 8, 2, 128, 300, 257, True, 128), (2, 4, 4, 32, 64, 33, False, 32), (1, 8, 8, 64, 128, 100, True, 32)])
def test_decoder_mha_rope_fused_matches_oracle_sequence(llmie, dtype, bs, nh, kvh, hs, max_seq, step, bias, rot):
    """rope kernel + mha kernel (oracle: orc_rope_decode then orc_decoder_mha) == the fused launch"""
    rng = np.random.default_rng(21)
    qkv = rnd(rng, (bs, nh + 2 * kvh, hs), 1.0, dtype)
    b = rnd(rng, ((nh + 2 * kvh) * hs,), 0.3, dtype) if bias else None
    kc = rnd(rng, (1, bs, kvh, max_seq, hs), 0.5, dtype)
    vc = rnd(rng, (1, bs, kvh, max_seq, hs), 0.5, dtype)
    kd, vd = dev(kc, dtype), dev(vc, dtype)
    out = torch.empty((bs, nh * hs), dtype=dtype, device=DEV)
    ws = torch.empty(llmie.decoder_mha_workspace_bytes(bs, nh, hs, max_seq) // 4, device=DEV)
    tickets = torch.zeros(bs * kvh, dtype=torch.int32, device=DEV)
    tab = dev(_rope_table(max_seq, hs, rot))
    llmie.decoder_mha_rope(dev(qkv, dtype), None if b is None else dev(b, dtype), kd, vd, out, 0, nh, kvh, step, ws, tab,
                           rot, tickets)
    assert int(tickets.abs().sum()) == 0  # re-armed by the last arriver
    q_rot = orc.rope_decode(qkv, nh, kvh, hs, step, rot, 10000.0)
    if dtype == torch.float16:
        q_rot = q_rot.astype(np.float16).astype(np.float32)  # the unfused path stores the rotated q/k in fp16
    eo = orc.decoder_mha(q_rot, b, kc, vc, 0, nh, kvh, hs, step)
    close(host(out), eo, *tol(dtype, f32=(2e-4, 2e-5), f16=(3e-3, 2e-3)))
    # appended k row = rotated k: the numpy table above and the oracle's libm angle differ by ~1 ulp(angle) ~ step*6e-8
    close(host(kd), kc.astype(np.float16).astype(np.float32) if dtype == torch.float16 else kc, 0,
          2e-3 if dtype == torch.float16 else 3e-4 * max(1.0, step / 256))


def test_in_launch_merge_is_bit_identical_to_merge_kernel_under_load(llmie):
    """Hand-off hazard test (Guideline 16): many back-to-back launches re-using the same partial slabs and ticket
    words with fresh data each time, both XCD-spread (bs*kvh*splits workgroups) and L1/L2-warm; every output word of
    the in-launch merge must equal the separate merge kernel's, and the tickets must always return to zero."""
    rng = np.random.default_rng(22)
    bs, nh, hs, max_seq = 4, 32, 128, 2048
    ws1 = torch.empty(llmie.decoder_mha_workspace_bytes(bs, nh, hs, max_seq) // 4, device=DEV)
    ws2 = torch.empty_like(ws1)
    tickets = torch.zeros(bs * nh, dtype=torch.int32, device=DEV)
    tab = dev(_rope_table(max_seq, hs, hs))
    kc = dev(rnd(rng, (1, bs, nh, max_seq, hs), 0.5, torch.float16), torch.float16)
    vc = dev(rnd(rng, (1, bs, nh, max_seq, hs), 0.5, torch.float16), torch.float16)
    o1 = torch.empty((bs, nh * hs), dtype=torch.float16, device=DEV)
    o2 = torch.empty_like(o1)
    big = torch.empty(64 << 20, dtype=torch.float32, device=DEV)
    for it in range(60):
        step = int(rng.integers(130, max_seq + 1))
        qkv = torch.randn((bs, 3 * nh, hs), device=DEV).to(torch.float16)
        k1, v1, k2, v2 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
        if it % 3 == 0:
            big.add_(1.0)  # concurrent-ish streaming traffic right before the launch (uneven load, evictions)
        llmie.decoder_mha_rope(qkv, None, k1, v1, o1, 0, nh, nh, step, ws1, tab, hs, tickets)
        llmie.decoder_mha_rope(qkv, None, k2, v2, o2, 0, nh, nh, step, ws2, tab, hs, None)
        assert torch.equal(o1, o2), "iteration %d (step %d): in-launch merge differs" % (it, step)
        assert int(tickets.abs().sum()) == 0

"""Tile-packed weight images and the batch-decode linear on them (llmie_pack_weight / llmie_linear_packed).

No reference counterpart (the reference streams row-major weights through cuBLAS, src/kernels/linear.cu:10-87): the packer
is pinned bit-exactly by its numpy definition below, the linear by the oracle's fp32 linears over the SAME weights
(orc.linear / orc.linear_w8 / orc.linear_w4 restate CPUlinear, tests/unit_tests/test_linear.cu:17-33), with the fused
RMSNorm prologue / SwiGLU epilogue checked against orc.rmsnorm (rmsnorm.cu:35-80) and orc.silu_and_mul
(silu_and_mul.cu:25-41).  Tolerances: 2e-3 abs + 2e-3 rel per projection (fp32 accumulate, one fp16 rounding), as for the
other fp16 kernels."""
import numpy as np
import pytest
import torch

import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
F16 = torch.float16


def _h(a):
    return a.astype(np.float16).astype(np.float32)


def _d(a, dt=F16):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dt)


def _quant8_ref(w):
    s = (np.abs(w).max(axis=1) / np.float32(127.0)).astype(np.float16)
    s[s == 0] = np.float16(1.0)
    return np.clip(np.rint(w / s.astype(np.float32)[:, None]), -127, 127).astype(np.int8), s


def _quant4_ref(w, group=128):
    N, K = w.shape
    wg = w.reshape(N, K // group, group)
    s = (np.abs(wg).max(axis=2) / np.float32(7.0)).astype(np.float16)
    s[s == 0] = np.float16(1.0)
    q = np.clip(np.rint(wg / s.astype(np.float32)[:, :, None]), -8, 7).astype(np.int32).reshape(N, K) + 8
    return (q[:, 0::2] | (q[:, 1::2] << 4)).astype(np.uint8), s


def _src_rows(N, swiglu):
    """source row of (tile, r) for every packed tile; N = "no row" (zero fill)"""
    if not swiglu:
        tiles = (N + 15) // 16
        rows = np.arange(tiles * 16).reshape(tiles, 16)
        return np.where(rows < N, rows, N)
    inter = N // 2
    pairs = (inter + 15) // 16
    i = np.arange(pairs * 16).reshape(pairs, 16)
    gate = np.where(i < inter, i, N)
    up = np.where(i < inter, inter + i, N)
    return np.stack([gate, up], axis=1).reshape(2 * pairs, 16)


def pack_ref(w_bytes, N, K, kb, elem_bytes_num, elem_bytes_den, swiglu, nibbles=False):
    """numpy definition of the image: [tile][block][q][r][16 bytes]; w_bytes = row-major storage as uint8 [N, row_bytes]"""
    rows = _src_rows(N, swiglu)                      # [tiles, 16]
    tiles, nblk = rows.shape[0], K // kb
    src = np.concatenate([w_bytes, np.zeros((1, w_bytes.shape[1]), np.uint8)], axis=0)   # row N = zeros
    out = np.zeros((tiles, nblk, 4, 16, 16), np.uint8)
    steps = kb // 32                                 # MFMA k-steps per block: 32 k each, 8 per lane group q
    bps = 16 // steps                                # image bytes per (lane, step)
    for s in range(steps):
        for q in range(4):
            k0 = 32 * s + 8 * q                      # first of the 8 k values of (step, q) inside the block
            b0 = k0 * elem_bytes_num // elem_bytes_den
            nb = 8 * elem_bytes_num // elem_bytes_den
            for j in range(nblk):
                off = j * kb * elem_bytes_num // elem_bytes_den + b0
                chunk = src[rows][:, :, off:off + nb]        # [tiles, 16, nb]
                if nibbles:  # 8 nibbles k..k+7 -> image nibble positions (0,4,1,5,2,6,3,7)
                    v = chunk.astype(np.uint32)
                    word = v[..., 0] | (v[..., 1] << 8) | (v[..., 2] << 16) | (v[..., 3] << 24)
                    o = np.zeros_like(word)
                    for i, pos in enumerate((0, 4, 1, 5, 2, 6, 3, 7)):
                        o |= ((word >> np.uint32(4 * i)) & np.uint32(0xF)) << np.uint32(4 * pos)
                    chunk = np.stack([(o >> np.uint32(8 * b)) & np.uint32(0xFF) for b in range(4)], axis=-1).astype(np.uint8)
                out[:, j, q, :, s * bps:(s + 1) * bps] = chunk
    return out.reshape(-1)


FMT_PARAMS = {  # fmt -> (kb, bytes per element as a fraction)
    "f16": (32, 2, 1), "int8": (64, 1, 1), "fp8": (64, 1, 1), "int4": (128, 1, 2),
}


def _fmt_code(llmie, fmt):
    return {"f16": llmie.W_F16, "int8": llmie.W_INT8, "int4": llmie.W_INT4, "fp8": llmie.W_FP8}[fmt]


@pytest.mark.parametrize("fmt", ["f16", "int8", "fp8", "int4"])
@pytest.mark.parametrize("N,K,swiglu", [(48, 512, False), (40, 1024, False), (64, 512, True), (88, 1024, True)])
def test_pack_weight_bit_exact(llmie, fmt, N, K, swiglu):
    rng = np.random.default_rng(41)
    kb, num, den = FMT_PARAMS[fmt]
    raw = rng.integers(0, 256, (N, K * num // den), dtype=np.uint8)
    if fmt == "f16":
        w = torch.from_numpy(raw.view(np.float16).reshape(N, K)).to(DEV)
    else:
        w = torch.from_numpy(raw).to(DEV)
    sc = None
    if fmt == "int4":
        sc = torch.from_numpy(rng.standard_normal((N, K // 128)).astype(np.float16)).to(DEV)
    packed, pscale = llmie.pack_weight(_fmt_code(llmie, fmt), w, sc, swiglu)
    exp = pack_ref(raw, N, K, kb, num, den, swiglu, nibbles=(fmt == "int4"))
    assert np.array_equal(packed.cpu().numpy(), exp)
    if fmt == "int4":
        rows = _src_rows(N, swiglu)
        s = np.concatenate([sc.cpu().numpy(), np.zeros((1, K // 128), np.float16)], axis=0)
        exp_s = np.transpose(s[rows], (0, 2, 1))       # [tiles, nblk, 16]
        got = pscale.cpu().numpy().view(np.float16)[:exp_s.size].reshape(exp_s.shape)   # (+ 256 bytes of padding behind it)
        assert np.array_equal(got, exp_s)


def _make(fmt, rng, N, K):
    """row-major weights in fmt's storage + the fp32 weights the oracle multiplies with"""
    w = _h(rng.standard_normal((N, K)).astype(np.float32) / np.sqrt(K))
    if fmt == "f16":
        return dict(store=_d(w), scale=None, deq=w)
    if fmt == "int8":
        q, s = _quant8_ref(w)
        return dict(store=torch.from_numpy(q).to(DEV), scale=torch.from_numpy(s).to(DEV), deq=q.astype(np.float32) * s.astype(np.float32)[:, None])
    if fmt == "int4":
        q, s = _quant4_ref(w)
        nib = np.stack([(q & 0xF), (q >> 4)], axis=-1).reshape(N, K).astype(np.float32) - 8
        return dict(store=torch.from_numpy(q).to(DEV), scale=torch.from_numpy(s).to(DEV), deq=nib * np.repeat(s.astype(np.float32), 128, axis=1))
    raise AssertionError(fmt)


SHAPES = [(32, 4096, 1024), (17, 4096, 768), (5, 4096, 512), (16, 4096, 4096), (32, 11008, 512), (9, 11008, 4096),
          (32, 1024, 144), (32, 4096, 12288), (3, 512, 64)]


@pytest.mark.parametrize("fmt", ["int8", "f16", "int4"])
@pytest.mark.parametrize("M,K,N", SHAPES)
def test_linear_packed_matches_oracle(llmie, fmt, M, K, N):
    rng = np.random.default_rng(42)
    wt = _make(fmt, rng, N, K)
    x = _h(rng.standard_normal((M, K)).astype(np.float32))
    packed, pscale = llmie.pack_weight(_fmt_code(llmie, fmt), wt["store"], wt["scale"], False)
    y = torch.full((M, N), 7.0, dtype=F16, device=DEV)
    llmie.linear_packed(_fmt_code(llmie, fmt), _d(x), packed, pscale if fmt == "int4" else wt["scale"], y, N)
    exp = orc.linear(x, wt["deq"])
    err = np.abs(y.float().cpu().numpy() - exp)
    assert (err <= 2e-3 + 2e-3 * np.abs(exp)).all(), (err.max(), np.unravel_index(err.argmax(), err.shape))


@pytest.mark.parametrize("fmt", ["int8", "f16", "int4"])
@pytest.mark.parametrize("M,K,N", [(32, 4096, 4096), (20, 11008, 4096), (7, 4096, 256)])
def test_linear_packed_residual_in_place(llmie, fmt, M, K, N):
    """O / down projection form: y = x . W^T + residual with residual aliasing y (self_decoder.cpp:111)"""
    rng = np.random.default_rng(43)
    wt = _make(fmt, rng, N, K)
    x = _h(rng.standard_normal((M, K)).astype(np.float32))
    res = _h(rng.standard_normal((M, N)).astype(np.float32))
    packed, pscale = llmie.pack_weight(_fmt_code(llmie, fmt), wt["store"], wt["scale"], False)
    y = _d(res)
    llmie.linear_packed(_fmt_code(llmie, fmt), _d(x), packed, pscale if fmt == "int4" else wt["scale"], y, N, residual=y)
    exp = orc.linear(x, wt["deq"]) + res
    err = np.abs(y.float().cpu().numpy() - exp)
    assert (err <= 3e-3 + 2e-3 * np.abs(exp)).all(), err.max()


@pytest.mark.parametrize("fmt", ["int8", "f16", "int4"])
@pytest.mark.parametrize("M,K,I,pre_bias", [(32, 4096, 11008, False), (13, 4096, 1376, True), (32, 1024, 88, False)])
def test_linear_packed_norm_swiglu(llmie, fmt, M, K, I, pre_bias):
    """FFN front half in one launch: act = silu(h.Wg^T) * (h.Wu^T), h = rmsnorm(x + pre_bias) * gamma (ffn.cpp:105-122 behind
    add_residual_and_rmsnorm.cu:68-98's normalised branch)"""
    rng = np.random.default_rng(44)
    wt = _make(fmt, rng, 2 * I, K)
    x = _h(rng.standard_normal((M, K)).astype(np.float32) * 2)
    gamma = _h(1 + 0.1 * rng.standard_normal(K).astype(np.float32))
    pb = _h(0.1 * rng.standard_normal(K).astype(np.float32)) if pre_bias else None
    packed, pscale = llmie.pack_weight(_fmt_code(llmie, fmt), wt["store"], wt["scale"], True)
    y = torch.full((M, I), 7.0, dtype=F16, device=DEV)
    llmie.linear_packed(_fmt_code(llmie, fmt), _d(x), packed, pscale if fmt == "int4" else wt["scale"], y, 2 * I, swiglu=True,
                        gamma=_d(gamma), pre_bias=_d(pb) if pre_bias else None, eps=1e-5)
    xin = _h(x + pb[None, :]) if pre_bias else x
    hn, _ = orc.rmsnorm(xin.copy(), gamma, 1e-5)
    gu = orc.linear(_h(hn), wt["deq"])
    exp = orc.silu_and_mul(gu.reshape(M, 2, I))
    err = np.abs(y.float().cpu().numpy() - exp)
    assert (err <= 3e-3 + 3e-3 * np.abs(exp)).all(), (err.max(), np.unravel_index(err.argmax(), err.shape))


@pytest.mark.parametrize("fmt", ["int8", "f16", "int4"])
def test_linear_packed_norm_plain(llmie, fmt):
    """QKV form: qkv = rmsnorm(x) * gamma . Wqkv^T (self_decoder.cpp:77 + self_attention.cpp:79)"""
    rng = np.random.default_rng(45)
    M, K, N = 32, 4096, 12288
    wt = _make(fmt, rng, N, K)
    x = _h(rng.standard_normal((M, K)).astype(np.float32) * 3)
    gamma = _h(1 + 0.1 * rng.standard_normal(K).astype(np.float32))
    packed, pscale = llmie.pack_weight(_fmt_code(llmie, fmt), wt["store"], wt["scale"], False)
    y = torch.empty((M, N), dtype=F16, device=DEV)
    llmie.linear_packed(_fmt_code(llmie, fmt), _d(x), packed, pscale if fmt == "int4" else wt["scale"], y, N, gamma=_d(gamma), eps=1e-5)
    hn, _ = orc.rmsnorm(x.copy(), gamma, 1e-5)
    exp = orc.linear(_h(hn), wt["deq"])
    err = np.abs(y.float().cpu().numpy() - exp)
    assert (err <= 3e-3 + 3e-3 * np.abs(exp)).all(), err.max()


@pytest.mark.parametrize("M,K,N", [(13, 4096, 1024), (32, 4096, 12288), (27, 4096, 768)])
def test_linear_packed_fp8_matches_e4m3_emulation(llmie, M, K, N):
    """e4m3 weights x per-token e4m3 activations: y = wscale[n] * xscale[m] * sum_k wq xq (llmie_linear_fp8 semantics; the
    numpy emulation below is the one tests/test_quant_gpu.py uses for the other fp8 kernels)"""
    rng = np.random.default_rng(46)
    w = _h(rng.standard_normal((N, K)).astype(np.float32) / np.sqrt(K))
    x = _h(rng.standard_normal((M, K)).astype(np.float32))
    wq = torch.empty((N, K), dtype=torch.uint8, device=DEV)
    ws = torch.empty(N, dtype=torch.float32, device=DEV)
    llmie.quantize_fp8(_d(w), wq, ws)
    packed, _ = llmie.pack_weight(llmie.W_FP8, wq, None, False)
    y = torch.empty((M, N), dtype=F16, device=DEV)
    llmie.linear_packed(llmie.W_FP8, _d(x), packed, ws, y, N)
    # reference: the row-major fp8 linear of this library on the same quantised weights (itself pinned against the numpy e4m3
    # emulation in tests/test_quant_gpu.py)
    work = torch.empty(llmie.linear_fp8_workspace_bytes(M, K, N), dtype=torch.uint8, device=DEV)
    y2 = torch.empty((M, N), dtype=F16, device=DEV)
    llmie.linear_fp8(_d(x), wq, ws, y2, work)
    err = (y.float() - y2.float()).abs().cpu().numpy()
    ref = y2.float().abs().cpu().numpy()
    # The packed kernel quantises x * (1 / scale) where the row-major path divides.  Quotients of fp16 numbers land EXACTLY on
    # midpoints of the e4m3 grid now and then (round-half-even there), the product with the rounded reciprocal lands just beside
    # them: about one activation per ten rows takes the neighbouring code, and every output of that row moves by
    # (one e4m3 step of that activation) x (its weight).  So: most outputs agree to fp16 rounding, every row agrees to well under
    # the e4m3 quantisation noise itself (~3 % of |y|).
    ok = err <= 3e-3 + 3e-3 * ref
    row_rel = np.linalg.norm(err, axis=1) / np.linalg.norm(ref, axis=1)
    assert ok.mean() > 0.9 and row_rel.max() < 1e-2, (1 - ok.mean(), row_rel.max())


def test_linear_packed_fp8_k_split_quantises_per_slice(llmie):
    """K beyond the register slice (7B down projection): the launch splits K over workgroups and every slice quantises its part of
    a token's activations with its own amax -- a finer e4m3 grid than one scale per token, so the result is not bit-comparable
    with llmie_linear_fp8; it has to track the product of the SAME e4m3 weights with the unquantised activations as closely as
    the per-token form does"""
    rng = np.random.default_rng(47)
    M, K, N = 32, 11008, 4096
    w = _h(rng.standard_normal((N, K)).astype(np.float32) / np.sqrt(K))
    x = _h(rng.standard_normal((M, K)).astype(np.float32))
    wq = torch.empty((N, K), dtype=torch.uint8, device=DEV)
    ws = torch.empty(N, dtype=torch.float32, device=DEV)
    llmie.quantize_fp8(_d(w), wq, ws)
    packed, _ = llmie.pack_weight(llmie.W_FP8, wq, None, False)
    res = _h(rng.standard_normal((M, N)).astype(np.float32))
    y = _d(res)
    llmie.linear_packed(llmie.W_FP8, _d(x), packed, ws, y, N, residual=y)
    work = torch.empty(llmie.linear_fp8_workspace_bytes(M, K, N), dtype=torch.uint8, device=DEV)
    y2 = torch.empty((M, N), dtype=F16, device=DEV)
    llmie.linear_fp8(_d(x), wq, ws, y2, work, residual=_d(res))
    # exact product of the de-quantised weights with the fp16 activations
    b = np.arange(256)
    sgn, ex, mant = b >> 7, (b >> 3) & 0xF, b & 7                               # OCP e4m3fn decode table
    tab = np.where(ex == 0, (mant / 8.0) * 2.0 ** -6, (1 + mant / 8.0) * 2.0 ** (ex.astype(np.float64) - 7))
    tab = np.where(sgn == 1, -tab, tab).astype(np.float32)
    wdeq = tab[wq.cpu().numpy()] * ws.cpu().numpy()[:, None]
    exact = orc.linear(x, wdeq) + res
    e_split = np.linalg.norm(y.float().cpu().numpy() - exact) / np.linalg.norm(exact)
    e_token = np.linalg.norm(y2.float().cpu().numpy() - exact) / np.linalg.norm(exact)
    assert e_split <= 1.1 * e_token + 1e-3 and e_split < 0.03, (e_split, e_token)


def _x32_ref(a, C):
    """numpy definition of the x32 image of a [M <= 32, C] fp16 matrix: [C/32][2][4 q][16 r][8] halves, zero rows past M"""
    M = a.shape[0]
    full = np.zeros((32, C), np.float16)
    full[:M] = a
    # (t, r, kstep, q, e) -> (kstep, t, q, r, e)
    return np.ascontiguousarray(full.reshape(2, 16, C // 32, 4, 8).transpose(2, 0, 3, 1, 4)).reshape(-1)


@pytest.mark.parametrize("M,C", [(32, 4096), (5, 96), (17, 11008)])
def test_x32_convert_bit_exact(llmie, M, C):
    rng = np.random.default_rng(47)
    a = rng.standard_normal((M, C)).astype(np.float16)
    img = torch.full((32 * C,), 3.0, dtype=F16, device=DEV)
    llmie.x32_convert(torch.from_numpy(a).to(DEV), img, M, C, True)
    assert np.array_equal(img.cpu().numpy(), _x32_ref(a, C))
    back = torch.zeros((M, C), dtype=F16, device=DEV)
    llmie.x32_convert(img, back, M, C, False)
    assert np.array_equal(back.cpu().numpy(), a)


@pytest.mark.parametrize("fmt", ["int8", "f16", "int4"])
@pytest.mark.parametrize("M,K,N,mode", [(32, 4096, 4096, "resid"), (21, 4096, 12288, "norm"), (32, 4096, 22016, "swiglu"),
                                        (32, 11008, 4096, "resid"), (9, 4096, 512, "plain"),
                                        # block-major first pass over 2 / 3 / 4 tiles, then tile-major left-overs; waves that own
                                        # fewer blocks than their register slice holds (K 2048), SwiGLU with one unit per workgroup
                                        (32, 4096, 8192, "plain"), (16, 4096, 8192, "swiglu"), (32, 4096, 10240, "norm"),
                                        (32, 4096, 32000, "plain"), (32, 2048, 12288, "plain"), (27, 2048, 16384, "resid"),
                                        (32, 5120, 15360, "resid"), (32, 3072, 24576, "swiglu")])
def test_linear_packed_x32_layout_is_bit_identical(llmie, fmt, M, K, N, mode):
    """the x32 activation layout changes addresses, not arithmetic: x / y / residual in x32 give exactly the row-major result"""
    rng = np.random.default_rng(48)
    wt = _make(fmt, rng, N, K)
    code = _fmt_code(llmie, fmt)
    x = _h(rng.standard_normal((M, K)).astype(np.float32))
    gamma = _d(_h(1 + 0.1 * rng.standard_normal(K).astype(np.float32)))
    swiglu = mode == "swiglu"
    outN = N // 2 if swiglu else N
    res = _h(rng.standard_normal((M, outN)).astype(np.float32))
    packed, pscale = llmie.pack_weight(code, wt["store"], wt["scale"], swiglu)
    wscale = pscale if fmt == "int4" else wt["scale"]
    kw = dict(swiglu=swiglu, gamma=gamma if mode in ("norm", "swiglu") else None, eps=1e-5)
    y_rm = _d(res) if mode == "resid" else torch.zeros((M, outN), dtype=F16, device=DEV)
    llmie.linear_packed(code, _d(x), packed, wscale, y_rm, N, residual=y_rm if mode == "resid" else None, **kw)
    # same call with every activation operand in x32 (the residual in place in the x32 output buffer)
    x_img = torch.empty(32 * K, dtype=F16, device=DEV)
    llmie.x32_convert(_d(x), x_img, M, K, True)
    y_img = torch.zeros(32 * outN, dtype=F16, device=DEV)
    flags = llmie.X32_X | llmie.X32_Y
    if mode == "resid":
        llmie.x32_convert(_d(res), y_img, M, outN, True)
        flags |= llmie.X32_RES
    llmie.linear_packed(code, x_img, packed, wscale, y_img, N, residual=y_img if mode == "resid" else None, M=M, K=K,
                        x32_flags=flags, **kw)
    back = torch.zeros((M, outN), dtype=F16, device=DEV)
    llmie.x32_convert(y_img, back, M, outN, False)
    assert torch.equal(back, y_rm)

"""The prefill path AT THE BENCH'S OWN SIZES: one Llama-2-7B-geometry layer (H = 4096, I = 11008, 32 heads)
prefilled as 1 x 2048 and as 8 x 512 tokens -- the eight-phase 256 x 256 / 256 x 128 GEMM plans, the two-launch SwiGLU plan and the
flash kernel with 16 query tiles -- against the oracle's composition of the reference kernels (context_decoder.cpp:58-199) on the
same inputs, for fp16 weights and for weight-only int8 / int4 (oracle on the DE-QUANTISED weights; README.md:36-39 plans them,
the reference holds no quantised kernel).  The oracle evaluates 80 sampled token rows of the layer output (oracle.py
context_decoder_rows: K and V of ALL tokens, everything else for the sampled rows -- a whole layer is ~0.8 TFLOP on the host) and
every K / V cache row.  (8 x 512 = 4096 tokens, 1 x 2048 = 2048: the bench's two prefill configurations.)"""
import numpy as np
import pytest
import torch

import oracle as orc
from conftest import systematic_error

pytestmark = pytest.mark.gpu
DEV, F16 = "cuda", torch.float16
NH, HS, I, GROUP = 32, 128, 11008, 128
H, QKV = NH * HS, 3 * NH * HS
# the bounds of tests/test_prefill_gpu.py (small models): element-wise fp16 pipeline tolerance, relative Frobenius error and
# projection of the error on the signal (a 1 % gain error of the layer gives 1e-2 on both)
FRO, PROJ = 3e-3, 2e-4


def _h(a):
    return a.astype(np.float16).astype(np.float32)


def _quant8(w):
    amax = np.abs(w).max(axis=1)
    s = (amax / np.float32(127.0)).astype(np.float16)
    s[s == 0] = np.float16(1.0)
    q = np.clip(np.rint(w / s.astype(np.float32)[:, None]), -127, 127).astype(np.int8)
    return q, s, q.astype(np.float32) * s.astype(np.float32)[:, None]


def _quant4(w, group):
    N, K = w.shape
    wg = w.reshape(N, K // group, group)
    s = (np.abs(wg).max(axis=2) / np.float32(7.0)).astype(np.float16)
    s[s == 0] = np.float16(1.0)
    q = np.clip(np.rint(wg / s.astype(np.float32)[:, :, None]), -8, 7).astype(np.int32)
    deq = (q.astype(np.float32) * s.astype(np.float32)[:, :, None]).reshape(N, K)
    q = q.reshape(N, K) + 8
    return (q[:, 0::2] | (q[:, 1::2] << 4)).astype(np.uint8), s, deq


@pytest.fixture(scope="module")
def weights():
    rng = np.random.default_rng(77)
    u = lambda n, k: _h(rng.uniform(-1, 1, (n, k)).astype(np.float32) * 2 / np.sqrt(k))
    return dict(qkv=u(QKV, H), o=u(H, H), gate_up=u(2 * I, H), down=u(H, I),
                attn_norm=_h(rng.uniform(0.8, 1.2, H).astype(np.float32)), ffn_norm=_h(rng.uniform(0.8, 1.2, H).astype(np.float32)),
                x=_h(rng.standard_normal((4096, H)).astype(np.float32)))


@pytest.mark.parametrize("fmt", ["f16", "int8", "int4"])
def test_one_7b_layer_prefill_at_bench_sizes_matches_oracle(llmie, weights, fmt):
    W = weights
    d = lambda a: torch.from_numpy(a).to(DEV)
    eng, ol = dict(attn_norm=d(W["attn_norm"]).to(F16), ffn_norm=d(W["ffn_norm"]).to(F16)), \
        dict(attn_norm=W["attn_norm"], ffn_norm=W["ffn_norm"], qkv_bias=None, o_bias=None)
    for name in ("qkv", "o", "gate_up", "down"):
        if fmt == "f16":
            eng[name], ol[name] = dict(data=d(W[name]).to(F16)), W[name]
        else:
            q, s, deq = _quant8(W[name]) if fmt == "int8" else _quant4(W[name], GROUP)
            eng[name], ol[name] = dict(data=d(q), scale=d(s)), deq
    max_seq = 2048
    cfg = dict(head_num=NH, kv_head_num=NH, head_size=HS, inter_size=I, num_layers=1, vocab_size=100, max_seq_len=max_seq, max_batch=8,
               rotary_dim=HS, rotary_base=10000.0, rms_eps=1e-5, dtype=llmie.F16,
               wfmt=dict(f16=llmie.W_F16, int8=llmie.W_INT8, int4=llmie.W_INT4)[fmt], int4_group=GROUP)
    dec = llmie.Decoder(cfg, [eng])
    ocfg = dict(head_num=NH, kv_head_num=NH, head_size=HS, inter_size=I, rms_eps=1e-5, rotary_dim=HS, rotary_base=10000.0)
    rng = np.random.default_rng(78)
    kv_proj = []   # the oracle's K / V projection of the 4096 rows (the 1 x 2048 case uses its first 2048: the projection is row-wise)
    for lens in ([512] * 8, [2048]):
        bs, T = len(lens), sum(lens)
        hist = [0] * bs
        x = W["x"][:T]
        xd = d(x).to(F16)
        kd = torch.zeros((1, bs, NH, max_seq, HS), dtype=F16, device=DEV)
        vd = torch.zeros_like(kd)
        out = torch.empty_like(xd)
        dec.prefill(xd, out, kd, vd, torch.tensor(lens, dtype=torch.int32, device=DEV), torch.tensor(hist, dtype=torch.int32, device=DEV),
                    max(lens))
        # sampled rows: the edges of the 256-row GEMM tiles and of the 128-row query tiles, sequence starts / ends, random others
        fixed = [0, 1, 15, 16, 127, 128, 255, 256, 511, 512, 1023, 1024, 1791, 1792, 2046, 2047, T - 2, T - 1]
        rows = np.array(sorted(set(fixed) | set(rng.choice(T, 64, replace=False).tolist())))
        kc = np.zeros((1, bs, NH, max_seq, HS), np.float32)
        vc = np.zeros_like(kc)
        exp = orc.context_decoder_rows(ocfg, ol, x, kc, vc, lens, hist, rows, kv_proj=kv_proj[0][:T] if kv_proj else kv_proj)
        got = out.float().cpu().numpy()[rows]
        err = np.abs(got - exp)
        fro, proj = systematic_error(got, exp)
        kerr = np.abs(kd.float().cpu().numpy() - kc).max()
        verr = np.abs(vd.float().cpu().numpy() - vc).max()
        print("%s %dx%d: max err %.4g (|exp| max %.3g), rel Frobenius %.3g, projection %.3g, K cache %.3g, V cache %.3g"
              % (fmt, bs, lens[0], err.max(), np.abs(exp).max(), fro, proj, kerr, verr))
        assert (err <= 3e-2 + 3e-2 * np.abs(exp)).all(), "max err %g (|exp| max %g)" % (err.max(), np.abs(exp).max())
        assert fro <= FRO and proj <= PROJ, "relative Frobenius error %.3g, projection on the signal %.3g" % (fro, proj)
        # every K / V row of every token: the QKV GEMM's K / V columns + RoPE + append at full size (fp16 rounding of O(1) values)
        assert kerr <= 2e-2 and verr <= 2e-2
    dec.close()

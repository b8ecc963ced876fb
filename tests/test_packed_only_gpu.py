"""LLMIE_DEC_PACKED_ONLY (round 3): a quantised engine whose tile-packed images are the ONLY copy of its layer
matrices -- the row-major `data` arrays handed to llmie_decoder_create are overwritten with garbage right after create, then the
engine decodes (every batch on the packed kernels) and prefills (unpack pass + fp16 GEMM).  Against the default engine on the same
weights: bit-identical wherever both run the packed kernels (batch above the GEMV range), fp16-pipeline tolerance where the default
engine runs its GEMV / prefill kernels (those are pinned to the oracle by tests/test_quant_gpu.py, test_prefill_gpu.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV, F16 = "cuda", torch.float16
NH, HS, I, L = 32, 128, 11008, 2
H, QKV = NH * HS, 3 * NH * HS


def _layers(llmie, rng, wfmt):
    u = lambda shape, s: torch.from_numpy((rng.uniform(-1, 1, shape) * s).astype(np.float32)).to(DEV).to(F16)

    def q(w):
        n, k = w.shape
        if wfmt == "f16":
            return dict(data=w)
        if wfmt == "int8":
            d, sc = torch.empty((n, k), dtype=torch.int8, device=DEV), torch.empty(n, dtype=F16, device=DEV)
            llmie.quantize_w8(w, d, sc)
        else:
            d, sc = torch.empty((n, k // 2), dtype=torch.uint8, device=DEV), torch.empty((n, k // 128), dtype=F16, device=DEV)
            llmie.quantize_w4(w, d, sc, 128)
        return dict(data=d, scale=sc)

    return [dict(attn_norm=u((H,), 0.2) + 1, ffn_norm=u((H,), 0.2) + 1, qkv=q(u((QKV, H), 2 / np.sqrt(H))), o=q(u((H, H), 2 / np.sqrt(H))),
                 gate_up=q(u((2 * I, H), 2 / np.sqrt(H))), down=q(u((H, I), 2 / np.sqrt(I)))) for _ in range(L)]


@pytest.mark.parametrize("wfmt", ["int8", "int4", "f16"])
def test_packed_only_engine_runs_without_the_row_major_weights(llmie, wfmt):
    rng = np.random.default_rng(57)
    layers = _layers(llmie, rng, wfmt)
    fmt = dict(f16=llmie.W_F16, int8=llmie.W_INT8, int4=llmie.W_INT4)[wfmt]
    gemv_max = dict(f16=5, int8=2, int4=2)[wfmt]
    max_seq, maxb = 384, 32
    cfg = dict(head_num=NH, kv_head_num=NH, head_size=HS, inter_size=I, num_layers=L, vocab_size=100, max_seq_len=max_seq, max_batch=maxb,
               rotary_dim=HS, rotary_base=10000.0, rms_eps=1e-5, dtype=llmie.F16, wfmt=fmt, int4_group=128)
    ref = llmie.Decoder(cfg, layers)
    # the packed-only engine gets COPIES of the matrices, which are trashed as soon as create has returned
    copies = [{k: (dict(v, data=v["data"].clone()) if isinstance(v, dict) else v) for k, v in lw.items()} for lw in layers]
    po = llmie.Decoder(dict(cfg, flags=llmie.DEC_PACKED_ONLY), copies)
    for lw in copies:
        for k in ("qkv", "o", "gate_up", "down"):
            lw[k]["data"].fill_(0x55 if lw[k]["data"].dtype != F16 else 7.0)
    torch.cuda.synchronize()
    g = torch.Generator(device="cpu").manual_seed(5)
    for bs in (1, 2, 3, 5, 17, 32):
        kc = (torch.randn((L, maxb, NH, max_seq, HS), generator=g) * 0.5).to(DEV).to(F16)
        vc = (torch.randn((L, maxb, NH, max_seq, HS), generator=g) * 0.5).to(DEV).to(F16)
        x = torch.randn((bs, H), generator=g).to(DEV).to(F16)
        k2, v2 = kc[:, :bs].contiguous(), vc[:, :bs].contiguous()
        k1, v1 = kc[:, :bs].contiguous(), vc[:, :bs].contiguous()
        a = ref.forward(x, torch.empty_like(x), k1, v1, 200)
        b = po.forward(x, torch.empty_like(x), k2, v2, 200)
        assert torch.isfinite(b.float()).all()
        if bs > gemv_max:
            assert torch.equal(a, b) and torch.equal(k1, k2), "batch %d: packed-only differs from the default engine's packed path" % bs
        else:   # default engine: GEMV kernels; packed-only: MFMA kernels on the image
            d = (a.float() - b.float()).abs()
            assert bool((d <= 3e-2 + 3e-2 * a.float().abs()).all()), "batch %d: max diff %g" % (bs, d.max().item())
            rel = ((a.float() - b.float()).norm() / a.float().norm()).item()
            assert rel < 5e-3, rel
    # prefill: unpack pass + fp16 GEMM against the default engine's prefill kernels
    for T in (300,):
        xs = torch.randn((T, H), generator=g).to(DEV).to(F16)
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)
        mk = lambda: torch.zeros((L, 1, NH, max(max_seq, T), HS), dtype=F16, device=DEV)
        k1, v1, k2, v2 = mk(), mk(), mk(), mk()
        a = ref.prefill(xs, torch.empty_like(xs), k1, v1, i32([T]), i32([0]), T).float()
        b = po.prefill(xs, torch.empty_like(xs), k2, v2, i32([T]), i32([0]), T).float()
        d = (a - b).abs()
        assert bool((d <= 3e-2 + 3e-2 * a.abs()).all()), "prefill %d: max diff %g" % (T, d.max().item())
        assert ((a - b).norm() / a.norm()).item() < 5e-3
        assert (k1.float() - k2.float()).abs().max().item() <= 2e-2
    # llmie_decoder_repack: new weights into the same engine -> what a fresh engine on those weights computes
    layers2 = _layers(llmie, np.random.default_rng(58), wfmt)
    ref2 = llmie.Decoder(cfg, layers2)
    po.repack([{k: (dict(v, data=v["data"].clone()) if isinstance(v, dict) else v) for k, v in lw.items()} for lw in layers2])
    bs = 17
    kc = (torch.randn((L, bs, NH, max_seq, HS), generator=g) * 0.5).to(DEV).to(F16)
    vc = (torch.randn((L, bs, NH, max_seq, HS), generator=g) * 0.5).to(DEV).to(F16)
    x = torch.randn((bs, H), generator=g).to(DEV).to(F16)
    a = ref2.forward(x, torch.empty_like(x), kc.clone(), vc.clone(), 150)
    b = po.forward(x, torch.empty_like(x), kc.clone(), vc.clone(), 150)
    assert torch.equal(a, b), "repacked engine differs from a fresh engine on the new weights"
    ref.close()
    ref2.close()
    po.close()


def test_resident_weight_bytes(llmie):
    q = llmie.lib().llmie_decoder_resident_weight_bytes
    base = dict(head_num=32, kv_head_num=32, head_size=128, inter_size=11008, num_layers=32, vocab_size=32000, max_seq_len=2048, max_batch=32,
                rotary_dim=128, rotary_base=10000.0, rms_eps=1e-5, dtype=llmie.F16, int4_group=128)
    cfgs = lambda **kw: C.byref(llmie.DecoderConfig(**dict(base, **kw)))
    elems = 32 * (3 * 4096 * 4096 + 4096 * 4096 + 3 * 4096 * 11008)
    i8, i8_po, i8_nc = q(cfgs(wfmt=llmie.W_INT8)), q(cfgs(wfmt=llmie.W_INT8, flags=llmie.DEC_PACKED_ONLY)), q(cfgs(wfmt=llmie.W_INT8, flags=llmie.DEC_NO_PACKED_COPY))
    assert 2 * elems <= i8 <= 2.02 * elems          # row-major + image
    assert elems <= i8_po <= 7.5e9                   # the image alone (+ scales): the 7B decoder in ~6.5 GB
    assert elems <= i8_nc <= 1.01 * elems            # row-major alone
    i4_po = q(cfgs(wfmt=llmie.W_INT4, flags=llmie.DEC_PACKED_ONLY))
    assert elems // 2 <= i4_po <= 4.5e9
    assert q(cfgs(wfmt=llmie.W_F16, max_batch=1)) == 2 * elems   # GEMV-range engine: no image

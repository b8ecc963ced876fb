"""BASELINE.json full-size checks (Llama-2-7B: 32 layers, H=4096, I=11008, V=32000, fp16, max_seq 2048) through
size-independent properties -- the oracle would need minutes for one token at this size:
  * determinism / idempotence: the same decode step run twice writes the same KV row and returns the same bits;
  * hipGraph replay == eager launches, bit for bit (device-resident step);
  * batch symmetry: two identical sequences in one batch give identical rows and identical sampled tokens;
  * prefill(n+1)[-1] == prefill(n) -> decode(n+1) through all 32 layers (fp16 tolerance);
  * int8 weight-only engine tracks the fp16 engine on the same weights (quantisation error bound)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV, F16 = "cuda", torch.float16


@pytest.fixture(scope="module")
def full(llmie):
    import bench
    cfg = dict(bench.LLAMA2_7B)
    weights = bench.build_weights(torch, cfg, seed=7)
    return bench, cfg, weights


def _decoder(full, llmie, wfmt, layers, batch, max_seq=2048):
    bench, cfg, weights = full
    return bench.make_decoder(torch, llmie, cfg, weights, layers, wfmt, batch, max_seq)


def test_decode_step_is_deterministic_and_idempotent(llmie, full):
    _, cfg, weights = full
    dec, kc, vc = _decoder(full, llmie, "f16", weights["layers"], 1)
    H = 4096
    x = torch.randn((1, H), device=DEV).to(F16)
    step = 2048  # last slot of the full context
    o1 = dec.forward(x, torch.empty_like(x), kc, vc, step)
    k_after = kc[:, :, :, step - 1].clone()
    o2 = dec.forward(x, torch.empty_like(x), kc, vc, step)
    assert torch.equal(o1, o2) and torch.equal(kc[:, :, :, step - 1], k_after)
    assert torch.isfinite(o1.float()).all()
    dec.close()


def test_graph_replay_equals_eager(llmie, full):
    _, cfg, weights = full
    dec, kc, vc = _decoder(full, llmie, "f16", weights["layers"], 1, max_seq=512)
    H = 4096
    x = torch.randn((1, H), device=DEV).to(F16)
    step_dev = torch.tensor([300], dtype=torch.int32, device=DEV)
    k0, v0 = kc.clone(), vc.clone()
    eager = dec.forward(x, torch.empty_like(x), kc, vc, -1, step_dev=step_dev).clone()
    k_eager = kc.clone()
    kc.copy_(k0)
    vc.copy_(v0)
    out = torch.empty_like(x)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        dec.forward(x, out, kc, vc, -1, step_dev=step_dev)  # warm-up on the capture stream
    s.synchronize()
    kc.copy_(k0)
    vc.copy_(v0)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        dec.forward(x, out, kc, vc, -1, step_dev=step_dev)
    kc.copy_(k0)
    vc.copy_(v0)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager) and torch.equal(kc, k_eager)
    # the graph follows the device-resident step
    step_dev.fill_(301)
    g.replay()
    torch.cuda.synchronize()
    # same kernels, same inputs (the cache as the step-300 launch left it), the step passed by value this time: bit-identical,
    # and the token row of step 301 landed in the graph's cache
    k_ref, v_ref = k_eager.clone(), vc.clone()
    ref = dec.forward(x, torch.empty_like(x), k_ref, v_ref, 301)
    assert torch.isfinite(out.float()).all() and torch.equal(out, ref)
    assert torch.equal(kc, k_ref) and torch.equal(vc, v_ref)
    assert not torch.equal(kc[:, :, :, 300], k_eager[:, :, :, 300])
    dec.close()


def test_batch_rows_are_symmetric(llmie, full):
    _, cfg, weights = full
    dec, kc, vc = _decoder(full, llmie, "f16", weights["layers"], 2, max_seq=256)
    kc[:, 1] = kc[:, 0]
    vc[:, 1] = vc[:, 0]
    x = torch.randn((1, 4096), device=DEV).to(F16).repeat(2, 1).contiguous()
    out = dec.forward(x, torch.empty_like(x), kc, vc, 200)
    assert torch.equal(out[0], out[1])
    assert torch.equal(kc[:, 0], kc[:, 1])
    dec.close()


def test_prefill_then_decode_consistency_32_layers(llmie, full):
    _, cfg, weights = full
    dec, kc, vc = _decoder(full, llmie, "f16", weights["layers"], 1, max_seq=256)
    n = 130
    ids = torch.randint(0, 32000, (n + 1,), dtype=torch.int32, device=DEV)
    xs = torch.empty((n + 1, 4096), dtype=F16, device=DEV)
    llmie.input_embedding(ids, weights["embed"], xs)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)
    k1, v1 = torch.zeros_like(kc), torch.zeros_like(vc)
    k2, v2 = torch.zeros_like(kc), torch.zeros_like(vc)
    full_out = dec.prefill(xs, torch.empty_like(xs), k1, v1, i32([n + 1]), i32([0]), n + 1)
    dec.prefill(xs[:n].contiguous(), torch.empty((n, 4096), dtype=F16, device=DEV), k2, v2, i32([n]), i32([0]), n)
    last = dec.forward(xs[n:n + 1].contiguous(), torch.empty((1, 4096), dtype=F16, device=DEV), k2, v2, n + 1)
    a, b = last.float(), full_out[n:n + 1].float()
    rel = (a - b).norm() / b.norm()
    assert rel.item() < 2e-2, rel.item()  # 32 layers of fp16 activations through two different kernel families
    # the caches of the two routes: rows 0..n-1 come from two prefills of different row counts (other tile plans), row n from
    # the prefill in one and the decode kernels in the other; every layer's K / V rows are fp16 roundings of projections of
    # hidden states that agree to the fp16 decoder tolerance
    for a_c, b_c, nm in ((k1, k2, "k"), (v1, v2, "v")):
        af, bf = a_c[:, :, :, :n + 1].float(), b_c[:, :, :, :n + 1].float()
        d = (af - bf).abs()
        relf = ((af - bf).flatten(1).norm(dim=1) / bf.flatten(1).norm(dim=1)).max().item()
        print("%s cache: max abs diff %.4g, worst per-layer rel Frobenius %.4g" % (nm, d.max().item(), relf))
        assert relf < 1e-2, (nm, relf)
        assert (d <= 2e-2 + 2e-2 * bf.abs()).all(), (nm, d.max().item())
    dec.close()


def test_int8_engine_tracks_fp16_engine(llmie, full):
    bench, cfg, weights = full
    q8 = bench.quantize_layers(torch, llmie, weights["layers"], "int8")
    d16, kc, vc = _decoder(full, llmie, "f16", weights["layers"], 1, max_seq=256)
    d8, _, _ = _decoder(full, llmie, "int8", q8, 1, max_seq=256)
    x = torch.randn((1, 4096), device=DEV).to(F16)
    o16 = d16.forward(x, torch.empty_like(x), kc.clone(), vc.clone(), 129)
    o8 = d8.forward(x, torch.empty_like(x), kc.clone(), vc.clone(), 129)
    rel = (o8.float() - o16.float()).norm() / o16.float().norm()
    assert rel.item() < 0.10, rel.item()  # per-row int8 weight noise (~0.4 % per GEMM) through 32 layers
    d16.close()
    d8.close()


@pytest.mark.parametrize("wfmt,batch,kv8", [("int8", 32, False), ("f16", 16, True), ("fp8", 24, False)])
def test_batch_path_graph_replay_equals_eager_and_rows_are_symmetric(llmie, full, wfmt, batch, kv8):
    """fused batch decode path (split-K slabs in the library's scratch, consumed by the attention / row kernels): a captured
    graph replays bit-identically to the eager launches, identical sequences in a batch give identical rows, all 32 layers"""
    bench, cfg, weights = full
    layers = weights["layers"] if wfmt == "f16" else bench.quantize_layers(torch, llmie, weights["layers"], wfmt)
    dec, kc, vc = bench.make_decoder(torch, llmie, cfg, weights, layers, wfmt, batch, 384, kv8)
    kc[:, 1:] = kc[:, :1]
    vc[:, 1:] = vc[:, :1]
    x = torch.randn((1, 4096), device=DEV).to(F16).repeat(batch, 1).contiguous()
    step_dev = torch.tensor([260], dtype=torch.int32, device=DEV)
    k0, v0 = kc.clone(), vc.clone()
    eager = dec.forward(x, torch.empty_like(x), kc, vc, -1, step_dev=step_dev).clone()
    k_eager = kc.clone()
    assert torch.isfinite(eager.float()).all()
    for b in range(1, batch):
        assert torch.equal(eager[b], eager[0])
    assert torch.equal(kc[:, 1], kc[:, 0])
    out = torch.empty_like(x)
    s = torch.cuda.Stream()
    kc.copy_(k0); vc.copy_(v0)
    with torch.cuda.stream(s):
        dec.forward(x, out, kc, vc, -1, step_dev=step_dev)
    s.synchronize()
    kc.copy_(k0); vc.copy_(v0)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        dec.forward(x, out, kc, vc, -1, step_dev=step_dev)
    kc.copy_(k0); vc.copy_(v0)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager) and torch.equal(kc, k_eager)
    dec.close()

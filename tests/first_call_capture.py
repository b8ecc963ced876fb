"""Run by tests/test_workspace_gpu.py in a FRESH process: the very first launches this process makes of

  * llmie_linear / llmie_linear_w8a16 / llmie_linear_fp8 at 128 rows (split-K over caller-owned slabs), and
  * llmie_decoder_forward at batch 128 (the engine's split-K batch path)

are recorded into a hipGraph.  Nothing on the compute path may allocate (an allocation while the stream is capturing is an
error), so the capture succeeds only if every scratch byte comes from the caller's workspaces.  The replay is then compared with
the same calls run eagerly.  Prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import load_llmie  # noqa: E402

llmie = load_llmie()
DEV = "cuda"
F16 = torch.float16
M, K, N = 128, 4096, 4096
g = torch.Generator(device="cpu").manual_seed(7)
x = (torch.randn((M, K), generator=g) * 0.5).to(DEV).to(F16)
w = (torch.randn((N, K), generator=g) / np.sqrt(K)).to(DEV).to(F16)
wq = torch.empty((N, K), dtype=torch.int8, device=DEV)
sc = torch.empty(N, dtype=F16, device=DEV)
f8 = torch.empty((N, K), dtype=torch.uint8, device=DEV)
f8s = torch.empty(N, dtype=torch.float32, device=DEV)
llmie.quantize_w8(w, wq, sc)       # (not a split-K launch: no scratch of any kind)
llmie.quantize_fp8(w, f8, f8s)
ws16 = torch.empty(llmie.linear_workspace_bytes(llmie.W_F16, M, K, N), dtype=torch.uint8, device=DEV)
ws8 = torch.empty(llmie.linear_workspace_bytes(llmie.W_INT8, M, K, N), dtype=torch.uint8, device=DEV)
wsf = torch.empty(llmie.linear_fp8_workspace_bytes(M, K, N), dtype=torch.uint8, device=DEV)
assert ws16.numel() > 0 and ws8.numel() > 0 and wsf.numel() > M * K

# a small decoder whose batch-128 step takes the split-K batch path
nh, hs, I, L, bs, max_seq = 8, 64, 768, 2, 128, 48
H, QKV = nh * hs, 3 * nh * hs


def t(shape, scale):
    return (torch.rand(shape, generator=g) * 2 - 1).mul(scale).to(DEV).to(F16)


layers = [dict(attn_norm=t((H,), 0.2) + 1, ffn_norm=t((H,), 0.2) + 1, qkv=dict(data=t((QKV, H), 2 / np.sqrt(H))),
               o=dict(data=t((H, H), 2 / np.sqrt(H))), gate_up=dict(data=t((2 * I, H), 2 / np.sqrt(H))),
               down=dict(data=t((H, I), 2 / np.sqrt(I)))) for _ in range(L)]
cfg = dict(head_num=nh, kv_head_num=nh, head_size=hs, inter_size=I, num_layers=L, vocab_size=1000, max_seq_len=max_seq,
           max_batch=bs, rotary_dim=hs, rotary_base=10000.0, rms_eps=1e-5, dtype=llmie.F16, wfmt=llmie.W_F16, int4_group=128)
dec = llmie.Decoder(cfg, layers)   # create: the one place that may set things up (off the compute path)
kc0, vc0 = t((L, bs, nh, max_seq, hs), 0.5), t((L, bs, nh, max_seq, hs), 0.5)
hin = t((bs, H), 1.0)
step = torch.tensor([17], dtype=torch.int32, device=DEV)

y16, y8, yf = (torch.zeros((M, N), dtype=F16, device=DEV) for _ in range(3))
hout = torch.zeros((bs, H), dtype=F16, device=DEV)
kc, vc = kc0.clone(), vc0.clone()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.graph(graph, stream=s):   # FIRST launches of this process, under capture
    llmie.linear(x, w, y16, workspace=ws16)
    llmie.linear_w8a16(x, wq, sc, y8, workspace=ws8)
    llmie.linear_fp8(x, f8, f8s, yf, wsf)
    dec.forward(hin, hout, kc, vc, -1, step_dev=step)
graph.replay()
torch.cuda.synchronize()
got = [a.clone() for a in (y16, y8, yf, hout)]

e16, e8, ef = (torch.zeros((M, N), dtype=F16, device=DEV) for _ in range(3))
eh = torch.zeros((bs, H), dtype=F16, device=DEV)
kc2, vc2 = kc0.clone(), vc0.clone()
llmie.linear(x, w, e16, workspace=ws16)
llmie.linear_w8a16(x, wq, sc, e8, workspace=ws8)
llmie.linear_fp8(x, f8, f8s, ef, wsf)
dec.forward(hin, eh, kc2, vc2, -1, step_dev=step)
torch.cuda.synchronize()
ref = (x.float() @ w.float().t())
out = dict(
    equal=[bool(torch.equal(a, b)) for a, b in zip(got, (e16, e8, ef, eh))],
    nonzero=[bool(a.abs().max().item() > 0) for a in got],
    f16_err=float((got[0].float() - ref).abs().max().item()),
    kv_equal=bool(torch.equal(kc, kc2) and torch.equal(vc, vc2)),
    # without a workspace fp16 still answers (non-split kernels); int8 at 128 rows has only the split-K form and must say so
    no_ws_f16_err=float((llmie.linear(x, w, torch.zeros_like(e16), workspace=None).float() - ref).abs().max().item()),
)
try:
    llmie.linear_w8a16(x, wq, sc, torch.zeros_like(e8), workspace=None)
    out["no_ws_int8"] = "ran"
except llmie.LlmieError as e:
    out["no_ws_int8"] = str(e)
try:
    llmie.linear(x, w, torch.zeros_like(e16), workspace=ws16[:4096])
    out["small_ws"] = "ran"
except llmie.LlmieError as e:
    out["small_ws"] = str(e)
dec.close()
print(json.dumps(out))

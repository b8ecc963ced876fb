"""dev: chain launch determinism at 7B geometry (eager x3, then vs LLMIE_NO_CHAIN from a second process via npz)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.conftest import load_llmie
llmie = load_llmie()
DEV, F16 = "cuda", torch.float16
wfmt = sys.argv[1] if len(sys.argv) > 1 else "int8"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rng = np.random.default_rng(23)
nh, hs, I, max_seq = 32, 128, 11008, 160
H, QKV = nh * hs, 3 * nh * hs
u = lambda shape, s: torch.from_numpy((rng.uniform(-1, 1, shape) * s).astype(np.float32)).to(DEV).to(F16)
def quantised(w):
    n, k = w.shape
    if wfmt == "f16":
        return dict(data=w)
    q, sc = torch.empty((n, k), dtype=torch.int8, device=DEV), torch.empty(n, dtype=F16, device=DEV)
    llmie.quantize_w8(w, q, sc)
    return dict(data=q, scale=sc)
layers = [dict(attn_norm=u((H,), 0.2) + 1, ffn_norm=u((H,), 0.2) + 1, qkv=quantised(u((QKV, H), 2 / np.sqrt(H))), o=quantised(u((H, H), 2 / np.sqrt(H))),
               gate_up=quantised(u((2 * I, H), 2 / np.sqrt(H))), down=quantised(u((H, I), 2 / np.sqrt(I)))) for _ in range(L)]
fmt = {"f16": llmie.W_F16, "int8": llmie.W_INT8}[wfmt]
res = {}
for bs in (5, 17, 32):
    cfg = dict(head_num=nh, kv_head_num=nh, head_size=hs, inter_size=I, num_layers=L, vocab_size=100, max_seq_len=max_seq, max_batch=bs,
               rotary_dim=hs, rotary_base=10000.0, rms_eps=1e-5, dtype=llmie.F16, wfmt=fmt, int4_group=128)
    dec = llmie.Decoder(cfg, layers)
    g = torch.Generator(device="cpu").manual_seed(bs)
    kc = (torch.randn((L, bs, nh, max_seq, hs), generator=g) * 0.5).to(DEV).to(F16)
    vc = (torch.randn((L, bs, nh, max_seq, hs), generator=g) * 0.5).to(DEV).to(F16)
    x = torch.randn((bs, H), generator=g).to(DEV).to(F16)
    k0, v0 = kc.clone(), vc.clone()
    outs = []
    for rep in range(4):
        kc.copy_(k0); vc.copy_(v0)
        o = dec.forward(x, torch.empty_like(x), kc, vc, 130).clone()
        dec.status()
        outs.append(o)
    for rep in range(1, 4):
        d = (outs[rep].float() - outs[0].float()).abs()
        nz = d.nonzero()
        print("bs %d rep %d: max diff %.4g, differing elements %d, rows %s cols[min,max] %s" % (
            bs, rep, d.max().item(), nz.shape[0], sorted(set(nz[:, 0].tolist()))[:8], (nz[:, 1].min().item(), nz[:, 1].max().item()) if nz.shape[0] else None))
    res["b%d" % bs] = outs[0].float().cpu().numpy()
    dec.close()
if len(sys.argv) > 3:
    np.savez(sys.argv[3], **res)

# graph replay vs eager (device-resident step)
for bs in (5, 32):
    cfg = dict(head_num=nh, kv_head_num=nh, head_size=hs, inter_size=I, num_layers=L, vocab_size=100, max_seq_len=max_seq, max_batch=bs,
               rotary_dim=hs, rotary_base=10000.0, rms_eps=1e-5, dtype=llmie.F16, wfmt=fmt, int4_group=128)
    dec = llmie.Decoder(cfg, layers)
    g = torch.Generator(device="cpu").manual_seed(bs)
    kc = (torch.randn((L, bs, nh, max_seq, hs), generator=g) * 0.5).to(DEV).to(F16)
    vc = (torch.randn((L, bs, nh, max_seq, hs), generator=g) * 0.5).to(DEV).to(F16)
    x = torch.randn((bs, H), generator=g).to(DEV).to(F16)
    k0, v0 = kc.clone(), vc.clone()
    step_dev = torch.tensor([130], dtype=torch.int32, device=DEV)
    e_host = dec.forward(x, torch.empty_like(x), kc, vc, 130).clone()
    kc.copy_(k0); vc.copy_(v0)
    e_dev = dec.forward(x, torch.empty_like(x), kc, vc, -1, step_dev=step_dev).clone()
    print("bs %d eager host-step vs device-step equal:" % bs, torch.equal(e_host, e_dev))
    y = torch.empty_like(x)
    s = torch.cuda.Stream()
    kc.copy_(k0); vc.copy_(v0)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        dec.forward(x, y, kc, vc, -1, step_dev=step_dev)
    s.synchronize()
    print("bs %d side-stream eager equal:" % bs, torch.equal(y, e_host))
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        dec.forward(x, y, kc, vc, -1, step_dev=step_dev)
    for rep in range(3):
        kc.copy_(k0); vc.copy_(v0); y.zero_()
        torch.cuda.synchronize()
        gr.replay()
        torch.cuda.synchronize()
        d = (y.float() - e_host.float()).abs()
        nz = d.nonzero()
        print("bs %d replay %d: equal %s max diff %.4g n %d rows %s" % (bs, rep, torch.equal(y, e_host), d.max().item(), nz.shape[0], sorted(set(nz[:, 0].tolist()))[:8]))
    try:
        dec.status()
        print("status ok")
    except Exception as ex:
        print("status:", ex)
    dec.close()

"""debugging aid: where do the fused and the two-launch prefill differ?  usage: qkv_rope_diff.py <case>"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
case = sys.argv[1]
outs = []
for tag, env in (("fused", {}), ("plain", {"LLMIE_NO_QKV_ROPE_FUSION": "1"})):
    o = "/tmp/qr_%s.npz" % tag
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "qkv_rope_probe.py"), o, case], env=dict(os.environ, **env), cwd=ROOT)
    outs.append(dict(np.load(o)))
f, p = outs
for key in ("k", "v"):
    a, b = f[case + "/" + key], p[case + "/" + key]
    if a.dtype == np.int16:
        a, b = a.view(np.float16).astype(np.float32), b.view(np.float16).astype(np.float32)
    d = a != b
    print(key, a.shape, "differ", int(d.sum()), "max abs diff", float(np.abs(a - b).max()), "max |b|", float(np.abs(b).max()))
    # layer 0 only: later layers inherit
    d0 = d[0]
    print("  layer0 differ", int(d0.sum()), "by head-dim (d) counts:", d0.sum(axis=(0, 1, 2)).tolist())
    print("  by kv head:", d0.sum(axis=(0, 2, 3)).tolist())
    rows = d0.sum(axis=(1, 3))
    print("  by batch x position (nonzero positions):", [(int(i), int(j)) for i, j in zip(*np.nonzero(rows))][:20], "... total", int((rows > 0).sum()))
    idx = np.argwhere(d0)[:8]
    for i in idx:
        print("   ", tuple(int(x) for x in i), a[0][tuple(i)], b[0][tuple(i)])

"""RoPE + KV-cache append as the epilogue of the prefill's QKV projection (round 3; context_attention.cpp:158-205 -- the QKV GEMM,
launchFusedQKVAddBiasAndTransposeAndRope and launchConcatKVCache -- as one launch sequence: gemm8p.cuh ROPE forms).  The fused
epilogue performs prefill_rope_append_kernel's arithmetic on the fp16-rounded accumulator, so a pass with the fusion must be
BIT-identical to the pass with LLMIE_NO_QKV_ROPE_FUSION=1 (projection + rope/append launch, which the other prefill tests pin to
the oracle): hidden states and every byte of both caches, including the rows neither pass may touch -- for fp16 / int8 / int4 /
e4m3 weights, fp16 and e4m3 caches, dense and paged layouts, ragged batches with history, GQA, QKV bias and a partial rotary_dim;
and for short prefills (<= 128 tokens), where the QKV projection's split-K slab consumer takes the RoPE + append in
(prefill.hip splitk_finalize_qkv_rope_kernel)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, name, env_extra):
    out = os.path.join(str(tmp_path), name + ".npz")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "qkv_rope_probe.py"), out], capture_output=True, text=True, timeout=900,
                       cwd=ROOT, env=dict(os.environ, **env_extra))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return dict(np.load(out))


def test_fused_qkv_rope_append_is_bit_identical_to_the_two_launches(tmp_path):
    fused = _run(tmp_path, "fused", {})
    plain = _run(tmp_path, "plain", {"LLMIE_NO_QKV_ROPE_FUSION": "1"})
    assert set(fused) == set(plain) and len(fused) >= 42
    bad = []
    for k in sorted(fused):
        a, b = fused[k], plain[k]
        if not np.array_equal(a, b):
            bad.append("%s: %d of %d elements differ" % (k, int((a != b).sum()), a.size))
    if bad:
        pytest.fail("\n".join(bad))

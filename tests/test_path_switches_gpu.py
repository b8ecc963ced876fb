"""Every environment switch the library still reads selects a whole alternative launch sequence (the tuning knobs of round 1
are gone: constants now).  Each switch is exercised here: the alternative path gives the default path's result within fp16
rounding (different kernels, different summation order), so none of them is dead or stale code."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWITCHES = [
    ("LLMIE_NO_FUSED_DECODE", "f16"),         # per-kernel launch sequence of the reference instead of the fused decode paths
    ("LLMIE_NO_FUSED_BATCH", "f16"),          # batch > gemv_max: GEMV / unfused sequence instead of the split-K batch path
    ("LLMIE_NO_PACKED_BATCH", "int8"),        # 4 < batch <= 32: split-K batch path instead of the packed-weight path
    ("LLMIE_NO_FUSED_SHORT_PREFILL", "f16"),  # <= 128 tokens: prefill-sized launch sequence instead of the slab-fused one
    ("LLMIE_NO_NORM_QUANT", "fp8"),           # fp8 prefill: RMSNorm + quantise as two launches
    ("LLMIE_CHAIN", "int8"),                  # opt-in: 4 < batch <= 32 as attention + ONE persistent chain launch per layer instead of six launches
]
# read by the library as well, with a test of its own (prefill-sized passes, bit-identity): tests/test_qkv_rope_fusion_gpu.py
ELSEWHERE = {"LLMIE_NO_QKV_ROPE_FUSION"}


def _run(tmp_path, name, wfmt, env_extra):
    out = os.path.join(str(tmp_path), name + ".npz")
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "path_switch_probe.py"), out, wfmt], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return dict(np.load(out))


@pytest.mark.parametrize("switch,wfmt", SWITCHES, ids=[s for s, _ in SWITCHES])
def test_switched_path_matches_the_default_path(tmp_path, switch, wfmt):
    base = _run(tmp_path, "default", wfmt, {})
    alt = _run(tmp_path, "alt", wfmt, {switch: "1"})
    assert set(base) == set(alt) and base
    for k in base:
        a, b = base[k], alt[k]
        assert np.isfinite(a).all() and np.isfinite(b).all()
        tol = 6e-2 if wfmt == "fp8" else 3e-2
        assert (np.abs(a - b) <= tol + tol * np.abs(a)).all(), "%s / %s: max diff %g" % (switch, k, np.abs(a - b).max())
        if switch == "LLMIE_CHAIN":   # the chain's phases ARE the launches' code (pk_phase): same arithmetic, same order
            assert np.array_equal(a, b), "%s / %s: the chain launch is not bit-identical to the launch sequence" % (switch, k)


def test_no_other_switches_are_read():
    """the source reads exactly the switches tested above (a new getenv needs a test here)"""
    import glob
    import re
    found = set()
    for f in glob.glob(os.path.join(ROOT, "llm-inference-engine_amd", "csrc", "*.*")):
        if f.endswith((".hip", ".cuh", ".h", ".cpp")):
            found |= set(re.findall(r'getenv\("(LLMIE_[A-Z0-9_]+)"\)', open(f).read()))
    assert found == {s for s, _ in SWITCHES} | ELSEWHERE, found

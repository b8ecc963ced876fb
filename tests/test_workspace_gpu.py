"""Nothing on the compute path allocates: every scratch byte of the split-K projections is the caller's (round-1 review:
`splitk_scratch()` grew a library-owned buffer with hipMalloc on first use, which a first call under hipGraph capture cannot do)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_first_call_at_batch_128_is_captured_into_a_graph():
    """fresh process: its first split-K launches (fp16 / int8 / fp8 linear at 128 rows, decoder step at batch 128) are recorded
    into a hipGraph, replayed, and equal the eager launches bit for bit"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "first_call_capture.py")], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["equal"] == [True, True, True, True], out
    assert out["nonzero"] == [True, True, True, True], out
    assert out["kv_equal"], out
    assert out["f16_err"] < 0.05 and out["no_ws_f16_err"] < 0.05, out
    assert "workspace" in out["no_ws_int8"], out       # only a split-K form at 128 rows: says what it needs
    assert "workspace too small" in out["small_ws"], out


def test_workspace_queries_cover_the_plans(llmie):
    # zero where no split-K form exists, positive for decode batches; the fp8 query adds its activation part
    assert llmie.linear_workspace_bytes(llmie.W_F16, 1000, 4096, 4096) == 0
    assert llmie.linear_workspace_bytes(llmie.W_F16, 128, 4096, 12288) >= 128 * 12288 * 4
    assert llmie.linear_workspace_bytes(llmie.W_INT8, 32, 11008, 4096) >= 32 * 4096 * 4
    assert llmie.linear_workspace_bytes(llmie.W_INT4, 32, 4096, 4096) >= 32 * 4096 * 4
    assert llmie.linear_workspace_bytes(llmie.W_F16, 64, 500, 4096) == 0       # K not a multiple of the sub-block
    a = llmie.linear_fp8_workspace_bytes(64, 4096)
    assert a >= 64 * 4096 + 64 * 4
    assert llmie.linear_fp8_workspace_bytes(64, 4096, 4096) >= a + 64 * 4096 * 4

"""Runs the C++ API-mirror drivers (llm-inference-engine_amd/cpp_tests) on the GPU: the reference's unit tests
replayed through the `launch*` templates, and the layer classes / LlamaModel chat flow against the oracle."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "llm-inference-engine_amd", "cpp_tests")


@pytest.mark.parametrize("exe", ["test_kernels_api", "test_layers_api"])
@pytest.mark.parametrize("fp16", [False, True], ids=["fp32", "fp16"])
def test_cpp_driver(exe, fp16):
    path = os.path.join(BIN, exe)
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", BIN, exe])
    r = subprocess.run([path] + (["1"] if fp16 else []), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "all passed" in r.stdout and "FAIL" not in r.stdout

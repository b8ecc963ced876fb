import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def load_llmie():
    """import llm-inference-engine_amd/ (hyphenated directory) as module `llmie_amd`"""
    import importlib.util
    if "llmie_amd" in sys.modules:
        return sys.modules["llmie_amd"]
    spec = importlib.util.spec_from_file_location(
        "llmie_amd", os.path.join(ROOT, "llm-inference-engine_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["llmie_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def llmie():
    return load_llmie()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "known_answers.json")) as f:
        return json.load(f)


def systematic_error(got, exp):
    """(relative Frobenius error, |projection of the error on the expected signal|): element-wise bounds of a few fp16 ulps
    per stage cannot see a 1 % gain error on a whole layer -- these two can (a scale error s gives a projection of |s|, and
    both are ~1e-3 for correct fp16 pipelines)"""
    import numpy as np
    got = np.asarray(got, np.float64).ravel()
    exp = np.asarray(exp, np.float64).ravel()
    den = float(np.dot(exp, exp)) + 1e-30
    d = got - exp
    return float(np.sqrt(np.dot(d, d) / den)), abs(float(np.dot(d, exp)) / den)

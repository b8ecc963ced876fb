#!/usr/bin/env python3
"""Development tool: bench.py --only <spec> against another build of the library (LLMIE_LIB=path) -- for A/B runs in one gpurun call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
_load = bench.load_llmie


def load():
    m = _load()
    if os.environ.get("LLMIE_LIB"):
        m.LIB_PATH = os.environ["LLMIE_LIB"]
    return m


bench.load_llmie = load
sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()

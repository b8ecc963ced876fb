"""Builds llm-inference-engine_amd/lib/libllmie.so for gfx950 with hipcc (in-tree, no JIT cache).

    python llm-inference-engine_amd/build.py [--force] [-j N]

Each .hip/.cpp under csrc/ is compiled to an object only when it (or a header) is newer than
the object; objects live in csrc/_obj/ (git-ignored).  hipcc cross-compiles without a GPU.
"""
import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libllmie.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".cuh", ".hpp"))]
    hs.append(os.path.join(HERE, "..", "include", "llmie.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force, hdr_m):
    s = os.path.join(CSRC, src)
    o = os.path.join(OBJ, src + ".o")
    if (not force and os.path.exists(o) and os.path.getmtime(o) >= max(os.path.getmtime(s), hdr_m)):
        return o, False
    cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", s, "-o", o]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return o, True


def build(force=False, jobs=4, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hdr_m = _headers_mtime()
    srcs = _sources()
    objs, rebuilt = [], False
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        for o, did in ex.map(lambda s: _compile(s, force, hdr_m), srcs):
            objs.append(o)
            rebuilt |= did
    if rebuilt or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("linked", LIB)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-j", type=int, default=4)
    a = ap.parse_args()
    print(build(a.force, a.j, verbose=True))

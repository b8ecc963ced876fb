#!/usr/bin/env python3
"""Reads the AMDGPU code-object metadata of a built object / library and lists per-kernel registers, spills, scratch, LDS.

    python tools/check_kernel_resources.py llm-inference-engine_amd/csrc/_obj/pk_linear.hip.o [substring]

The packed-weight kernels (pk_mfma_kernel) issue their global loads from inline asm with hand-counted waits: a register spill
there can store a register whose load is still in flight.  tests/test_abi_cpu.py asserts that none of them spills."""
import re
import subprocess
import sys

CLANG_OFFLOAD = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def kernel_metadata(path):
    """[{name, vgpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds}] for every kernel of the gfx950 code object inside `path`"""
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        co = os.path.join(td, "dev.co")
        kind = "o" if path.endswith(".o") else "so"
        # host objects / shared libraries carry the device code object in a fat-binary bundle
        r = subprocess.run([CLANG_OFFLOAD, "--type=" + ("o" if kind == "o" else "o"), "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--input=" + path, "--output=" + co, "--unbundle"], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            # fall back: extract the .hip_fatbin section first
            fat = os.path.join(td, "fat.bin")
            subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, path, os.path.join(td, "x")])
            subprocess.check_call([CLANG_OFFLOAD, "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat,
                                   "--output=" + co, "--unbundle"])
        notes = subprocess.check_output([READELF, "--notes", co], text=True)
    out = []
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        def g(key, default="0"):
            m = re.search(r"\." + key + r":\s+(\S+)", blk)
            return m.group(1) if m else default
        out.append(dict(name=g("name", "?"), vgpr=int(g("vgpr_count")), sgpr=int(g("sgpr_count")), vgpr_spill=int(g("vgpr_spill_count")),
                        sgpr_spill=int(g("sgpr_spill_count")), scratch=int(g("private_segment_fixed_size")), lds=int(g("group_segment_fixed_size"))))
    return out


if __name__ == "__main__":
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in kernel_metadata(sys.argv[1]):
        if sub in k["name"]:
            print("%-90s vgpr %3d sgpr %3d spill v%d s%d scratch %d" % (k["name"][:90], k["vgpr"], k["sgpr"], k["vgpr_spill"], k["sgpr_spill"], k["scratch"]))

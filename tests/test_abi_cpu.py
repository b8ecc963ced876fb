"""CPU-side checks of the C-ABI boundary: the library builds/loads, exports every symbol that
include/llmie.h declares, and rejects bad arguments before touching the GPU (no compute here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(llmie):
    llmie.build()
    return llmie.lib()


def _declared():
    txt = open(os.path.join(ROOT, "include", "llmie.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(llmie_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(lib, llmie):
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libllmie.so does not export %s" % n
    # and the python signature table covers the header exactly
    assert sorted(llmie.EXPORTS) == names


def test_integration_notes_name_every_entry():
    """INTEGRATION.md maps each exported entry to the reference interface it stands for (whole names, in code spans or code blocks)."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    named = set(re.findall(r"\bllmie_[a-z0-9_]+", doc))
    # `llmie_linear_w8a16/w4a16/fp8`-style shorthands in the launcher table count for their expansions
    for m in re.finditer(r"\b(llmie_[a-z0-9]+_)([a-z0-9]+(?:/[a-z0-9_]+)+)", doc):
        named.update(m.group(1) + tail for tail in m.group(2).split("/"))
    missing = [n for n in _declared() if n not in named]
    assert not missing, "INTEGRATION.md does not mention: %s" % ", ".join(missing)


def test_version_and_arch(lib, llmie):
    assert lib.llmie_abi_version() == 3 == llmie.ABI_VERSION
    assert "#define LLMIE_ABI_VERSION 3" in open(os.path.join(ROOT, "include", "llmie.h")).read()
    assert lib.llmie_target_arch() == b"gfx950"


def test_code_object_is_gfx950(llmie):
    data = open(llmie.LIB_PATH, "rb").read()
    assert b"gfx950" in data
    for other in (b"gfx942", b"gfx90a", b"sm_86"):
        assert other not in data


def test_invalid_arguments_fail_loudly(lib):
    # NULL pointers / bad shapes are rejected on the host with a message (reference: LLM_CHECK throws)
    assert lib.llmie_rmsnorm(None, None, None, 1e-6, 4, 8, 0, None) == -1
    assert b"rmsnorm" in lib.llmie_last_error()
    assert lib.llmie_linear(None, None, None, 1, 1, 1, 1, None, None, 1, None, 0, None) == -1
    assert lib.llmie_add_residual(None, None, 0, 0, 0, None) == -1
    one = C.c_void_p(16)  # never dereferenced: shape check fails first
    assert lib.llmie_topk(one, one, one, one, one, 1, 100, 64, 8, 0, None) == -1
    assert b"K=64" in lib.llmie_last_error()
    assert lib.llmie_decoder_mha(one, None, one, one, one, 0, 1, 3, 2, 8, 16, 1, None, None, 0, 0, None) == -1
    assert b"kv_head_num" in lib.llmie_last_error()
    assert lib.llmie_rmsnorm(one, None, one, 1e-6, 4, 8, 7, None) == -2  # unknown dtype


def test_workspace_queries(lib):
    assert lib.llmie_decoder_mha_workspace_bytes(1, 32, 128, 2048) == 1 * 32 * 64 * 130 * 4
    assert lib.llmie_decoder_mha_workspace_bytes(0, 32, 128, 2048) == 0


def test_missing_library_raises(llmie, monkeypatch):
    monkeypatch.setattr(llmie, "_lib", None)
    monkeypatch.setattr(llmie, "LIB_PATH", "/nonexistent/libllmie.so")
    with pytest.raises(llmie.LlmieError):
        llmie.lib()


def test_packed_kernels_do_not_spill():
    """pk_mfma_kernel issues its global loads from inline asm with hand-counted waits: a spilled register there could be one whose
    load is still in flight, so no instantiation may use scratch (checked on the built code object, no GPU needed)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("ckr", os.path.join(root, "tools", "check_kernel_resources.py"))
    ckr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ckr)
    obj = os.path.join(root, "llm-inference-engine_amd", "csrc", "_obj", "pk_linear.hip.o")
    ks = [k for k in ckr.kernel_metadata(obj) if "pk_mfma_kernel" in k["name"]]
    assert len(ks) >= 24
    bad = [k for k in ks if k["vgpr_spill"] or k["scratch"]]
    assert not bad, bad
    assert all(k["vgpr"] <= 256 for k in ks)


def test_chain_kernels_use_no_scratch():
    """pk_chain_kernel runs pk_phase's hand-counted DMA waits: no instantiation the host will launch may touch scratch memory (the
    32-row int4 one does and is excluded by pk_chain_begin)"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("ckr", os.path.join(root, "tools", "check_kernel_resources.py"))
    ckr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ckr)
    obj = os.path.join(root, "llm-inference-engine_amd", "csrc", "_obj", "pk_linear.hip.o")
    ks = [k for k in ckr.kernel_metadata(obj) if "pk_chain_kernel" in k["name"]]
    assert len(ks) == 8
    bad = [k for k in ks if (k["vgpr_spill"] or k["scratch"]) and "ILi2ELi4E" not in k["name"]]
    assert not bad, bad


def test_eight_phase_gemm_kernels_do_not_spill():
    """gemm8p.cuh counts its LDS-DMA by hand (inline asm) and runs one 512-thread workgroup per CU, 2 waves per
    SIMD: every instantiation (fp16 / e4m3, with and without epilogue, SwiGLU, both tile widths) must fit 256 VGPRs without
    scratch -- a spill adds VMEM operations the counted waits do not know about, and the e4m3 forms did spill before their MFMAs
    were pinned in place."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("ckr", os.path.join(root, "tools", "check_kernel_resources.py"))
    ckr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ckr)
    obj = os.path.join(root, "llm-inference-engine_amd", "csrc", "_obj", "linear.hip.o")
    ks = [k for k in ckr.kernel_metadata(obj) if "gemm8p" in k["name"]]
    assert len(ks) >= 10, [k["name"] for k in ks]
    bad = [k for k in ks if k["vgpr_spill"] or k["sgpr_spill"] or k["scratch"] or k["vgpr"] > 256]
    assert not bad, bad


def test_row_swaps_use_both_result_registers():
    """v_permlane16_swap / v_permlane32_swap (gfx950) exchange rows between TWO registers.  hipcc (ROCm 7.2) mis-lowers the second
    element of the builtin's result pair (both elements read the first register): `max(s[0], s[1])` silently became `s[0]`
    (device_utils.cuh lane_row_swap carries the workaround; tools/micro/lane_xor_check.hip checks the values on the GPU).  Here,
    on the built objects: the signature of the collapse -- the swap's SECOND register overwritten right behind the swap without having
    been read (`v_permlane32_swap v109, v110; v_mov v110, v109`) -- must not occur."""
    import glob
    import os
    import re
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    llvm = "/opt/rocm/lib/llvm/bin/"
    swaps = 0
    for obj in sorted(glob.glob(os.path.join(root, "llm-inference-engine_amd", "csrc", "_obj", "*.hip.o"))):
        with tempfile.TemporaryDirectory() as td:
            fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
            if subprocess.run([llvm + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(td, "x")],
                              capture_output=True).returncode != 0:
                continue   # (a translation unit without device code)
            subprocess.check_call([llvm + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat,
                                   "--output=" + co, "--unbundle"])
            lines = [l.split("//")[0].strip() for l in subprocess.check_output([llvm + "llvm-objdump", "-d", co], text=True).splitlines()]
        lines = [l for l in lines if l and not l.endswith(":")]

        def regs(tok):   # v12 -> {12}; v[8:11] -> {8..11}
            m = re.fullmatch(r"v(\d+)", tok)
            if m:
                return {int(m.group(1))}
            m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
            return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()

        for i, l in enumerate(lines):
            m = re.match(r"v_permlane(16|32)_swap_b32\S*\s+v(\d+),\s*v(\d+)", l)
            if not m:
                continue
            swaps += 1
            second = int(m.group(3))
            verdict = True   # (nothing conclusive within the window -- far uses, control flow -- is not the collapse signature)
            for nxt in lines[i + 1:i + 13]:
                parts = nxt.split(None, 1)
                if len(parts) < 2 or parts[0].startswith(("s_", "ds_write", "global_store", "buffer_store")) and "v" not in parts[1]:
                    continue
                ops = [t.strip() for t in re.split(r",\s*", parts[1].split(" row_")[0].split(" quad_perm")[0])]
                writes_first = not parts[0].startswith(("global_store", "ds_write", "buffer_store", "scratch_store", "v_cmp", "s_"))
                srcs = ops[1:] if writes_first else ops
                if any(second in regs(t.split(" ")[0]) for t in srcs):
                    verdict = True
                    break
                if writes_first and ops and second in regs(ops[0].split(" ")[0]):
                    verdict = False
                    break
            assert verdict, "%s: the second register of `%s` is overwritten or never read: the swap's result pair collapsed" % (os.path.basename(obj), l)
    assert swaps >= 4, swaps   # (flash prefill: 2 per instantiation; decode attention: many)

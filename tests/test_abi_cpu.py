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

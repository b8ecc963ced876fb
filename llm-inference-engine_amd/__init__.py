"""llm-inference-engine_amd -- MI355X-native Llama-2 decoder hot path.

Python side = plumbing only: it loads the in-tree C-ABI library
(lib/libllmie.so, built by build.py with hipcc for gfx950; see include/llmie.h)
through ctypes and hands it device pointers of torch tensors.  There is NO CPU or
PyTorch fallback: if the library is missing or a call fails, an exception is raised.

The directory name carries a hyphen (it mirrors the reference repo's name), so import
it with importlib (tests/conftest.py, bench.py and __graft_entry__.py show how) under
the module name ``llmie_amd``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libllmie.so")

F32, F16 = 0, 1
W_F16, W_INT8, W_INT4, W_FP8, W_F32 = 0, 1, 2, 3, 4
KV_NATIVE, KV_FP8 = 0, 1
DEC_NO_PACKED_COPY, DEC_PACKED_ONLY = 1, 2
ABI_VERSION = 3  # LLMIE_ABI_VERSION of include/llmie.h this binding mirrors

_lib = None


class LlmieError(RuntimeError):
    pass


def build(force=False, jobs=4):
    """Compile every HIP source for gfx950 (hipcc) into lib/libllmie.so."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("llmie_amd_build", os.path.join(_HERE, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(force=force, jobs=jobs)


_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> argtypes (restype int unless listed in _RESTYPES); mirrors include/llmie.h
_SIGS = {
    "llmie_input_embedding": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "llmie_cal_padding_offset": [_vp, _vp, _vp, _i, _i, _vp],
    "llmie_build_causal_mask": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "llmie_rmsnorm": [_vp, _vp, _vp, _f, _i, _i, _i, _vp],
    "llmie_fused_add_bias_residual_rmsnorm": [_vp, _vp, _vp, _vp, _f, _i, _i, _i, _vp],
    "llmie_add_residual": [_vp, _vp, _i, _i, _i, _vp],
    "llmie_linear_workspace_bytes": [_i, _i, _i, _i],
    "llmie_linear": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _sz, _vp],
    "llmie_linear_swiglu": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp],
    "llmie_batched_gemm": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "llmie_qkv_bias_transpose_rope": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    "llmie_rope_decode": [_vp, _i, _i, _i, _i, _i, _vp, _i, _f, _i, _vp],
    "llmie_decoder_mha_workspace_bytes": [_i, _i, _i, _i],
    "llmie_decoder_mha": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _i, _vp],
    "llmie_decoder_mha_rope": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp, _i, _vp, _i,
                               _vp],
    "llmie_concat_kv": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "llmie_repeat_kv": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "llmie_scale_mask_softmax": [_vp, _vp, _vp, _f, _i, _i, _i, _i, _i, _vp],
    "llmie_transpose_remove_padding": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "llmie_silu_and_mul": [_vp, _vp, _i, _i, _i, _vp],
    "llmie_topk": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "llmie_sampling": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _vp],
    "llmie_linear_w8a16": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp],
    "llmie_linear_w4a16": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp],
    "llmie_linear_fp8": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp],
    "llmie_linear_fp8_workspace_bytes": [_i, _i, _i],
    "llmie_linear_fp8_swiglu": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp],
    "llmie_packed_weight_bytes": [_i, _i, _i, _i],
    "llmie_packed_scale_bytes": [_i, _i, _i, _i],
    "llmie_pack_weight": [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "llmie_linear_packed_workspace_bytes": [_i, _i, _i, _i],
    "llmie_linear_packed": [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _f, _vp, _sz, _vp],
    "llmie_x32_bytes": [_i],
    "llmie_x32_convert": [_vp, _vp, _i, _i, _i, _vp],
    "llmie_quantize_w8": [_vp, _vp, _vp, _i, _i, _vp],
    "llmie_quantize_w4": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "llmie_quantize_fp8": [_vp, _vp, _vp, _i, _i, _vp],
    "llmie_decoder_workspace_bytes": [_vp],
    "llmie_decoder_create": [_vp, _vp, _vp, _sz],
    "llmie_decoder_destroy": [_vp],
    "llmie_decoder_forward": [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp],
    "llmie_decoder_forward_ragged": [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "llmie_decoder_forward_paged_ragged": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "llmie_decoder_mha_ragged": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp, _i, _vp, _i, _i, _i, _vp],
    "llmie_decoder_prefill_workspace_bytes": [_vp, _i, _i],
    "llmie_decoder_prefill": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp],
    "llmie_lm_head_sample": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp,
                             _i, _vp],
    "llmie_lm_head_sample_next": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp,
                                  _i, _vp],
    "llmie_advance_step": [_vp, _vp],
    "llmie_decoder_forward_paged": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "llmie_decoder_prefill_paged": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _sz, _vp],
    "llmie_kv_pages_copy": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "llmie_decoder_profile_begin": [_vp, _i],
    "llmie_decoder_profile_end": [_vp, _vp, _vp, _vp],
    "llmie_decoder_status": [_vp, _vp],
    "llmie_decoder_resident_weight_bytes": [_vp],
    "llmie_decoder_repack": [_vp, _vp, _vp],
    "llmie_decoder_debug_stamps": [_vp, _vp],
    "llmie_abi_version": [],
    "llmie_last_error": [],
    "llmie_target_arch": [],
}
_RESTYPES = {
    "llmie_decoder_mha_workspace_bytes": _sz,
    "llmie_decoder_resident_weight_bytes": _sz,
    "llmie_linear_fp8_workspace_bytes": _sz,
    "llmie_linear_workspace_bytes": _sz,
    "llmie_packed_weight_bytes": _sz,
    "llmie_x32_bytes": _sz,
    "llmie_packed_scale_bytes": _sz,
    "llmie_linear_packed_workspace_bytes": _sz,
    "llmie_decoder_workspace_bytes": _sz,
    "llmie_decoder_prefill_workspace_bytes": _sz,
    "llmie_decoder_create": _vp,
    "llmie_decoder_destroy": None,
    "llmie_last_error": C.c_char_p,
    "llmie_target_arch": C.c_char_p,
}

EXPORTS = tuple(sorted(_SIGS))


def lib():
    """The loaded C-ABI library.  Raises (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LlmieError(
                "HIP library %s is missing: run `python llm-inference-engine_amd/build.py` "
                "(or __graft_entry__.build()).  There is no CPU fallback." % LIB_PATH)
        # torch ships its own libamdhip64; load it FIRST so libllmie.so binds to the same HIP runtime
        # (two runtimes in one process = "no ROCm-capable device" on the second one).
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(l, name)  # AttributeError if the ABI lost a symbol
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, C.c_int)
        if l.llmie_abi_version() != ABI_VERSION:
            raise LlmieError("libllmie.so ABI version %d, this binding was written against %d (include/llmie.h)"
                             % (l.llmie_abi_version(), ABI_VERSION))
        _lib = l
    return _lib


def _check(rc, what):
    if rc != 0:
        raise LlmieError("%s failed (%d): %s" % (what, rc, lib().llmie_last_error().decode()))


def _dt(t):
    import torch
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float16:
        return F16
    raise LlmieError("unsupported dtype %s" % t.dtype)


def _p(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensors only"
    return t.data_ptr()


def _st():
    import torch
    return torch.cuda.current_stream().cuda_stream


# ------------------------------------------------------------------ per-kernel wrappers
def input_embedding(ids, table, out):
    _check(lib().llmie_input_embedding(_p(ids), _p(table), _p(out), ids.numel(), table.shape[1], table.shape[0],
                                       _dt(table), _st()), "input_embedding")
    return out


def cal_padding_offset(padding_offset, cum_seqlens, lens):
    _check(lib().llmie_cal_padding_offset(_p(padding_offset), _p(cum_seqlens), _p(lens), lens.numel(),
                                          padding_offset.shape[1], _st()), "cal_padding_offset")


def build_causal_mask(mask, q_lens, k_lens):
    _check(lib().llmie_build_causal_mask(_p(mask), _p(q_lens), _p(k_lens), mask.shape[0], mask.shape[1],
                                         mask.shape[2], _dt(mask), _st()), "build_causal_mask")
    return mask


def rmsnorm(x, resid, gamma, eps):
    _check(lib().llmie_rmsnorm(_p(x), _p(resid), _p(gamma), eps, x.shape[0], x.shape[1], _dt(x), _st()), "rmsnorm")


def fused_add_bias_residual_rmsnorm(resid, out, bias, gamma, eps):
    _check(lib().llmie_fused_add_bias_residual_rmsnorm(_p(resid), _p(out), _p(bias), _p(gamma), eps, out.shape[0],
                                                       out.shape[1], _dt(out), _st()),
           "fused_add_bias_residual_rmsnorm")


def add_residual(resid, out):
    _check(lib().llmie_add_residual(_p(resid), _p(out), out.shape[0], out.shape[1], _dt(out), _st()), "add_residual")


def linear_workspace_bytes(fmt, M, K, N):
    """bytes of caller-owned split-K slab scratch llmie_linear / _swiglu / _w8a16 / _w4a16 use for this shape (0: none)"""
    return lib().llmie_linear_workspace_bytes(fmt, M, K, N)


_scratch = {}


def _auto_ws(fmt, M, K, N, device):
    """workspace="auto": one grow-only torch buffer per device, owned by this module (the C library itself never allocates)"""
    need = linear_workspace_bytes(fmt, M, K, N)
    if need == 0:
        return None
    import torch
    buf = _scratch.get(device)
    if buf is None or buf.numel() < need:
        buf = torch.empty(need, dtype=torch.uint8, device=device)
        _scratch[device] = buf
    return buf


def _ws_args(workspace, fmt, M, K, N, device):
    if isinstance(workspace, str):
        workspace = _auto_ws(fmt, M, K, N, device)
    return _p(workspace), (workspace.numel() * workspace.element_size() if workspace is not None else 0)


def linear(x, w, y, trans_b=True, bias=None, residual=None, workspace="auto"):
    """workspace: a device tensor of linear_workspace_bytes(W_F16, M, K, N) bytes, None (non-split kernels), or the string auto"""
    M, K = x.shape[0], x.numel() // x.shape[0]
    N = y.numel() // y.shape[0]
    wp, wb = _ws_args(workspace, W_F16, M, K, N, x.device)
    _check(lib().llmie_linear(_p(x), _p(w), _p(y), M, K, N, int(trans_b), _p(bias), _p(residual), _dt(x), wp, wb, _st()),
           "linear")
    return y


def linear_swiglu(x, w_gate_up, y, workspace="auto"):
    wp, wb = _ws_args(workspace, W_F16, x.shape[0], x.shape[1], w_gate_up.shape[0], x.device)
    _check(lib().llmie_linear_swiglu(_p(x), _p(w_gate_up), _p(y), x.shape[0], x.shape[1], w_gate_up.shape[0],
                                     _dt(x), wp, wb, _st()), "linear_swiglu")
    return y


def batched_gemm(a, b, c, trans_b):
    bs, nh, m, k = a.shape
    n = c.shape[3]
    _check(lib().llmie_batched_gemm(_p(a), _p(b), _p(c), bs * nh, m, n, k, int(trans_b), _dt(a), _st()),
           "batched_gemm")
    return c


def qkv_bias_transpose_rope(q, k, v, qkv, bias, padding_offset, history_len, rotary_dim, rotary_base):
    bs, nh, S, hs = q.shape
    _check(lib().llmie_qkv_bias_transpose_rope(_p(q), _p(k), _p(v), _p(qkv), _p(bias), _p(padding_offset),
                                               _p(history_len), bs, S, qkv.shape[0], nh, k.shape[1], hs,
                                               rotary_dim, rotary_base, _dt(qkv), _st()), "qkv_bias_transpose_rope")


def rope_decode(qkv, head_num, kv_head_num, step, rotary_dim, rotary_base, step_dev=None):
    _check(lib().llmie_rope_decode(_p(qkv), qkv.shape[0], head_num, kv_head_num, qkv.shape[2], step, _p(step_dev),
                                   rotary_dim, rotary_base, _dt(qkv), _st()), "rope_decode")


def decoder_mha_workspace_bytes(batch, head_num, head_size, max_seq_len):
    return lib().llmie_decoder_mha_workspace_bytes(batch, head_num, head_size, max_seq_len)


def decoder_mha(qkv, qkv_bias, k_cache, v_cache, out, layer, head_num, kv_head_num, step, workspace,
                step_dev=None):
    bs, _, hs = qkv.shape
    max_seq = k_cache.shape[3]
    _check(lib().llmie_decoder_mha(_p(qkv), _p(qkv_bias), _p(k_cache), _p(v_cache), _p(out), layer, bs, head_num,
                                   kv_head_num, hs, max_seq, step, _p(step_dev), _p(workspace),
                                   0 if workspace is None else workspace.numel() * workspace.element_size(),
                                   _dt(qkv), _st()), "decoder_mha")
    return out


def decoder_mha_ragged(qkv, qkv_bias, k_cache, v_cache, out, layer, head_num, kv_head_num, ctx_len, workspace, rope_table,
                       rotary_dim, max_seq_len, block_table=None):
    """ragged batch: ctx_len device int32 [batch]; caches dense [L, batch, kvh, max_seq, hs] or (block_table given) page pools
    [L, num_pages, kvh, 128, hs]"""
    batch, hs = qkv.shape[0], qkv.shape[-1]
    _check(lib().llmie_decoder_mha_ragged(_p(qkv), _p(qkv_bias), _p(k_cache), _p(v_cache), _p(out), layer, batch, head_num,
                                          kv_head_num, hs, max_seq_len, _p(ctx_len), _p(workspace),
                                          workspace.numel() * workspace.element_size(), _p(rope_table), rotary_dim,
                                          _p(block_table), block_table.shape[1] if block_table is not None else 0,
                                          k_cache.shape[1] if block_table is not None else 0, _dt(qkv), _st()),
           "decoder_mha_ragged")
    return out


def decoder_mha_rope(qkv, qkv_bias, k_cache, v_cache, out, layer, head_num, kv_head_num, step, workspace, rope_table,
                     rotary_dim, tickets, step_dev=None):
    bs, _, hs = qkv.shape
    _check(lib().llmie_decoder_mha_rope(_p(qkv), _p(qkv_bias), _p(k_cache), _p(v_cache), _p(out), layer, bs, head_num,
                                        kv_head_num, hs, k_cache.shape[3], step, _p(step_dev), _p(workspace),
                                        workspace.numel() * workspace.element_size(), _p(rope_table), rotary_dim,
                                        _p(tickets), _dt(qkv), _st()), "decoder_mha_rope")
    return out


def concat_kv(src, cache, cur_len, history_len, layer):
    bs, kvh, max_q, hs = src.shape
    _check(lib().llmie_concat_kv(_p(src), _p(cache), _p(cur_len), _p(history_len), layer, bs, kvh, max_q,
                                 cache.shape[3], hs, _dt(src), _st()), "concat_kv")


def repeat_kv(cache, dst, ctx_len, layer):
    bs, nh, max_k, hs = dst.shape
    _check(lib().llmie_repeat_kv(_p(cache), _p(dst), _p(ctx_len), layer, bs, nh, cache.shape[2], max_k,
                                 cache.shape[3], hs, _dt(dst), _st()), "repeat_kv")


def scale_mask_softmax(qk, mask, out, scale):
    bs, nh, ql, kl = qk.shape
    _check(lib().llmie_scale_mask_softmax(_p(qk), _p(mask), _p(out), scale, bs, nh, ql, kl, _dt(qk), _st()),
           "scale_mask_softmax")
    return out


def transpose_remove_padding(src, padding_offset, dst):
    bs, nh, S, hs = src.shape
    _check(lib().llmie_transpose_remove_padding(_p(src), _p(dst), _p(padding_offset), dst.shape[0], bs, S, nh, hs,
                                                _dt(src), _st()), "transpose_remove_padding")
    return dst


def silu_and_mul(x, out):
    _check(lib().llmie_silu_and_mul(_p(x), _p(out), x.shape[0], x.shape[2], _dt(x), _st()), "silu_and_mul")
    return out


def topk(probs, tmp_ids, tmp_vals, ids, vals, blocks_per_row=8):
    rows, vocab = probs.shape
    K = ids.shape[-1]
    _check(lib().llmie_topk(_p(probs), _p(tmp_ids), _p(tmp_vals), _p(ids), _p(vals), rows, vocab, K, blocks_per_row,
                            _dt(probs), _st()), "topk")


def sampling(topk_id, topk_val, seq_len, finished, out_id, step, end_id, vocab, step_dev=None):
    bs, K = topk_id.shape
    _check(lib().llmie_sampling(_p(topk_id), _p(topk_val), _p(seq_len), _p(finished), _p(out_id), bs, K, step,
                                _p(step_dev), end_id, vocab, _dt(topk_val), _st()), "sampling")


def advance_step(step_dev):
    _check(lib().llmie_advance_step(_p(step_dev), _st()), "advance_step")


# ------------------------------------------------------------------ fused decoder engine
class Matrix(C.Structure):
    _fields_ = [("data", _vp), ("scale", _vp), ("bias", _vp)]


class LayerWeights(C.Structure):
    _fields_ = [("attn_norm_gamma", _vp), ("qkv", Matrix), ("o", Matrix), ("ffn_norm_gamma", _vp),
                ("gate_up", Matrix), ("down", Matrix)]


class DecoderConfig(C.Structure):
    _fields_ = [("head_num", _i), ("kv_head_num", _i), ("head_size", _i), ("inter_size", _i), ("num_layers", _i),
                ("vocab_size", _i), ("max_seq_len", _i), ("max_batch", _i), ("rotary_dim", _i),
                ("rotary_base", _f), ("rms_eps", _f), ("dtype", _i), ("wfmt", _i), ("int4_group", _i),
                ("kv_fmt", _i), ("k_scale", _f), ("v_scale", _f), ("flags", _i)]


def _mat(m):
    """m: tensor | (data, scale) | (data, scale, bias) | dict"""
    if isinstance(m, dict):
        return Matrix(_p(m["data"]), _p(m.get("scale")), _p(m.get("bias")))
    if isinstance(m, (tuple, list)):
        m = list(m) + [None] * (3 - len(m))
        return Matrix(_p(m[0]), _p(m[1]), _p(m[2]))
    return Matrix(_p(m), None, None)


class Decoder:
    """Thin owner of an llmie_decoder handle + its workspace (torch only allocates the bytes).

    layers: list of dicts with keys attn_norm, qkv, o, ffn_norm, gate_up, down; matrix entries are a
    tensor (fp16/fp32 [N,K]) or dict(data=, scale=, bias=).
    """

    def __init__(self, cfg, layers):
        import torch
        self.cfg = DecoderConfig(**cfg)
        self._keep = layers  # keep the weight tensors alive
        arr = (LayerWeights * len(layers))()
        for i, lw in enumerate(layers):
            arr[i] = LayerWeights(_p(lw["attn_norm"]), _mat(lw["qkv"]), _mat(lw["o"]), _p(lw["ffn_norm"]),
                                  _mat(lw["gate_up"]), _mat(lw["down"]))
        nbytes = lib().llmie_decoder_workspace_bytes(C.byref(self.cfg))
        if nbytes == 0:
            raise LlmieError("invalid decoder config")
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        self.handle = lib().llmie_decoder_create(C.byref(self.cfg), arr, self.workspace.data_ptr(), nbytes)
        if not self.handle:
            raise LlmieError("decoder_create: " + lib().llmie_last_error().decode())

    def forward(self, hidden_in, hidden_out, k_cache, v_cache, step, step_dev=None):
        _check(lib().llmie_decoder_forward(self.handle, _p(hidden_in), _p(hidden_out), _p(k_cache), _p(v_cache),
                                           hidden_in.shape[0], step, _p(step_dev), _st()), "decoder_forward")
        return hidden_out

    def forward_ragged(self, hidden_in, hidden_out, k_cache, v_cache, ctx_len):
        """ctx_len: device int32 [batch], context length of each sequence including this step's token"""
        _check(lib().llmie_decoder_forward_ragged(self.handle, _p(hidden_in), _p(hidden_out), _p(k_cache), _p(v_cache),
                                                  hidden_in.shape[0], _p(ctx_len), _st()), "decoder_forward_ragged")
        return hidden_out

    def forward_paged_ragged(self, hidden_in, hidden_out, k_pool, v_pool, block_table, ctx_len):
        _check(lib().llmie_decoder_forward_paged_ragged(self.handle, _p(hidden_in), _p(hidden_out), _p(k_pool), _p(v_pool),
                                                        _p(block_table), block_table.shape[1], k_pool.shape[1],
                                                        hidden_in.shape[0], _p(ctx_len), _st()), "decoder_forward_paged_ragged")
        return hidden_out

    def forward_paged(self, hidden_in, hidden_out, k_pool, v_pool, block_table, step, step_dev=None):
        """pools [L, num_pages, kvh, 128, hs]; block_table int32 [batch, max_pages] (device)"""
        _check(lib().llmie_decoder_forward_paged(self.handle, _p(hidden_in), _p(hidden_out), _p(k_pool), _p(v_pool),
                                                 _p(block_table), block_table.shape[1], k_pool.shape[1], hidden_in.shape[0],
                                                 step, _p(step_dev), _st()), "decoder_forward_paged")
        return hidden_out

    def prefill(self, hidden_in, hidden_out, k_cache, v_cache, input_lengths, history_lengths, max_q_len):
        import torch
        T, bs = hidden_in.shape[0], input_lengths.numel()
        need = lib().llmie_decoder_prefill_workspace_bytes(C.byref(self.cfg), T, bs)
        if getattr(self, "_pf_ws", None) is None or self._pf_ws.numel() < need:
            self._pf_ws = torch.empty(need, dtype=torch.uint8, device="cuda")
        _check(lib().llmie_decoder_prefill(self.handle, _p(hidden_in), _p(hidden_out), _p(k_cache), _p(v_cache),
                                           _p(input_lengths), _p(history_lengths), bs, T, max_q_len,
                                           self._pf_ws.data_ptr(), self._pf_ws.numel(), _st()), "decoder_prefill")
        return hidden_out

    def prefill_paged(self, hidden_in, hidden_out, k_pool, v_pool, block_table, input_lengths, history_lengths, max_q_len):
        """prefill on the paged cache of forward_paged (pools [L, num_pages, kvh, 128, hs])"""
        import torch
        T, bs = hidden_in.shape[0], input_lengths.numel()
        need = lib().llmie_decoder_prefill_workspace_bytes(C.byref(self.cfg), T, bs)
        if getattr(self, "_pf_ws", None) is None or self._pf_ws.numel() < need:
            self._pf_ws = torch.empty(need, dtype=torch.uint8, device="cuda")
        _check(lib().llmie_decoder_prefill_paged(self.handle, _p(hidden_in), _p(hidden_out), _p(k_pool), _p(v_pool),
                                                 _p(block_table), block_table.shape[1], k_pool.shape[1], _p(input_lengths),
                                                 _p(history_lengths), bs, T, max_q_len, self._pf_ws.data_ptr(),
                                                 self._pf_ws.numel(), _st()), "decoder_prefill_paged")
        return hidden_out

    def lm_head_sample(self, hidden, final_gamma, lm_head, lm_fmt, logits, tmp_ids, tmp_vals, topk_ids, topk_vals,
                       seq_len, finished, out_ids, step, end_id, blocks_per_row=8, step_dev=None, embed=None, next_hidden=None,
                       advance=False, fused_tail=False):
        """fused_tail (or embed / advance): llmie_lm_head_sample_next -- top-k round 2 + sampling (+ next_hidden[b] = embed[out_ids[b]])
        (+ step_dev += 1) in one launch"""
        m = _mat(lm_head)
        if fused_tail or embed is not None or advance:
            _check(lib().llmie_lm_head_sample_next(self.handle, _p(hidden), _p(final_gamma), C.byref(m), lm_fmt, _p(logits),
                                                   _p(tmp_ids), _p(tmp_vals), _p(topk_ids), _p(topk_vals), topk_ids.shape[-1],
                                                   blocks_per_row, _p(seq_len), _p(finished), _p(out_ids), hidden.shape[0], step,
                                                   _p(step_dev), end_id, _p(embed), _p(next_hidden), 1 if advance else 0, _st()),
                   "lm_head_sample_next")
            return
        _check(lib().llmie_lm_head_sample(self.handle, _p(hidden), _p(final_gamma), C.byref(m), lm_fmt, _p(logits),
                                          _p(tmp_ids), _p(tmp_vals), _p(topk_ids), _p(topk_vals),
                                          topk_ids.shape[-1], blocks_per_row, _p(seq_len), _p(finished), _p(out_ids),
                                          hidden.shape[0], step, _p(step_dev), end_id, _st()), "lm_head_sample")

    OPS = ("attn_norm", "qkv_gemm", "rope", "mha", "o_gemm", "ffn_norm", "gate_up_swiglu", "down_gemm",
           "final_norm", "lm_head", "topk", "sampling", "chain")

    def profile_begin(self, max_events):
        _check(lib().llmie_decoder_profile_begin(self.handle, max_events), "decoder_profile_begin")

    def profile_end(self):
        """-> {op: (total_ms, launches)}; synchronises the current stream"""
        ms = (C.c_double * len(self.OPS))()
        n = (C.c_int * len(self.OPS))()
        _check(lib().llmie_decoder_profile_end(self.handle, _st(), ms, n), "decoder_profile_end")
        return {op: (ms[i], n[i]) for i, op in enumerate(self.OPS)}

    def repack(self, layers):
        """rebuild the tile-packed weight images from `layers` (same structure as at construction)"""
        arr = (LayerWeights * len(layers))()
        for i, lw in enumerate(layers):
            arr[i] = LayerWeights(_p(lw["attn_norm"]), _mat(lw["qkv"]), _mat(lw["o"]), _p(lw["ffn_norm"]), _mat(lw["gate_up"]), _mat(lw["down"]))
        _check(lib().llmie_decoder_repack(self.handle, arr, _st()), "decoder_repack")
        self._keep = layers

    def debug_stamps(self, buf):
        """diagnostic: chain launches write their phase-edge timestamps into buf (uint64 [256, 16] on the device); None disarms"""
        _check(lib().llmie_decoder_debug_stamps(self.handle, _p(buf)), "decoder_debug_stamps")

    def status(self):
        """synchronises the current stream; raises if a grid barrier of a persistent chain launch timed out"""
        _check(lib().llmie_decoder_status(self.handle, _st()), "decoder_status")

    def close(self):
        if self.handle:
            lib().llmie_decoder_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ weight-only quantised linears
def quantize_w8(w, wq, scale):
    _check(lib().llmie_quantize_w8(_p(w), _p(wq), _p(scale), w.shape[0], w.shape[1], _st()), "quantize_w8")


def quantize_w4(w, wq, scale, group):
    _check(lib().llmie_quantize_w4(_p(w), _p(wq), _p(scale), w.shape[0], w.shape[1], group, _st()), "quantize_w4")


def linear_w8a16(x, wq, scale, y, bias=None, residual=None, workspace="auto"):
    wp, wb = _ws_args(workspace, W_INT8, x.shape[0], x.shape[1], wq.shape[0], x.device)
    _check(lib().llmie_linear_w8a16(_p(x), _p(wq), _p(scale), _p(y), x.shape[0], x.shape[1], wq.shape[0], _p(bias),
                                    _p(residual), wp, wb, _st()), "linear_w8a16")
    return y


def linear_w4a16(x, wq, scale, y, group, bias=None, residual=None, workspace="auto"):
    wp, wb = _ws_args(workspace, W_INT4, x.shape[0], x.shape[1], wq.shape[0], x.device)
    _check(lib().llmie_linear_w4a16(_p(x), _p(wq), _p(scale), _p(y), x.shape[0], x.shape[1], wq.shape[0], group,
                                    _p(bias), _p(residual), wp, wb, _st()), "linear_w4a16")
    return y


def quantize_fp8(w, wq, scale):
    _check(lib().llmie_quantize_fp8(_p(w), _p(wq), _p(scale), w.shape[0], w.shape[1], _st()), "quantize_fp8")


KV_PAGE_TOKENS = 128


def kv_pages_copy(dense, pool, block_table, ctx_len, to_pages):
    """dense [L, batch, kvh, max_seq, hs] <-> pool [L, num_pages, kvh, 128, hs] for the first ctx_len[b] tokens of each sequence"""
    L, batch, kvh, max_seq, hs = dense.shape
    _check(lib().llmie_kv_pages_copy(_p(dense), _p(pool), _p(block_table), _p(ctx_len), 1 if to_pages else 0, L, batch, kvh,
                                     max_seq, hs, block_table.shape[1], pool.shape[1], dense.element_size(), _st()),
           "kv_pages_copy")


def linear_fp8_workspace_bytes(M, K, N=0):
    """N = 0: the activation part only (linear_fp8_swiglu, prefill-sized linear_fp8)"""
    return lib().llmie_linear_fp8_workspace_bytes(M, K, N)


def linear_fp8_swiglu(x, wq, wscale, y, workspace):
    _check(lib().llmie_linear_fp8_swiglu(_p(x), _p(wq), _p(wscale), _p(y), x.shape[0], x.shape[1], wq.shape[0], _p(workspace),
                                         workspace.numel() * workspace.element_size(), _st()), "linear_fp8_swiglu")
    return y


def linear_fp8(x, wq, wscale, y, workspace, bias=None, residual=None):
    _check(lib().llmie_linear_fp8(_p(x), _p(wq), _p(wscale), _p(y), x.shape[0], x.shape[1], wq.shape[0], _p(bias),
                                  _p(residual), _p(workspace), workspace.numel() * workspace.element_size(), _st()),
           "linear_fp8")
    return y


# ---- tile-packed weight images + batch-decode linear on them (include/llmie.h section 2) ----
def pack_weight(fmt, w, scale=None, swiglu_pairs=False):
    """row-major weights (fmt's storage, [N, K] logical) -> (packed image uint8 tensor, packed int4 group scales or None)"""
    import torch
    N = w.shape[0]
    K = w.shape[1] * (2 if fmt == W_INT4 else 1)
    nbytes = lib().llmie_packed_weight_bytes(fmt, N, K, int(swiglu_pairs))
    if nbytes == 0:
        raise LlmieError("pack_weight: shape N=%d K=%d not packable in format %d" % (N, K, fmt))
    packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    sbytes = lib().llmie_packed_scale_bytes(fmt, N, K, int(swiglu_pairs))
    pscale = torch.empty(sbytes, dtype=torch.uint8, device=w.device) if sbytes else None
    _check(lib().llmie_pack_weight(fmt, _p(w), _p(scale), _p(packed), _p(pscale), N, K, int(swiglu_pairs), _st()), "pack_weight")
    return packed, pscale


X32_X, X32_Y, X32_RES = 1, 2, 4


def linear_packed(fmt, x, packed, scale, y, N, swiglu=False, residual=None, gamma=None, pre_bias=None, eps=0.0, M=None, K=None,
                  x32_flags=0):
    """y = [swiglu](rmsnorm(x + pre_bias) * gamma . W^T) (+ residual) on a packed image; scale = per-row (int8 / fp8) or the
    packed group scales (int4).  x32_flags: operands in the x32 activation layout (pass M, K explicitly for an x32 x)"""
    import torch
    if M is None:
        M, K = x.shape
    ws_bytes = lib().llmie_linear_packed_workspace_bytes(fmt, M, K, N)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    _check(lib().llmie_linear_packed(fmt, _p(x), _p(packed), _p(scale), _p(y), M, K, N, int(swiglu), int(x32_flags), _p(residual),
                                     _p(gamma), _p(pre_bias), float(eps), _p(ws), ws_bytes, _st()), "linear_packed")
    return y


def x32_convert(src, dst, M, C, to_x32):
    """row-major [M, C] fp16 <-> x32 image (llmie_x32_bytes(C) bytes)"""
    _check(lib().llmie_x32_convert(_p(src), _p(dst), M, C, int(to_x32), _st()), "x32_convert")
    return dst

"""Pins the CPU oracle (oracle/llmie_oracle.c) against every known answer the
reference's own tests hold for the hot path (tests/golden/known_answers.json,
produced by tests/golden/make_golden.py -- each entry cites its reference
file:line).  CPU only."""
import ctypes
import hashlib

import numpy as np
import pytest

import oracle as orc

libc = ctypes.CDLL("libc.so.6")
libc.rand.restype = ctypes.c_int


def _rand(n, mod, add=0):
    out = np.empty(n, np.int64)
    for i in range(n):
        out[i] = libc.rand() % mod + add
    return out


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("key", ["padding_offset_doc_example", "padding_offset_test_comment"])
def test_padding_offset(golden, key):
    g = golden[key]
    off, cum = orc.cal_padding_offset(g["lens"], g["max_q_len"], fill=-7)
    n = sum(g["lens"])
    assert cum.tolist() == g["cum_seqlens"]
    assert off.reshape(-1)[:n].tolist() == g["padding_offset_packed"]
    # entries past the packed prefix are never written (reference kernel leaves them as is)
    assert (off.reshape(-1)[n:] == -7).all()


def test_topk_ramp(golden):
    g = golden["topk_ramp"]
    probs = np.arange(g["rows"] * g["vocab"], dtype=np.float32).reshape(g["rows"], g["vocab"])
    ids, vals = orc.topk(probs, g["K"])
    assert ids.tolist() == g["ids"]
    assert vals.tolist() == g["vals"]


def test_topk_ties_and_negatives():
    p = np.array([[-3.0, -1.0, -1.0, -2.0, -5.0]], np.float32)
    ids, vals = orc.topk(p, 3)
    assert ids.tolist() == [[1, 2, 3]]  # ties -> lower id first; negatives are legal logits
    assert vals.tolist() == [[-1.0, -1.0, -2.0]]


def test_swiglu_ones(golden):
    g = golden["swiglu_ones"]
    x = np.ones((g["batch"], 2, g["inter"]), np.float32)
    y = orc.silu_and_mul(x)
    assert np.abs(y - g["expected"]).max() <= g["tol"]


def test_rmsnorm_ones(golden):
    g = golden["rmsnorm_ones"]
    x = np.ones((g["tokens"], g["hidden"]), np.float32)
    y, r = orc.rmsnorm(x, np.ones(g["hidden"], np.float32), g["eps"])
    assert np.abs(y - g["expected"]).max() <= g["tol"]
    assert (r == x).all()


def test_rmsnorm_fp32_pattern(golden):
    g = golden["rmsnorm_fp32_pattern"]
    T, H = g["tokens"], g["hidden"]
    idx = np.arange(T * H, dtype=np.int64)
    x = ((idx * idx) % 3 + 1).astype(np.float32).reshape(T, H)
    gam = (np.arange(H) % 3 + 1).astype(np.float32)
    y, _ = orc.rmsnorm(x, gam, g["eps"])
    flat = y.reshape(-1)
    for i, e in zip(g["sample_index"], g["sample_expected"]):
        assert abs(flat[i] - e) <= g["tol"]
    assert np.allclose(y.sum(axis=1)[:4], g["row_sums"], rtol=1e-5)


def test_add_residual_pattern(golden):
    g = golden["add_residual_pattern"]
    n = g["tokens"] * g["hidden"]
    a = (np.arange(n) % 2 + 1).astype(np.float32).reshape(g["tokens"], g["hidden"])
    y = orc.add_residual(a, a)
    assert (y == 2 * a).all()


def test_transpose_remove_padding(golden):
    g = golden["transpose_remove_padding"]
    src = np.arange(np.prod(g["shape"]), dtype=np.float32).reshape(g["shape"])
    y = orc.transpose_remove_padding(src, g["padding_offset"], g["num_tokens"])
    assert y.reshape(-1).tolist() == g["expected"]


def test_embedding_rowid(golden):
    g = golden["embedding_rowid"]
    rng = np.random.default_rng(42)
    ids = rng.integers(0, g["vocab"], g["tokens"]).astype(np.int32)
    table = np.repeat(np.arange(g["vocab"], dtype=np.float32)[:, None], g["hidden"], axis=1)
    y = orc.input_embedding(ids, table)
    assert (y == ids[:, None].astype(np.float32)).all()


def test_linear_srand233(golden):
    g = golden["linear_srand233"]
    libc.srand(g["srand"])
    w = _rand(g["N"] * g["K"], 3).reshape(g["N"], g["K"])
    x = _rand(g["M"] * g["K"], 3).reshape(g["M"], g["K"])
    assert _sha(w.astype(np.int8)) == g["w_sha256_int8"]
    assert _sha(x.astype(np.int8)) == g["x_sha256_int8"]
    y = orc.linear(x.astype(np.float32), w.astype(np.float32), trans_b=True)
    yi = np.rint(y).astype(np.int32)
    assert np.abs(y - yi).max() == 0.0          # small ints: exact in fp32
    assert _sha(yi) == g["y_sha256_int32"]
    assert yi.reshape(-1)[:5].tolist() == g["y_first5"]


def test_causal_mask_rand(golden):
    g = golden["causal_mask_rand"]
    m = orc.build_causal_mask(g["q_lens"], g["k_lens"], g["max_q_len"], g["max_k_len"])
    assert set(np.unique(m).tolist()) <= {0.0, 1.0}
    assert _sha(m.astype(np.uint8)) == g["mask_sha256_uint8"]
    assert int(m.sum()) == g["ones"]


def test_fused_norm_ones(golden):
    g = golden["fused_norm_ones"]  # pins a5 (fused add-bias-residual-RMSNorm)
    T, H = g["tokens"], g["hidden"]
    y, r = orc.fused_add_bias_residual_rmsnorm(np.full((T, H), g["residual_fill"], np.float32), np.full((T, H), g["out_fill"], np.float32),
                                               np.full(H, g["bias_fill"], np.float32), np.full(H, g["gamma_fill"], np.float32), g["eps"])
    assert np.abs(y - g["expected"]).max() <= 1e-6 and abs(g["expected"] - 0.8164966) < 1e-7
    assert (r == g["expected_residual"]).all()


def test_softmax_mod8(golden):
    g = golden["softmax_mod8"]  # pins a14 (scale-mask-softmax)
    bs, nh, ql, kl = g["shape"]
    qk = (np.arange(bs * nh * ql * kl) % 8).astype(np.float32).reshape(bs, nh, ql, kl)
    y = orc.scale_mask_softmax(qk, np.ones((bs, ql, kl), np.float32), g["scale"])
    assert np.abs(y - np.array(g["row"], np.float64)).max() <= g["tol"]
    assert abs(sum(g["row"]) - 1.0) < 2e-6


def test_concat_kv_ones(golden):
    g = golden["concat_kv_ones"]
    src = np.full((g["batch"], g["kv_head_num"], g["max_q_len"], g["head_size"]), g["src_fill"], np.float32)
    cache = np.full((1, g["batch"], g["kv_head_num"], g["max_seq_len"], g["head_size"]), -7.0, np.float32)
    orc.concat_kv(src, cache, g["cur_query_length"], g["history_length"], g["layer"])
    lo, hi = g["written_rows"]
    assert (cache[0, :, :, lo:hi + 1] == 1.0).all()
    assert (cache[0, :, :, :lo] == -7.0).all() and (cache[0, :, :, hi + 1:] == -7.0).all()


def test_repeat_kv_ramp(golden):
    g = golden["repeat_kv_ramp"]
    cache = np.arange(np.prod(g["cache_shape"]), dtype=np.float32).reshape(g["cache_shape"])
    y = orc.repeat_kv(cache, g["ctx_len"], g["layer"], g["head_num"], g["max_k_len"])
    assert y.reshape(-1).tolist() == g["expected"]

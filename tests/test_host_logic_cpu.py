"""Host-side planning logic of the C-ABI library, exercised WITHOUT a GPU (pure arithmetic behind the size queries a caller
makes before it allocates): split-K slab workspaces, packed-weight images, decoder workspaces."""
import ctypes as C

import pytest


@pytest.fixture(scope="module")
def lib(llmie):
    return llmie.lib()


def test_linear_workspace_query_follows_the_split_k_plans(llmie, lib):
    F16, I8, I4, FP8 = llmie.W_F16, llmie.W_INT8, llmie.W_INT4, llmie.W_FP8
    q = lib.llmie_linear_workspace_bytes
    # fp16: decode and short-prefill batches, and (round 3) prefill-sized row counts whose 256-row grid does not fill the chip --
    # those may run as 128-row split-K passes (N = 4096: up to 768 rows; N = 12288 fills from 257 rows); where the grid fills, the
    # tiled kernels need no slabs
    assert q(F16, 1, 4096, 4096) > 0 and q(F16, 192, 4096, 4096) > 0
    assert q(F16, 193, 4096, 4096) == q(F16, 128, 4096, 4096) and q(F16, 768, 4096, 4096) == q(F16, 128, 4096, 4096)
    assert q(F16, 1024, 4096, 4096) == 0 and q(F16, 2048, 4096, 12288) == 0 and q(F16, 512, 4096, 12288) == 0
    # shapes without a split-K form: K too short / not a multiple of the sub-block
    assert q(F16, 32, 256, 4096) == 0 and q(F16, 32, 4096 + 64, 4096) == 0 and q(I8, 32, 4096 + 128, 4096) == 0
    # at least one slab of M x N floats, at most 16 slabs; more rows never need less
    for fmt in (F16, I8, I4, FP8):
        prev = 0
        for m in (1, 8, 16, 32, 33, 64, 65, 128):
            if fmt == I4 and m > 64:
                continue
            b = q(fmt, m, 4096, 12288)
            assert m * 12288 * 4 <= b <= 16 * m * 12288 * 4, (fmt, m, b)
            assert b >= prev or m in (33, 65), (fmt, m)   # (a different kernel form starts at 33 and 65 rows: fewer, larger slices)
            prev = b
    # 128-row kernel: the tile width follows the chip fill -- 192-row tiles for the 7B gate/up (115 x 2 slices) and QKV (64 x 4)
    # projections, 128-row tiles for the 4096-wide ones (32 x 8)
    assert q(F16, 128, 4096, 22016) == 2 * 128 * 22016 * 4
    assert q(F16, 128, 4096, 12288) == 4 * 128 * 12288 * 4
    assert q(F16, 128, 4096, 4096) == 8 * 128 * 4096 * 4
    assert q(F16, 64, 4096, 22016) == 2 * 64 * 22016 * 4
    # more than one pass of 128 rows reuses the same slabs (below the prefill-sized forms: 192 rows)
    assert q(I8, 191, 4096, 4096) == q(I8, 128, 4096, 4096)
    # int8 / int4 from 192 rows (round 3): room for the fp16 image of W in front of the slabs
    img = 4096 * 4096 * 2
    assert q(I8, 500, 4096, 4096) == img + q(I8, 128, 4096, 4096) and q(I4, 192, 4096, 4096) == img + q(I4, 64, 4096, 4096)
    # garbage in, zero out
    assert q(F16, 0, 4096, 4096) == 0 and q(F16, 8, -1, 4096) == 0 and q(99, 8, 4096, 4096) == 0


def test_fp8_workspace_query(lib):
    q = lib.llmie_linear_fp8_workspace_bytes
    act = q(64, 4096, 0)
    assert act >= 64 * 4096 + 64 * 4 and act % 256 == 0
    assert q(64, 4096, 4096) >= act + 64 * 4096 * 4          # + split-K slabs
    assert q(8, 4096, 4096) == q(8, 4096, 0)                 # GEMV sizes need no slabs
    assert q(0, 4096, 4096) == 0


def test_packed_image_sizes(llmie, lib):
    wb, sb = lib.llmie_packed_weight_bytes, lib.llmie_packed_scale_bytes
    # tile = 16 rows, padded up; bytes per element by format
    assert wb(llmie.W_F16, 4096, 4096, 0) == 4096 * 4096 * 2
    assert wb(llmie.W_INT8, 4096, 4096, 0) == 4096 * 4096 and wb(llmie.W_FP8, 4096, 4096, 0) == 4096 * 4096
    assert wb(llmie.W_INT4, 4096, 4096, 0) == 4096 * 4096 // 2
    assert wb(llmie.W_F16, 40, 512, 0) == 48 * 512 * 2                     # 40 rows -> 3 tiles
    assert wb(llmie.W_INT8, 88, 1024, 1) == 2 * 48 * 1024                  # SwiGLU pairs: 44 -> 48 gate rows + 48 up rows
    assert wb(llmie.W_INT8, 64, 1000, 0) == 0                              # K not a multiple of the block
    # only int4 has a scale image: [tiles][K / 128][16] fp16 + 256 bytes of padding for the 256-byte DMA
    assert sb(llmie.W_INT8, 4096, 4096, 0) == 0
    assert sb(llmie.W_INT4, 4096, 4096, 0) == 256 * 32 * 32 + 256
    assert lib.llmie_x32_bytes(4096) == 4096 * 64 and lib.llmie_x32_bytes(100) == 0


def _cfg(llmie, **kw):
    base = dict(head_num=32, kv_head_num=32, head_size=128, inter_size=11008, num_layers=2, vocab_size=32000, max_seq_len=512,
                max_batch=1, rotary_dim=128, rotary_base=10000.0, rms_eps=1e-5, dtype=llmie.F16, wfmt=llmie.W_F16, int4_group=128)
    base.update(kw)
    return llmie.DecoderConfig(**base)


def test_decoder_workspace_grows_with_what_it_has_to_hold(llmie, lib):
    q = lambda **kw: lib.llmie_decoder_workspace_bytes(C.byref(_cfg(llmie, **kw)))
    layer_bytes = (3 * 4096 * 4096 + 4096 * 4096 + 3 * 4096 * 11008) * 2
    b1, b5, b6, b32, b128 = q(max_batch=1), q(max_batch=5), q(max_batch=6), q(max_batch=32), q(max_batch=128)
    assert 0 < b1 <= b5 < b6 <= b32 < b128
    # fp16 engines above the GEMV range (5 rows since the crossover was re-measured at the end of round 3) carry the packed second
    # copy of the layer weights
    assert b6 - b5 >= 2 * layer_bytes and b5 < layer_bytes
    # int8: from batch 3; int4: from batch 3 as well (its batch 2 went back to the GEMV), with its group-scale images
    i8 = lambda b: q(max_batch=b, wfmt=llmie.W_INT8)
    assert i8(3) - i8(2) >= layer_bytes
    i4 = lambda b: q(max_batch=b, wfmt=llmie.W_INT4)
    assert i4(3) - i4(2) >= layer_bytes // 2
    # the split-K slab area covers the LM head too (a 32000-row vocabulary beside small layers: its slabs dominate)
    small = dict(head_num=8, kv_head_num=8, head_size=64, inter_size=768, max_batch=128)
    lm = lib.llmie_linear_workspace_bytes(llmie.W_F16, 128, 512, 32000)
    assert lm >= 128 * 32000 * 4
    assert q(vocab_size=32000, **small) - q(vocab_size=1000, **small) >= lm - lib.llmie_linear_workspace_bytes(llmie.W_F16, 128, 512, 2 * 768)
    # invalid configurations answer 0
    assert q(head_num=0) == 0 and q(max_batch=0) == 0 and q(head_num=32, kv_head_num=5) == 0

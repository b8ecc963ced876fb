"""Tokenizer: llm-inference-engine_amd/api/tokenizer.hpp (hash-map merges) against the restatement of the
reference's trie tokenizer (oracle/tokenizer_oracle.py, src/models/tokenizer.h) on a synthetic vocabulary file in the
reference's binary format.  Host-only: g++, no GPU.  Parity unpinned in the reference (no test, no vocabulary)."""
import os
import random
import struct
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from tokenizer_oracle import BLANK, RefTokenizer  # noqa: E402


def _write_vocab(path, entries, version=1):
    with open(path, "wb") as f:
        f.write(struct.pack("<i", version))
        if version >= 1:
            f.write(struct.pack("<i", 2))
            for k, v in ((b"tokenizer_use_score", b"1"), (b"model_type", b"llama")):
                f.write(struct.pack("<i", len(k)) + k + struct.pack("<i", len(v)) + v)
        f.write(struct.pack("<i", len(entries)))
        for b, tid, score in entries:
            f.write(struct.pack("<i", len(b)))
            for c in b:
                f.write(struct.pack("<i", c))
            f.write(struct.pack("<if", tid, score))


def _vocab(rng):
    entries, tid = [], 0
    for special in (b"<unk>", b"<s>", b"</s>"):
        entries.append((special, tid, 0.0)); tid += 1
    for c in range(256):
        entries.append((b"<0x%02X>" % c, tid, 0.0)); tid += 1
    singles = [bytes([c]) for c in b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789.,!?'-:;()"] + [BLANK]
    for i, b in enumerate(singles):
        entries.append((b, tid, -1000.0 - i)); tid += 1
    words = ["he", "ll", "lo", "hel", "hell", "hello", "wor", "world", "or", "ld", "th", "the", "in", "ing", "an", "and", "er",
             "re", "on", "at", "en", "is", "it", "to", "of", "ed", "ou", "you", "are", "con", "sci", "ous", "conscious", "talk",
             "me", "can", "Can", "Hey"]
    seen = set(singles)
    for w in words:
        for b in (w.encode(), BLANK + w.encode()):
            if b not in seen:
                seen.add(b)
                entries.append((b, tid, -float(rng.randint(1, 400)))); tid += 1
    for b, s in ((BLANK + BLANK, -5.0), (b"<n>", -2.0), (b"<|tab|>", -2.0), (b"\xc3\xa9", -50.0), (b"!!", -7.0), (b"??", -7.0)):
        entries.append((b, tid, s)); tid += 1
    return entries


@pytest.fixture(scope="module")
def cli(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("tok") / "tokenizer_cli")
    src = os.path.join(ROOT, "llm-inference-engine_amd", "cpp_tests", "tokenizer_cli.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", src, "-o", out])
    return out


def _run(cli, vocab, lines):
    p = subprocess.run([cli, vocab], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True)
    return p.stdout.splitlines()


@pytest.mark.parametrize("version", [0, 1])
def test_encode_decode_match_reference_restatement(cli, tmp_path, version):
    rng = random.Random(7)
    entries = _vocab(rng)
    vocab = str(tmp_path / "vocab.bin")
    _write_vocab(vocab, entries, version)
    ref = RefTokenizer()
    ref.load(vocab)
    texts = [b"hello world", b"Hey, are you conscious? Can you talk to me?", b"  leading spaces", b"a  b   c", b"trailing ",
             b"", b" ", b"caf\xc3\xa9 \xe4\xb8\xad\xe6\x96\x87", b"<FLM_FIX_TOKEN_123>hello", b"say <FLM_FIX_TOKEN_7> and <FLM",
             b"the thing is in the world!!??", b"\x00\x01\xff", b"hellohellohello", b"tab\there\nnewline"]
    alphabet = b"abcdehlortwy .,!?HC\xc3\xa9" + b"  "
    for _ in range(200):
        texts.append(bytes(rng.choice(alphabet) for _ in range(rng.randint(1, 40))))
    got = _run(cli, vocab, ["E " + t.hex() for t in texts])
    assert len(got) == len(texts)
    all_ids = []
    for t, line in zip(texts, got):
        ids = [int(x) for x in line.split()]
        assert ids == ref.encode(t), t
        all_ids.append(ids)
    # every byte is covered by a token or the byte fallback: decoding restores the text with its spaces normalised
    dec = _run(cli, vocab, ["D " + " ".join(map(str, ids)) for ids in all_ids])
    for t, ids, line in zip(texts, all_ids, dec):
        assert bytes.fromhex(line) == ref.decode(ids), t
        if not t.startswith(b"<FLM") and b"<FLM_FIX" not in t:
            norm = b" ".join(w for w in t.split(b" ") if w or False)
            assert bytes.fromhex(line).replace(b" ", b"") == t.replace(b" ", b"")
    # special tokens of Decode
    ids = {b: tid for b, tid, _ in entries}
    special = [ids[b"<n>"], ids[b"<|tab|>"], ids[b"<0x41>"], ids[BLANK + b"the"]]
    assert bytes.fromhex(_run(cli, vocab, ["D " + " ".join(map(str, special))])[0]) == b"\n\tA the" == ref.decode(special)


def test_known_small_case_by_hand(cli, tmp_path):
    """hand-checkable: vocabulary {▁, h, e, l, o, he(-1), ll(-2), hell(-3), ▁hell(-0.5), lo(-4)}: '▁hello' merges he, ll,
    then hell (he+ll), then ▁hell, leaving 'o' -> [▁hell, o]"""
    entries = [(BLANK, 0, -100.0), (b"h", 1, -100.0), (b"e", 2, -100.0), (b"l", 3, -100.0), (b"o", 4, -100.0),
               (b"he", 5, -1.0), (b"ll", 6, -2.0), (b"hell", 7, -3.0), (BLANK + b"hell", 8, -0.5), (b"lo", 9, -4.0)]
    vocab = str(tmp_path / "v.bin")
    _write_vocab(vocab, entries)
    assert _run(cli, vocab, ["E " + b"hello".hex()]) == ["8 4 "]
    ref = RefTokenizer()
    ref.load(vocab)
    assert ref.encode(b"hello") == [8, 4]

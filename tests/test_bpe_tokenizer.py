"""The open_clip flavour's BPE tokenizer (easynlp_amd/appzoo/clip/bpe_tokenizer.py) against the reference's SimpleTokenizer /
openclip_tokenize (modelzoo/models/clip/openclip_tokenizer.py, appzoo/clip/data.py:137-161): a fixture produced by the
reference (tools/make_golden.py: run_bpe_case) and, when the checkout is present, a live fuzz comparison."""
import gzip
import os
import random

import numpy as np
import pytest
import torch

from easynlp_amd.appzoo.clip.bpe_tokenizer import SimpleTokenizer, byte_symbols, openclip_tokenize
from oracle import ref_harness as R

GOLD = os.path.join(os.path.dirname(__file__), "golden", "openclip_bpe_corpus.npz")


def _tok(tmp_path):
    g = np.load(GOLD)
    path = os.path.join(str(tmp_path), "vocab.txt")
    with gzip.open(path, "wb") as f:
        f.write(g["merges"].tobytes())
    return g, path, SimpleTokenizer(path)


def test_byte_table_is_reversible_and_printable():
    s = byte_symbols()
    assert len(set(s)) == 256 and all(not c.isspace() for c in s)
    assert s[ord("a")] == "a" and s[0] == chr(256) and s[ord(" ")] == chr(256 + 32)


def test_bpe_ids_match_the_reference_fixture(tmp_path):
    g, path, tok = _tok(tmp_path)
    vocab_size, sot, eot, n_merges = [int(x) for x in g["meta"]]
    assert tok.vocab_size == vocab_size >= 512 + n_merges + 2     # (+1: the fixture's trailing empty line is a merge entry too)
    assert tok.encoder["<start_of_text>"] == sot and tok.encoder["<end_of_text>"] == eot == vocab_size - 1
    corpus = g["corpus"].tobytes().decode("utf-8").split("\x1e")
    for text, want in zip(corpus, g["corpus_ids"]):
        want = [int(x) for x in str(want).split(",")] if str(want) else []
        assert tok.encode(text) == want, repr(text)
    for L, key in ((77, "tokens77"), (16, "tokens16")):
        got = openclip_tokenize(corpus, context_length=L, _tokenizer=tok)
        assert got.dtype == torch.int64 and np.array_equal(got.numpy(), g[key]), key
    # a caption longer than the context loses its EOT (data.py:157-158): the arg-max pooling then lands elsewhere
    long_row = corpus.index("word " * 60)
    assert eot not in g["tokens16"][long_row] and int(g["tokens77"][0].max()) == eot
    assert tok.decode(tok.encode("a photo of a cat")).strip() == "a photo of a cat"


@pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")
def test_bpe_fuzz_against_the_live_reference(tmp_path):
    R.install_shims()
    from easynlp.modelzoo.models.clip.openclip_tokenizer import SimpleTokenizer as RefTok
    g, path, mine = _tok(tmp_path)
    ref = RefTok(bpe_path=path)
    assert ref.encoder == mine.encoder
    rnd = random.Random(3)
    words = ["photo", "photograph", "cat", "dogs", "the", "red", "running", "it's", "they've", "I'M", "café", "中文", "猫图", "2023",
             "100%", "&amp;", "&lt;b&gt;", "<start_of_text>", "<END_OF_TEXT>", "\U0001F600", "a_b-c", "...", "½", "x" * 30, "\t", "\n", "  "]
    for _ in range(3000):
        t = " ".join(rnd.choice(words) for _ in range(rnd.randint(0, 12)))
        if rnd.random() < 0.3:
            t = "".join(rnd.choice("abcdefghij klmno'.,!?-_#中文é\t") for _ in range(rnd.randint(0, 30)))
        assert ref.encode(t) == mine.encode(t), repr(t)

"""Input side of the Wukong application: the drop-in FullTokenizer / WukongCLIPDataset / WukongCLIPPredictor against the
reference's own (appzoo/wukong_clip/{bert_tokenizer,data,predictor}.py).  Fixture tests/golden/wukong_dataset_b5.npz
(tools/make_golden.py: run_wukong_dataset_case) = what the REFERENCE dataset and tokenizer produced; a live fuzz comparison
runs when the checkout is present."""
import base64
import hashlib
import io
import os
import random

import numpy as np
import pytest
import torch

from easynlp_amd import lib as L
from easynlp_amd.appzoo.wukong_clip import FullTokenizer, WukongCLIPDataset
from oracle import ref_harness as R

PIL = pytest.importorskip("PIL.Image")
GOLD = os.path.join(os.path.dirname(__file__), "golden", "wukong_dataset_b5.npz")
SCHEMA = "text:str:1,image:str:1"


def _materialise(tmp_path):
    g = np.load(GOLD)
    d = str(tmp_path)
    with open(os.path.join(d, "vocab.txt"), "wb") as f:
        f.write(g["vocab"].tobytes() + b"\n")
    with open(os.path.join(d, "data.tsv"), "wb") as f:
        f.write(g["tsv"].tobytes())
    return g, d


def test_tokenizer_matches_the_reference_on_the_edge_case_corpus(tmp_path):
    g, d = _materialise(tmp_path)
    tok = FullTokenizer(os.path.join(d, "vocab.txt"))
    assert tok.vocab["[CLS]"] == 101 and tok.vocab["[SEP]"] == 102
    corpus = g["corpus"].tobytes().decode("utf-8").split("\x1e")
    assert len(corpus) == len(g["corpus_ids"]) >= 25
    for text, want in zip(corpus, g["corpus_ids"]):
        want = [int(x) for x in str(want).split(",")] if str(want) else []
        assert tok.convert_tokens_to_ids(tok.tokenize(text)) == want, repr(text)
    # what distinguishes it from transformers' BertTokenizer: no special tokens inside text, 200 characters per word
    assert "[CLS]" not in tok.tokenize("[CLS] a") and tok.tokenize("[CLS] a")[-1] == "a"
    assert tok.tokenize("y" * 201) == ["[UNK]"] and tok.tokenize("x" * 150) != ["[UNK]"]
    ids = tok.tokenize_batch(["a photo of a cat", "", "zzz " * 40])
    assert tuple(ids.shape) == (3, 32) and ids.dtype == torch.int64
    assert ids[1].tolist() == [101, 102] + [0] * 30 and int(ids[2][31]) == 102 and int((ids[2] == 102).sum()) == 1


@pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")
def test_tokenizer_fuzz_against_the_live_reference(tmp_path):
    R.install_shims()
    from easynlp.appzoo.wukong_clip.bert_tokenizer import FullTokenizer as RefTok
    g, d = _materialise(tmp_path)
    vp = os.path.join(d, "vocab.txt")
    ref, mine = RefTok(vocab_file=vp), FullTokenizer(vp)
    assert ref.vocab == mine.vocab
    rnd = random.Random(11)
    alphabet = list("abcdefghijklmnopqrstuvwxyzABCXYZ  \t\n,.;:!?'\"-_()[]{}<>@#$%^&*+=~`|\\/0123456789") + \
        list("中文猫狗图的了，。！？「」·—…éÀüñçøßÆ") + ["​", "�", "\x00", "\x07", " ", "　", "́", " ",
                                              "\U0001F600", "\U00020000", "ａ", "①", "々", "〇", "ｶ", "ﾞ", "ก", "ั"]
    for it in range(4000):
        t = "".join(rnd.choice(alphabet) for _ in range(rnd.randint(0, 40)))
        if it % 50 == 0:
            t += "x" * rnd.choice([99, 100, 101, 199, 200, 201, 250])
        assert ref.tokenize(t) == mine.tokenize(t), repr(t)


def test_dataset_tokens_equal_the_reference_batch(tmp_path):
    g, d = _materialise(tmp_path)
    ds = WukongCLIPDataset(d, os.path.join(d, "data.tsv"), 32, input_schema=SCHEMA, first_sequence="text", second_sequence="image")
    assert len(ds) == 5
    batch = ds.batch_fn([ds[i] for i in range(len(ds))])
    assert sorted(batch) == ["image_size", "images", "input_ids"] and batch["image_size"] == 224
    assert np.array_equal(batch["input_ids"].numpy(), g["input_ids"])
    rows = g["tsv"].tobytes().decode("utf-8").split("\n")[:-1]
    for row, img in zip(rows, batch["images"]):
        ref = PIL.open(io.BytesIO(base64.urlsafe_b64decode(row.split("\t")[1])))
        assert np.array_equal(img, np.asarray(ref))
    # always 32 ids, whatever max_seq_length says (data.py:181,218)
    ds64 = WukongCLIPDataset(d, os.path.join(d, "data.tsv"), 64, input_schema=SCHEMA, first_sequence="text", second_sequence="image")
    assert tuple(ds64[0]["text"]["input_ids"].shape) == (1, 32)
    # greyscale: the reference's pipeline has no convert('RGB') and fails in _normalize -- an error here as well
    buf = io.BytesIO()
    PIL.fromarray(np.zeros((8, 8), np.uint8), "L").save(buf, format="PNG")
    with open(os.path.join(d, "grey.tsv"), "w") as f:
        f.write("a\t" + base64.urlsafe_b64encode(buf.getvalue()).decode() + "\n")
    dg = WukongCLIPDataset(d, os.path.join(d, "grey.tsv"), 32, input_schema=SCHEMA, first_sequence="text", second_sequence="image")
    with pytest.raises(RuntimeError):
        dg[0]


@pytest.mark.gpu
def test_gpu_pixel_values_equal_the_reference_wukong_dataset(tmp_path):
    g, d = _materialise(tmp_path)
    ds = WukongCLIPDataset(d, os.path.join(d, "data.tsv"), 32, input_schema=SCHEMA, first_sequence="text", second_sequence="image")
    batch = ds.batch_fn([ds[i] for i in range(len(ds))])
    px = L.preprocess_images(batch["images"], size=224, crop=224).cpu().numpy()
    for i in range(len(ds)):
        assert hashlib.sha256(np.ascontiguousarray(px[i]).tobytes()).hexdigest() == str(g["pixel_sha256"][i]), i

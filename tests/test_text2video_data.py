"""Input side of Text2VideoRetrieval: the drop-in dataset against the reference's own Text2VideoRetrievalDataset
(appzoo/text2video_retrieval/data.py:163-279).  Fixture tests/golden/t2v_dataset_b3.npz (tools/make_golden.py:
run_t2v_dataset_case) = three clips as frame directories, and what the REFERENCE produced: BPE token tensor, video masks,
per-frame SHA-256 of pixel_values keyed by file name (frame order is os.listdir's)."""
import gzip
import json
import os

import numpy as np
import pytest
import torch

from easynlp_amd import lib as L
from easynlp_amd.appzoo.text2video_retrieval import Text2VideoRetrievalDataset
from easynlp_amd.appzoo.text2video_retrieval.data import load_clip_frames, video_mask

PIL = pytest.importorskip("PIL.Image")
GOLD = os.path.join(os.path.dirname(__file__), "golden")
SCHEMA = "text:str:1,image:str:1"


def _materialise(tmp_path):
    g = np.load(os.path.join(GOLD, "t2v_dataset_b3.npz"))
    bpe = np.load(os.path.join(GOLD, "openclip_bpe_corpus.npz"))
    d = str(tmp_path)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump({"model_type": "open_clip"}, f)
    with gzip.open(os.path.join(d, "vocab.txt"), "wb") as f:
        f.write(bpe["merges"].tobytes())
    for k in g.files:
        if k.startswith("png/"):
            path = os.path.join(d, k[len("png/"):])
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "wb") as f:
                f.write(g[k].tobytes())
    with open(os.path.join(d, "data.tsv"), "w") as f:
        for i, cap in enumerate(g["captions"]):
            f.write(str(cap) + "\t" + os.path.join(d, "clip%d" % i) + "\n")
    return g, d


def _dataset(d):
    return Text2VideoRetrievalDataset(d, os.path.join(d, "data.tsv"), 77, input_schema=SCHEMA, first_sequence="text",
                                      second_sequence="image")


def test_tokens_masks_and_frames_equal_the_reference(tmp_path):
    g, d = _materialise(tmp_path)
    ds = _dataset(d)
    batch = ds.batch_fn([ds[i] for i in range(len(ds))])
    assert np.array_equal(batch["input_ids"].numpy(), g["input_ids"]) and tuple(batch["input_ids"].shape) == (3, 77)
    assert np.array_equal(batch["video_masks"].numpy(), g["video_masks"]) and batch["video_masks"].dtype == torch.int64
    assert batch["video_masks"].sum(1).tolist() == [12, 5, 1] and batch["label_ids"] == [] and batch["image_size"] == 224
    assert [len(c) for c in batch["images"]] == [12, 12, 12]
    for ci, clip in enumerate(batch["images"]):
        names = os.listdir(os.path.join(d, "clip%d" % ci))
        for fi, name in enumerate(names):                         # real frames: the decoded pixels, in os.listdir order
            ref = np.asarray(PIL.open(os.path.join(d, "clip%d" % ci, name)))
            assert np.array_equal(clip[fi], ref)
        for fi in range(len(names), 12):                          # padding: black 224 x 224 RGB (data.py:236-238)
            assert clip[fi].shape == (224, 224, 3) and not clip[fi].any()


def test_what_the_reference_does_not_define_is_an_error(tmp_path):
    g, d = _materialise(tmp_path)
    many = os.path.join(d, "many")
    os.makedirs(many)
    for i in range(13):
        PIL.fromarray(np.zeros((8, 8, 3), np.uint8)).save(os.path.join(many, "%02d.png" % i))
    with pytest.raises(L.EzclipError):
        load_clip_frames(many)                                    # 13 frames: mask has 12 slots, the reference breaks downstream
    assert video_mask(0).tolist() == [[0] * 12] and video_mask(12).tolist() == [[1] * 12]
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump({"model_type": "chinese_clip"}, f)
    with pytest.raises(L.EzclipError):
        _dataset(d)

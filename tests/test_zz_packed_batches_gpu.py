"""pack_batches=True of the drop-in datasets on the GPU: one packed uint8 tensor per batch (built in the DataLoader workers)
gives bit-identical pixel_values / outputs to the per-image list path (which tests/test_dataset.py verified on hardware).
Ordered last: written after the round's GPU minutes were spent."""
import json
import os

import numpy as np
import pytest
import torch

from easynlp_amd import lib as L
from easynlp_amd.appzoo.clip import CLIPApp
from easynlp_amd.appzoo.clip.data import CLIPDataset
from oracle import clip_oracle as O
from oracle import ref_harness as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "dataset_tsv_b7.npz")
SCHEMA = dict(input_schema="text:str:1,image:str:1", first_sequence="text", second_sequence="image")


def test_packed_and_list_batches_are_bit_identical(tmp_path):
    g = np.load(GOLD)
    d = str(tmp_path)
    vocab = g["vocab"].tobytes().decode("utf-8").split("\n")
    cfg = dict(O.CONFIGS["tiny"], vocab_size=len(vocab))
    R.write_checkpoint_dir(d, cfg, O.make_state_dict(cfg, 5))
    with open(os.path.join(d, "vocab.txt"), "wb") as f:
        f.write(g["vocab"].tobytes() + b"\n")
    tsv = os.path.join(d, "data.tsv")
    with open(tsv, "wb") as f:
        f.write(g["tsv"].tobytes())
    res = int(cfg["image_resolution"])
    plain = CLIPDataset(d, tsv, 20, image_size=res, **SCHEMA)
    packed = CLIPDataset(d, tsv, 20, image_size=res, pack_batches=True, **SCHEMA)
    b_list = plain.batch_fn([plain[i] for i in range(7)])
    # workers are spawned (the parent already holds a HIP context: no fork after that), and a stuck loader fails instead of hanging
    b_pack = next(iter(torch.utils.data.DataLoader(packed, batch_size=7, shuffle=False, collate_fn=packed.batch_fn,
                                                   num_workers=2, pin_memory=True, multiprocessing_context="spawn",
                                                   timeout=180)))
    assert L.is_packed_images(b_pack["images"])
    px_list = L.preprocess_images(b_list["images"], size=224, crop=224)
    px_pack = L.preprocess_images(b_pack["images"], size=224, crop=224)
    px_dev = L.preprocess_images({k: v.cuda() for k, v in b_pack["images"].items()}, size=224, crop=224)   # already on the device
    assert torch.equal(px_list, px_pack) and torch.equal(px_list, px_dev)
    app = CLIPApp(d, user_defined_parameters={"clip_compute_dtype": "fp32"}).cuda().eval()
    with torch.no_grad():
        a = app(b_list)
        b = app(b_pack)
    assert torch.equal(a["logits_per_text"], b["logits_per_text"]) and torch.equal(a["image_embeds"], b["image_embeds"])

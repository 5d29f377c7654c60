"""WukongCLIP with its drop-in dataset / predictor on the GPU (wukong_clip/{data,predictor}.py of the reference).

Ordered last on purpose (see test_zz_text2video_gpu.py): written after the round's GPU minutes were spent.  The pieces are
covered elsewhere -- tokens and decode on the CPU (test_wukong_data.py), pixel hashes (same file, GPU), towers
(test_wukong_gpu.py); this file only runs them through the application objects."""
import base64
import io
import json
import os

import numpy as np
import pytest
import torch

from easynlp_amd import lib as L
from easynlp_amd.appzoo.wukong_clip import WukongCLIP, WukongCLIPDataset, WukongCLIPPredictor
from oracle import wukong_oracle as WK

pytestmark = pytest.mark.gpu
PIL = pytest.importorskip("PIL.Image")
GOLD = os.path.join(os.path.dirname(__file__), "golden", "wukong_dataset_b5.npz")


def _checkpoint(tmp_path):
    g = np.load(GOLD)
    d = str(tmp_path)
    vocab = g["vocab"].tobytes().decode("utf-8").split("\n")
    cfg = json.loads(json.dumps(WK.WUKONG_CONFIGS["wk_small"]))
    cfg["model"]["text"]["vocab_size"] = len(vocab)
    sd = WK.make_state_dict(cfg, 21, small_embeddings=False)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg, f)
    torch.save(sd, os.path.join(d, "pytorch_model.bin"))
    with open(os.path.join(d, "vocab.txt"), "wb") as f:
        f.write(g["vocab"].tobytes() + b"\n")
    with open(os.path.join(d, "data.tsv"), "wb") as f:
        f.write(g["tsv"].tobytes())
    return g, d, cfg, sd


def test_wukong_forward_takes_dataset_batches(tmp_path):
    g, d, cfg, sd = _checkpoint(tmp_path)
    res = cfg["model"]["visual"]["input_resolution"]
    ds = WukongCLIPDataset(d, os.path.join(d, "data.tsv"), 32, input_schema="text:str:1,image:str:1", first_sequence="text",
                           second_sequence="image", image_size=res)
    app = WukongCLIP(d, user_defined_parameters={"clip_compute_dtype": "fp32"}).cuda().eval()
    dl = torch.utils.data.DataLoader(ds, batch_size=5, shuffle=False, collate_fn=ds.batch_fn)
    batch = next(iter(dl))
    assert np.array_equal(batch["input_ids"].numpy(), g["input_ids"])
    images = list(batch["images"])
    with torch.no_grad():
        out, _ = app(batch)
        px = L.preprocess_images(images, size=res, crop=res)
        ref = WK.wukong_forward(sd, cfg, px.cpu(), torch.from_numpy(g["input_ids"]))
    assert float((out["image_features"].cpu() - ref["image_features"]).abs().max()) < 1e-5
    assert float((out["text_features"].cpu() - ref["text_features"]).abs().max()) < 1e-5


def test_wukong_predictor_runs_the_reference_record_format(tmp_path):
    g, d, cfg, sd = _checkpoint(tmp_path)
    pred = WukongCLIPPredictor(d, first_sequence="text", second_sequence="image",
                               user_defined_parameters={"clip_compute_dtype": "fp32"})
    rows = g["tsv"].tobytes().decode("utf-8").split("\n")[:-1]
    trecs = [{"text": r.split("\t")[0]} for r in rows]
    out = pred.run(trecs)
    assert len(out) == 5 and set(out[0]) == {"text_feat"}
    feats = np.array([[float(x) for x in o["text_feat"].split("\t")] for o in out], np.float32)
    with torch.no_grad():
        ref = WK.wukong_forward(sd, cfg, None, torch.from_numpy(g["input_ids"]))["text_features"].numpy()
    assert np.abs(feats - ref).max() < 1e-5
    irecs = [{"image": r.split("\t")[1]} for r in rows[:2]]
    iout = pred.run(irecs)
    assert len(iout) == 2 and set(iout[0]) == {"image_feat"}
    res = cfg["model"]["visual"]["input_resolution"]
    imgs = [np.asarray(PIL.open(io.BytesIO(base64.urlsafe_b64decode(r["image"])))) for r in irecs]
    with torch.no_grad():
        px = L.preprocess_images(imgs, size=res, crop=res).cpu()
        iref = WK.wukong_forward(sd, cfg, px, None)["image_features"].numpy()
    ifeats = np.array([[float(x) for x in o["image_feat"].split("\t")] for o in iout], np.float32)
    assert np.abs(ifeats - iref).max() < 1e-5

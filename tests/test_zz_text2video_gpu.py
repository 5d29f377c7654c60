"""Drop-in Text2VideoRetrieval on the GPU (reference: appzoo/text2video_retrieval/model.py:39-121) against fixtures of the
REAL reference application (tools/make_golden.py: run_t2v_case).

Ordered last in the suite on purpose: it was written after the round's GPU minutes were spent, so the round-end run is the
first time it executes on hardware; everything it calls below the frame pooling (open_clip towers, similarity, InfoNCE) is
covered by test_openclip_gpu.py, the pooling itself by test_text2video_oracle.py on the CPU."""
import json
import os

import numpy as np
import pytest
import torch

from easynlp_amd.appzoo.text2video_retrieval import Text2VideoRetrieval, Text2VideoRetrievalEvaluator
from oracle import clip_oracle as O
from oracle import open_clip_oracle as OC
from oracle import text2video_oracle as TV

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg_name, B, T, wseed, iseed = [str(x) for x in z["meta"][:5]]
    return z, OC.OPENCLIP_CONFIGS[cfg_name], int(B), int(T), int(wseed), int(iseed)


def make_app(tmp_path, cfg, seed, dtype):
    sd = OC.make_state_dict(cfg, seed)
    with open(os.path.join(str(tmp_path), "config.json"), "w") as f:
        json.dump(cfg, f)
    torch.save({"open_clip." + k: v for k, v in sd.items()}, os.path.join(str(tmp_path), "pytorch_model.bin"))
    app = Text2VideoRetrieval(str(tmp_path), user_defined_parameters={"clip_compute_dtype": dtype}).cuda()
    assert app.model_type == "open_clip"
    return app, sd


@pytest.mark.parametrize("name", ["t2v_tiny_b4_t3", "t2v_small_b3_t5"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_text2video_forward_and_backward_match_reference_golden(tmp_path, name, dtype):
    z, cfg, B, T, wseed, iseed = load(name)
    app, sd = make_app(tmp_path, cfg, wseed, dtype)
    app.train()
    px, masks, ids = TV.make_inputs(cfg, B, T, iseed)
    out = app({"pixel_values": px.clone(), "video_masks": masks.clone(), "input_ids": ids.clone()})
    loss = app.compute_loss(out, [])["loss"]
    loss.backward()
    f32 = dtype == "fp32"
    for k in ("video_embeds", "text_embeds"):
        err = float((out[k].detach().cpu() - torch.from_numpy(z[k])).abs().max())
        assert err < (1e-5 if f32 else 1.5e-2), (k, err)
    assert tuple(out["logits_per_video"].shape) == (B, B)
    assert abs(loss.item() - float(z["loss"])) < (1e-5 if f32 else 2e-2)
    params = {n.replace("open_clip.", "", 1): p for n, p in app.named_parameters()}
    scale = {}
    for key in z.files:
        if key.startswith("gnorm/"):
            n = key.split("/", 1)[1]
            sk = ("visual" if n.startswith("visual") else "text", tuple(params[n].shape))
            scale[sk] = max(scale.get(sk, 0.0), float(z[key]))
    bad, seen = [], 0
    for key in z.files:
        if not key.startswith("gnorm/"):
            continue
        n = key.split("/", 1)[1]
        p = params[n]
        seen += 1
        floor = 0.0 if f32 else 2e-2 * scale[("visual" if n.startswith("visual") else "text", tuple(p.shape))]
        ref, got = float(z[key]), float(p.grad.double().norm())
        if abs(got - ref) > (2e-4 if f32 else 6e-2) * ref + floor + 1e-7:
            bad.append((n, got, ref))
    assert seen == len(OC.param_shapes(cfg)) and not bad, bad[:10]


def test_text2video_eval_path_and_evaluator(tmp_path):
    cfg = OC.OPENCLIP_CONFIGS["oc_small"]
    app, sd = make_app(tmp_path, cfg, 3, "bf16")
    app.eval()
    px, masks, ids = TV.make_inputs(cfg, 6, 4, 1)
    with torch.no_grad():
        ref = TV.forward(sd, cfg, px, masks, ids)
        out = app({"pixel_values": px.clone(), "video_masks": masks.clone(), "input_ids": ids.clone()})
        assert float((out["video_embeds"].cpu() - ref["video_embeds"]).abs().max()) < 1.5e-2
        assert float((out["text_embeds"].cpu() - ref["text_embeds"]).abs().max()) < 1.5e-2
        # frames behind the mask do not matter
        px2 = px.clone()
        px2[1, 1:] = 0.0
        out2 = app({"pixel_values": px2, "video_masks": masks.clone()}, feat=True)
        assert out2["text_embeds"] is None
        assert float((out2["video_embeds"][1] - out["video_embeds"][1]).abs().max()) < 1e-6

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 6

        def __getitem__(self, i):
            return {"pixel_values": px[i:i + 1], "video_masks": masks[i:i + 1], "input_ids": ids[i:i + 1]}

        @staticmethod
        def batch_fn(rows):
            return {k: torch.cat([r[k] for r in rows], dim=0) for k in rows[0]}

    res = Text2VideoRetrievalEvaluator(DS(), eval_batch_size=6).evaluate(app)
    want = O.recall_at_k(out["text_embeds"].cpu().float(), out["video_embeds"].cpu().float())
    assert res[0][0] == "mean_recall" and abs(res[0][1] - want[0]) < 1e-9


def test_text2video_dataset_batches_and_predictor(tmp_path):
    """DataLoader(Text2VideoRetrievalDataset) -> forward (frames resized on the GPU) and the predictor's two record kinds,
    against the oracle fed with the same pre-processed frames."""
    import gzip
    from easynlp_amd import lib as L
    from easynlp_amd.appzoo.text2video_retrieval import Text2VideoRetrievalDataset, Text2VideoRetrievalPredictor
    g = np.load(os.path.join(GOLD, "t2v_dataset_b3.npz"))
    bpe = np.load(os.path.join(GOLD, "openclip_bpe_corpus.npz"))
    d = str(tmp_path)
    cfg = dict(OC.OPENCLIP_CONFIGS["oc_small"], image_resolution=224, vision_patch_size=32,
               vocab_size=int(bpe["meta"][0]), context_length=77)
    sd = OC.make_state_dict(cfg, 9)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg, f)
    torch.save({"open_clip." + k: v for k, v in sd.items()}, os.path.join(d, "pytorch_model.bin"))
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
    ds = Text2VideoRetrievalDataset(d, os.path.join(d, "data.tsv"), 77, input_schema="text:str:1,image:str:1",
                                    first_sequence="text", second_sequence="image")
    app = Text2VideoRetrieval(d, user_defined_parameters={"clip_compute_dtype": "fp32"}).cuda().eval()
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False, collate_fn=ds.batch_fn)))
    clips = [list(c) for c in batch["images"]]
    with torch.no_grad():
        out = app(batch)
        px = L.preprocess_images([f for c in clips for f in c], size=224, crop=224).cpu().view(3, 12, 3, 224, 224)
        ref = TV.forward(sd, cfg, px, torch.from_numpy(g["video_masks"]), torch.from_numpy(g["input_ids"]))
    assert tuple(batch["pixel_values"].shape) == (36, 3, 224, 224)          # forward flattens the clips in place, as the reference
    assert float((out["video_embeds"].cpu() - ref["video_embeds"]).abs().max()) < 1e-5
    assert float((out["text_embeds"].cpu() - ref["text_embeds"]).abs().max()) < 1e-5
    vp = Text2VideoRetrievalPredictor(d, first_sequence="image", user_defined_parameters={"clip_compute_dtype": "fp32"})
    vout = vp.run([{"image": os.path.join(d, "clip%d" % i)} for i in range(3)])
    vf = np.array([[float(x) for x in o["video_feat"].split("\t")] for o in vout], np.float32)
    assert np.abs(vf - ref["video_embeds"].numpy()).max() < 1e-5
    tp = Text2VideoRetrievalPredictor(d, first_sequence="text", user_defined_parameters={"clip_compute_dtype": "fp32"})
    tout = tp.run([{"text": str(c)} for c in g["captions"]])
    tf = np.array([[float(x) for x in o["text_feat"].split("\t")] for o in tout], np.float32)
    assert np.abs(tf - ref["text_embeds"].numpy()).max() < 1e-5


def test_gpu_frame_pixel_values_equal_the_reference(tmp_path):
    """every frame's float32 pixel_values == the reference's, bit for bit (incl. the greyscale frame, up- and down-scaling,
    and the black padding frames)"""
    import hashlib
    from easynlp_amd import lib as L
    from tests.test_text2video_data import _dataset, _materialise
    g, d = _materialise(tmp_path)
    ds = _dataset(d)
    batch = ds.batch_fn([ds[i] for i in range(len(ds))])
    flat = [f for c in batch["images"] for f in c]
    px = L.preprocess_images(flat, size=224, crop=224).cpu().numpy().reshape(3, 12, 3, 224, 224)
    want = dict(zip([str(x) for x in g["frame_names"]], [str(x) for x in g["frame_sha256"]]))
    for ci in range(3):
        names = os.listdir(os.path.join(d, "clip%d" % ci))
        for fi, name in enumerate(names):
            assert hashlib.sha256(np.ascontiguousarray(px[ci, fi]).tobytes()).hexdigest() == want["clip%d/%s" % (ci, name)], (ci, name)
        for fi in range(len(names), 12):
            assert hashlib.sha256(np.ascontiguousarray(px[ci, fi]).tobytes()).hexdigest() == str(g["pad_sha256"])

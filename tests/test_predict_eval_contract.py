"""Drop-in boundary, evaluate and predict: the reference CLIPEvaluator / CLIPPredictor on the reference CLIPApp
against the drop-in CLIPEvaluator / CLIPPredictor on the drop-in CLIPApp, same checkpoint directory, same TSV / records, on
the CPU.  Device compute of the drop-in is stood in for by the oracle (encodes, image pre-processing, recall ranks), so the
test covers the surrounding contract: record formats in, feature strings out, which modality a record exports, default
arguments, the evaluator's metric tuple."""
import os

import numpy as np
import pytest
import torch

from oracle import clip_oracle as O
from oracle import preprocess_oracle as P
from oracle import ref_harness as R

GOLD = os.path.join(os.path.dirname(__file__), "golden", "dataset_tsv_b7.npz")
SCHEMA = dict(input_schema="text:str:1,image:str:1", first_sequence="text", second_sequence="image")


@pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")
def test_reference_and_dropin_predictor_and_evaluator_agree(tmp_path, monkeypatch):
    R.install_shims()
    from easynlp.appzoo.clip.data import CLIPDataset as RefDataset
    from easynlp.appzoo.clip.evaluator import CLIPEvaluator as RefEvaluator
    from easynlp.appzoo.clip.model import CLIPApp as RefApp
    from easynlp.appzoo.clip.predictor import CLIPPredictor as RefPredictor
    from easynlp_amd import lib as L
    from easynlp_amd.appzoo.clip import CLIPEvaluator, CLIPPredictor
    from easynlp_amd.appzoo.clip import evaluator as EV
    from easynlp_amd.appzoo.clip import model as CM
    from easynlp_amd.appzoo.clip.data import CLIPDataset

    g = np.load(GOLD)
    d = str(tmp_path)
    vocab = g["vocab"].tobytes().decode().split("\n")
    cfg = dict(O.CONFIGS["tiny"], vocab_size=len(vocab), image_resolution=224, vision_patch_size=32, vision_width=64, vision_layers=1)
    R.write_checkpoint_dir(d, cfg, O.make_state_dict(cfg, 8))
    with open(os.path.join(d, "vocab.txt"), "wb") as f:
        f.write(g["vocab"].tobytes() + b"\n")
    tsv = os.path.join(d, "valid.tsv")
    with open(tsv, "wb") as f:
        f.write(g["tsv"].tobytes())
    rows = [r.split("\t") for r in g["tsv"].tobytes().decode("utf-8").split("\n")[:-1]]

    # no GPU here: .cuda() is the identity for both implementations, the drop-in's device work is the oracle's
    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)

    def oracle_preprocess(images, size=224, crop=224, mean=L.CLIP_MEAN, std=L.CLIP_STD, device="cpu"):
        outs = []
        for im in images:
            a = np.asarray(im)
            outs.append(P.preprocess(np.repeat(a[:, :, None], 3, axis=2) if a.ndim == 2 else a, size=size, crop=crop))
        return torch.from_numpy(np.stack(outs))

    def oracle_encode(self, pixel_values=None, input_ids=None, token_type_ids=None, attention_mask=None, pack_hint=None):
        sd = {n: p for n, p in self.chinese_clip.named_parameters()}
        return (O.encode_image(sd, self.raw_config, pixel_values) if pixel_values is not None else None,
                O.encode_text(sd, self.raw_config, input_ids) if input_ids is not None else None)

    def oracle_recall(t, v, ks=(1, 5, 10)):
        r = O.recall_at_k(t.float(), v.float())
        return r, tuple(int(round(x * t.shape[0])) for x in r[1:])

    class OracleSimilarity:
        apply = staticmethod(lambda t, i, ls: (t @ i.t()) * ls.exp())

    monkeypatch.setattr(L, "preprocess_images", oracle_preprocess)
    monkeypatch.setattr(CM.CLIPApp, "encode", oracle_encode)
    monkeypatch.setattr(CM, "_SimilarityFn", OracleSimilarity)
    monkeypatch.setattr(EV, "recall_at_k", oracle_recall)

    def feats(out, key):
        assert all(set(o) == {key} for o in out)
        return np.array([[float(x) for x in o[key].split("\t")] for o in out], np.float32)

    # ---- predictor: image records, text records, records carrying both (the text wins, predictor.py:118-138)
    ref_p = RefPredictor(d, first_sequence="text", second_sequence="image", sequence_length=20)
    my_p = CLIPPredictor(d, first_sequence="text", second_sequence="image", sequence_length=20)
    for make, key in ((lambda r: {"image": r[1]}, "image_feat"), (lambda r: {"text": r[0]}, "text_feat"),
                      (lambda r: {"text": r[0], "image": r[1]}, "text_feat")):
        ref_out = ref_p.run([make(r) for r in rows])
        my_out = my_p.run([make(r) for r in rows])
        a, b = feats(ref_out, key), feats(my_out, key)
        assert a.shape == b.shape == (7, cfg["embed_dim"]) and np.abs(a - b).max() < 2e-6, key
    # per-record sequence_length overrides the constructor's (predictor.py:83-87)
    a = ref_p.preprocess([{"text": rows[0][0], "sequence_length": 12}])[0]["input_ids"]
    b = my_p.preprocess([{"text": rows[0][0], "sequence_length": 12}])[0]["input_ids"]
    assert tuple(a.shape) == tuple(b.shape) == (1, 12) and torch.equal(a, b)
    # defaults of the constructor (predictor.py:66-68)
    assert CLIPPredictor(d).sequence_length == RefPredictor(d).sequence_length == 128

    # ---- the --mode predict pipeline: the reference PredictorManager drives either predictor over a TSV (core/predictor.py:181-229)
    from easynlp.core.predictor import PredictorManager
    outs = {}
    for name, pred in (("reference", ref_p), ("dropin", my_p)):
        outs[name] = os.path.join(d, "pred_%s.tsv" % name)
        PredictorManager(predictor=pred, input_file=tsv, input_schema="text:str:1,image:str:1", output_file=outs[name],
                         output_schema="text_feat", append_cols="text", batch_size=3).run()
    la, lb = (open(outs[k]).read().split("\n") for k in ("reference", "dropin"))
    assert len(la) == len(lb) == 8 and la[-1] == lb[-1] == ""
    for ra, rb, row in zip(la[:-1], lb[:-1], rows):
        fa, fb = ra.split("\t"), rb.split("\t")
        assert len(fa) == len(fb) == cfg["embed_dim"] + 1 and fa[-1] == fb[-1] == row[0]          # features ..., appended text column
        assert np.abs(np.array(fa[:-1], np.float32) - np.array(fb[:-1], np.float32)).max() < 2e-6

    # ---- evaluator
    ref_ev = RefEvaluator(valid_dataset=RefDataset(d, tsv, 20, **SCHEMA), eval_batch_size=4)
    my_ev = CLIPEvaluator(valid_dataset=CLIPDataset(d, tsv, 20, **SCHEMA), eval_batch_size=4)
    ref_res = ref_ev.evaluate(RefApp(d))
    my_res = my_ev.evaluate(CM.CLIPApp(d))
    assert ref_res[0][0] == my_res[0][0] == "mean_recall" and abs(ref_res[0][1] - my_res[0][1]) < 1e-9


@pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")
def test_reference_and_dropin_wukong_predictor_agree(tmp_path, monkeypatch):
    """WukongCLIPPredictor (wukong_clip/predictor.py:30-139) on WukongCLIP: reference vs drop-in, same records (RGB rows of
    the TSV fixture, BERT-layout vocabulary with [SEP] = 102), drop-in device compute stood in for by the oracle."""
    import json
    R.install_shims()
    from easynlp.appzoo.wukong_clip.predictor import WukongCLIPPredictor as RefPredictor
    from easynlp_amd import lib as L
    from easynlp_amd.appzoo.clip import model as CM
    from easynlp_amd.appzoo.wukong_clip import WukongCLIPPredictor
    from oracle import wukong_oracle as WK
    g = np.load(os.path.join(os.path.dirname(GOLD), "wukong_dataset_b5.npz"))
    d = str(tmp_path)
    vocab = g["vocab"].tobytes().decode("utf-8").split("\n")
    cfg = {"model": {"visual": dict(input_resolution=224, patch_size=32, width=64, layers=1, heads=1, output_dim=64),
                     "text": dict(context_length=32, vocab_size=len(vocab), output_dim=64, width=64, layers=2, heads=1)}}
    sd = WK.make_state_dict(cfg, 4, small_embeddings=False)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg, f)
    torch.save(sd, os.path.join(d, "pytorch_model.bin"))
    with open(os.path.join(d, "vocab.txt"), "wb") as f:
        f.write(g["vocab"].tobytes() + b"\n")
    rows = [r.split("\t") for r in g["tsv"].tobytes().decode("utf-8").split("\n")[:-1]]

    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)

    def oracle_preprocess(images, size=224, crop=224, mean=L.CLIP_MEAN, std=L.CLIP_STD, device="cpu"):
        return torch.from_numpy(np.stack([P.preprocess(np.asarray(im), size=size, crop=crop) for im in images]))

    def oracle_encode(self, pixel_values=None, input_ids=None, token_type_ids=None, attention_mask=None, pack_hint=None):
        fo = WK.wukong_forward({"model." + n: p for n, p in self.model.named_parameters()}, self.raw_config, pixel_values, input_ids)
        return fo["image_features"], fo["text_features"]

    monkeypatch.setattr(L, "preprocess_images", oracle_preprocess)
    monkeypatch.setattr(CM.CLIPApp, "encode", oracle_encode)
    ref_p = RefPredictor(d, first_sequence="text", second_sequence="image")
    my_p = WukongCLIPPredictor(d, first_sequence="text", second_sequence="image")
    for make, key in ((lambda r: {"image": r[1]}, "image_feat"), (lambda r: {"text": r[0]}, "text_feat")):
        with torch.no_grad():
            ref_out = ref_p.run([make(r) for r in rows])
        my_out = my_p.run([make(r) for r in rows])
        assert all(set(o) == {key} for o in my_out) and len(ref_out) == len(my_out) == 5
        a = np.array([[float(x) for x in o[key].split("\t")] for o in ref_out], np.float32)
        b = np.array([[float(x) for x in o[key].split("\t")] for o in my_out], np.float32)
        assert np.abs(a - b).max() < 2e-6, key
    assert torch.equal(ref_p.tokenize(["a photo of a cat", ""]), my_p.tokenize(["a photo of a cat", ""]))
    # evaluator (wukong_clip/evaluator.py:26-80), each on its own dataset class
    from easynlp.appzoo.wukong_clip.data import WukongCLIPDataset as RefDataset
    from easynlp.appzoo.wukong_clip.evaluator import WukongCLIPEvaluator as RefEvaluator
    from easynlp.appzoo.wukong_clip.model import WukongCLIP as RefApp
    from easynlp_amd.appzoo.wukong_clip import WukongCLIP, WukongCLIPDataset, WukongCLIPEvaluator
    from easynlp_amd.appzoo.clip import evaluator as EV

    def oracle_recall(t, v, ks=(1, 5, 10)):
        r = O.recall_at_k(t.float(), v.float())
        return r, tuple(int(round(x * t.shape[0])) for x in r[1:])
    monkeypatch.setattr(EV, "recall_at_k", oracle_recall)
    tsv = os.path.join(d, "valid.tsv")
    with open(tsv, "wb") as f:
        f.write(g["tsv"].tobytes())
    ref_res = RefEvaluator(valid_dataset=RefDataset(d, tsv, 32, **SCHEMA), user_defined_parameters={}, eval_batch_size=2).evaluate(RefApp(d))
    my_res = WukongCLIPEvaluator(valid_dataset=WukongCLIPDataset(d, tsv, 32, **SCHEMA), user_defined_parameters={},
                                 eval_batch_size=2).evaluate(WukongCLIP(d))
    assert ref_res[0][0] == my_res[0][0] == "mean_recall" and abs(ref_res[0][1] - my_res[0][1]) < 1e-9


@pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")
def test_reference_and_dropin_text2video_predictor_and_evaluator_agree(tmp_path, monkeypatch):
    """Text2VideoRetrievalPredictor / -Evaluator (text2video_retrieval/{predictor,evaluator}.py): reference vs drop-in on the
    frame-directory fixture; the NAME of first_sequence selects the modality ('image' = a directory of frames)."""
    import gzip
    import json
    R.install_shims()
    from easynlp.appzoo.text2video_retrieval.data import Text2VideoRetrievalDataset as RefDataset
    from easynlp.appzoo.text2video_retrieval.evaluator import Text2VideoRetrievalEvaluator as RefEvaluator
    from easynlp.appzoo.text2video_retrieval.model import Text2VideoRetrieval as RefApp
    from easynlp.appzoo.text2video_retrieval.predictor import Text2VideoRetrievalPredictor as RefPredictor
    from easynlp_amd import lib as L
    from easynlp_amd.appzoo.clip import model as CM
    from easynlp_amd.appzoo.text2video_retrieval import (Text2VideoRetrieval, Text2VideoRetrievalDataset,
                                                          Text2VideoRetrievalEvaluator, Text2VideoRetrievalPredictor)
    from easynlp_amd.appzoo.clip import evaluator as EV
    from easynlp_amd.appzoo.text2video_retrieval import model as TM
    from oracle import open_clip_oracle as OC
    gold = os.path.dirname(GOLD)
    bpe, g = np.load(os.path.join(gold, "openclip_bpe_corpus.npz")), np.load(os.path.join(gold, "t2v_dataset_b3.npz"))
    d = str(tmp_path)
    cfg = dict(OC.OPENCLIP_CONFIGS["oc_tiny"], image_resolution=224, vision_patch_size=32, vision_width=64, vision_layers=1,
               context_length=77, vocab_size=int(bpe["meta"][0]))
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg, f)
    torch.save({"open_clip." + k: v for k, v in OC.make_state_dict(cfg, 6).items()}, os.path.join(d, "pytorch_model.bin"))
    with gzip.open(os.path.join(d, "vocab.txt"), "wb") as f:
        f.write(bpe["merges"].tobytes())
    for k in g.files:
        if k.startswith("png/"):
            os.makedirs(os.path.dirname(os.path.join(d, k[4:])), exist_ok=True)
            with open(os.path.join(d, k[4:]), "wb") as f:
                f.write(g[k].tobytes())
    clips = [os.path.join(d, "clip%d" % i) for i in range(3)]
    tsv = os.path.join(d, "valid.tsv")
    with open(tsv, "w") as f:
        for cap, c in zip(g["captions"], clips):
            f.write("%s\t%s\n" % (cap, c))

    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)

    def oracle_preprocess(images, size=224, crop=224, mean=L.CLIP_MEAN, std=L.CLIP_STD, device="cpu"):
        outs = []
        for im in images:
            a = np.asarray(im)
            outs.append(P.preprocess(np.repeat(a[:, :, None], 3, axis=2) if a.ndim == 2 else a, size=size, crop=crop))
        return torch.from_numpy(np.stack(outs))

    def oracle_encode(self, pixel_values=None, input_ids=None, token_type_ids=None, attention_mask=None, pack_hint=None):
        sd = {n: p for n, p in self.open_clip.named_parameters()}
        img = O.l2_normalize(O.vit_forward(sd, OC.chinese_style_config(cfg), pixel_values)) if pixel_values is not None else None
        txt = O.l2_normalize(OC.text_forward(sd, cfg, input_ids)) if input_ids is not None else None
        return img, txt

    def oracle_recall(t, v, ks=(1, 5, 10)):
        r = O.recall_at_k(t.float(), v.float())
        return r, tuple(int(round(x * t.shape[0])) for x in r[1:])

    class OracleSimilarity:
        apply = staticmethod(lambda t, i, ls: (t @ i.t()) * ls.exp())

    monkeypatch.setattr(L, "preprocess_images", oracle_preprocess)
    monkeypatch.setattr(CM.CLIPApp, "encode", oracle_encode)
    monkeypatch.setattr(TM, "_SimilarityFn", OracleSimilarity)
    monkeypatch.setattr(EV, "recall_at_k", oracle_recall)

    def feats(out, key):
        assert all(set(o) == {key} for o in out)
        return np.array([[float(x) for x in o[key].split("\t")] for o in out], np.float32)

    for first, records, key in (("image", [{"image": c} for c in clips], "video_feat"),
                                ("text", [{"text": str(c)} for c in g["captions"]], "text_feat")):
        with torch.no_grad():
            ref_out = RefPredictor(d, first_sequence=first).run([dict(r) for r in records])
        my_out = Text2VideoRetrievalPredictor(d, first_sequence=first).run([dict(r) for r in records])
        a, b = feats(ref_out, key), feats(my_out, key)
        assert a.shape == b.shape == (3, cfg["embed_dim"]) and np.abs(a - b).max() < 2e-6, key
    schema = dict(input_schema="text:str:1,image:str:1", first_sequence="text", second_sequence="image")
    ref_res = RefEvaluator(valid_dataset=RefDataset(d, tsv, 77, **schema), eval_batch_size=2).evaluate(RefApp(d))
    my_res = Text2VideoRetrievalEvaluator(valid_dataset=Text2VideoRetrievalDataset(d, tsv, 77, **schema),
                                          eval_batch_size=2).evaluate(Text2VideoRetrieval(d))
    assert ref_res[0][0] == my_res[0][0] == "mean_recall" and abs(ref_res[0][1] - my_res[0][1]) < 1e-9


@pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")
def test_forward_contract_matches_the_reference_app(tmp_path, monkeypatch):
    """CLIPApp.forward / compute_loss (appzoo/clip/model.py:106-164), reference vs drop-in (oracle compute): returned keys,
    which entries are None, shapes and values for both / one modality and feat=True, and the in-place mutation of the
    ``inputs`` dict (model.py:116-123) that callers such as the evaluator rely on."""
    R.install_shims()
    from easynlp.appzoo.clip.model import CLIPApp as RefApp
    from easynlp_amd.appzoo.clip import model as CM
    cfg = O.CONFIGS["tiny"]
    R.write_checkpoint_dir(str(tmp_path), cfg, O.make_state_dict(cfg, 2))

    def oracle_encode(self, pixel_values=None, input_ids=None, token_type_ids=None, attention_mask=None, pack_hint=None):
        sd = {n: p for n, p in self.chinese_clip.named_parameters()}
        return (O.encode_image(sd, self.raw_config, pixel_values) if pixel_values is not None else None,
                O.encode_text(sd, self.raw_config, input_ids) if input_ids is not None else None)

    class OracleSimilarity:
        apply = staticmethod(lambda t, i, ls: (t @ i.t()) * ls.exp())

    class OracleInfoNCE:
        apply = staticmethod(lambda logits: O.clip_loss(logits))

    monkeypatch.setattr(CM.CLIPApp, "encode", oracle_encode)
    monkeypatch.setattr(CM, "_SimilarityFn", OracleSimilarity)
    monkeypatch.setattr(CM, "_InfoNCEFn", OracleInfoNCE)
    ref, mine = RefApp(str(tmp_path)).eval(), CM.CLIPApp(str(tmp_path)).eval()
    px, ids = O.make_inputs(cfg, 5, 12, 1)
    tt, am = torch.zeros_like(ids), (ids != 0).long()

    def same(a, b, path):
        assert (a is None) == (b is None), path
        if a is None:
            return
        if isinstance(a, torch.Tensor):
            assert isinstance(b, torch.Tensor) and a.shape == b.shape and a.dtype == b.dtype, path
            assert float((a.float() - b.float()).abs().max()) < 2e-5, path
        else:
            assert a == b, path

    cases = [dict(pixel_values=px, input_ids=ids, token_type_ids=tt, attention_mask=am, label_ids=[]),
             dict(pixel_values=px, input_ids=ids), dict(input_ids=ids), dict(pixel_values=px)]
    # (an explicit ``None`` value fails in the reference, which tests key presence only, model.py:116-123; the drop-in accepts it)
    for case in cases:
        for feat in (None, True):
            if feat is None and not (case.get("pixel_values") is not None and case.get("input_ids") is not None):
                continue                              # logits need both modalities (the reference fails in torch.matmul)
            ia = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in case.items()}
            ib = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in case.items()}
            with torch.no_grad():
                oa, ob = ref(ia, feat=feat), mine(ib, feat=feat)
            assert set(oa) == set(ob), (sorted(case), feat)
            for k in oa:
                same(oa[k], ob[k], (sorted(case), feat, k))
            assert set(ia) == set(ib), (sorted(case), feat)             # the same keys were added to / kept in `inputs`
            for k in ia:
                same(ia[k], ib[k], (sorted(case), feat, "inputs." + k))
    # the loss dict, through autograd
    oa = ref({"pixel_values": px.clone(), "input_ids": ids.clone()})
    ob = mine({"pixel_values": px.clone(), "input_ids": ids.clone()})
    la, lb = ref.compute_loss(oa, [])["loss"], mine.compute_loss(ob, [])["loss"]
    assert la.shape == lb.shape == () and abs(la.item() - lb.item()) < 1e-5
    la.backward()
    lb.backward()
    ga = dict(ref.named_parameters())
    for n, p in mine.named_parameters():
        if ga[n].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
        else:
            assert float((p.grad - ga[n].grad).norm()) <= 1e-4 * float(ga[n].grad.norm()) + 1e-7, n
    # neither modality: both refuse
    with pytest.raises(Exception):
        ref({}, feat=True)
    with pytest.raises(Exception):
        mine({}, feat=True)


@pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")
def test_open_clip_flavour_predictor_and_dataset_agree_with_the_reference(tmp_path, monkeypatch):
    """model_type open_clip: raw captions go through the BPE tokenizer in both CLIPPredictor and CLIPDataset
    (predictor.py:91-94, data.py:246-249); reference vs drop-in on the same records / TSV."""
    import gzip
    import json
    R.install_shims()
    from easynlp.appzoo.clip.data import CLIPDataset as RefDataset
    from easynlp.appzoo.clip.predictor import CLIPPredictor as RefPredictor
    from easynlp_amd import lib as L
    from easynlp_amd.appzoo.clip import CLIPPredictor
    from easynlp_amd.appzoo.clip import model as CM
    from easynlp_amd.appzoo.clip.data import CLIPDataset
    from oracle import open_clip_oracle as OC
    gold = os.path.dirname(GOLD)
    bpe, g = np.load(os.path.join(gold, "openclip_bpe_corpus.npz")), np.load(GOLD)
    d = str(tmp_path)
    cfg = dict(OC.OPENCLIP_CONFIGS["oc_tiny"], image_resolution=224, vision_patch_size=32, vision_width=64, vision_layers=1,
               context_length=77, vocab_size=int(bpe["meta"][0]))
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg, f)
    torch.save({"open_clip." + k: v for k, v in OC.make_state_dict(cfg, 6).items()}, os.path.join(d, "pytorch_model.bin"))
    with gzip.open(os.path.join(d, "vocab.txt"), "wb") as f:
        f.write(bpe["merges"].tobytes())
    tsv = os.path.join(d, "valid.tsv")
    with open(tsv, "wb") as f:
        f.write(g["tsv"].tobytes())
    rows = [r.split("\t") for r in g["tsv"].tobytes().decode("utf-8").split("\n")[:-1]]

    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)

    def oracle_preprocess(images, size=224, crop=224, mean=L.CLIP_MEAN, std=L.CLIP_STD, device="cpu"):
        outs = []
        for im in images:
            a = np.asarray(im)
            outs.append(P.preprocess(np.repeat(a[:, :, None], 3, axis=2) if a.ndim == 2 else a, size=size, crop=crop))
        return torch.from_numpy(np.stack(outs))

    def oracle_encode(self, pixel_values=None, input_ids=None, token_type_ids=None, attention_mask=None, pack_hint=None):
        sd = {n: p for n, p in self.open_clip.named_parameters()}
        img = O.l2_normalize(O.vit_forward(sd, OC.chinese_style_config(cfg), pixel_values)) if pixel_values is not None else None
        txt = O.l2_normalize(OC.text_forward(sd, cfg, input_ids)) if input_ids is not None else None
        return img, txt

    monkeypatch.setattr(L, "preprocess_images", oracle_preprocess)
    monkeypatch.setattr(CM.CLIPApp, "encode", oracle_encode)
    ref_p = RefPredictor(d, first_sequence="text", second_sequence="image")
    my_p = CLIPPredictor(d, first_sequence="text", second_sequence="image")
    for make, key in ((lambda r: {"text": r[0]}, "text_feat"), (lambda r: {"image": r[1]}, "image_feat")):
        ref_out, my_out = ref_p.run([make(r) for r in rows]), my_p.run([make(r) for r in rows])
        a = np.array([[float(x) for x in o[key].split("\t")] for o in ref_out], np.float32)
        b = np.array([[float(x) for x in o[key].split("\t")] for o in my_out], np.float32)
        assert a.shape == b.shape == (7, cfg["embed_dim"]) and np.abs(a - b).max() < 2e-6, key
    schema = dict(input_schema="text:str:1,image:str:1", first_sequence="text", second_sequence="image")
    rds, mds = RefDataset(d, tsv, 32, **schema), CLIPDataset(d, tsv, 32, **schema)
    rb, mb = rds.batch_fn([rds[i] for i in range(7)]), mds.batch_fn([mds[i] for i in range(7)])
    assert torch.equal(rb["input_ids"], mb["input_ids"]) and tuple(mb["input_ids"].shape) == (7, 77)
    assert torch.equal(rb["pixel_values"], oracle_preprocess(mb["images"]))          # the oracle IS the reference's PIL pipeline


@pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")
def test_huggingface_flavour_predictor_agrees_with_the_reference(tmp_path, monkeypatch):
    """huggingface_clip: text records carry token_type_ids / attention_mask into RobertaModel (predictor.py:95-101,124-134)"""
    R.install_shims()
    from easynlp.appzoo.clip.predictor import CLIPPredictor as RefPredictor
    from easynlp_amd import lib as L
    from easynlp_amd.appzoo.clip import CLIPPredictor
    from easynlp_amd.appzoo.clip import model as CM
    from oracle import hf_clip_oracle as H
    g = np.load(GOLD)
    d = str(tmp_path)
    vocab = g["vocab"].tobytes().decode().split("\n")
    cfg = dict(text_config=dict(H.HF_CONFIGS["hf_tiny"]["text_config"], vocab_size=len(vocab)),
               vision_config=dict(H.HF_CONFIGS["hf_tiny"]["vision_config"], hidden_size=64, intermediate_size=256, num_hidden_layers=1,
                                  num_attention_heads=1, image_size=224, patch_size=32), projection_dim=64)
    R.write_hf_checkpoint_dir(d, cfg, H.make_state_dict(cfg, 3))
    with open(os.path.join(d, "vocab.txt"), "wb") as f:
        f.write(g["vocab"].tobytes() + b"\n")
    rows = [r.split("\t") for r in g["tsv"].tobytes().decode("utf-8").split("\n")[:-1]]
    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)

    def oracle_preprocess(images, size=224, crop=224, mean=L.CLIP_MEAN, std=L.CLIP_STD, device="cpu"):
        outs = []
        for im in images:
            a = np.asarray(im)
            outs.append(P.preprocess(np.repeat(a[:, :, None], 3, axis=2) if a.ndim == 2 else a, size=size, crop=crop))
        return torch.from_numpy(np.stack(outs))

    def oracle_encode(self, pixel_values=None, input_ids=None, token_type_ids=None, attention_mask=None, pack_hint=None):
        sd = dict(self._hf_params)
        img = txt = None
        if input_ids is not None:       # (hf_clip_forward wants both modalities: feed a dummy image / text for the other side)
            px = torch.zeros(input_ids.shape[0], 3, 224, 224)
            txt = H.hf_clip_forward(sd, cfg, px, input_ids, token_type_ids, attention_mask)["text_embeds"]
        if pixel_values is not None:
            ids = torch.ones(pixel_values.shape[0], 4, dtype=torch.long)
            img = H.hf_clip_forward(sd, cfg, pixel_values, ids, torch.zeros_like(ids), torch.ones_like(ids))["image_embeds"]
        return img, txt

    monkeypatch.setattr(L, "preprocess_images", oracle_preprocess)
    monkeypatch.setattr(CM.CLIPApp, "encode", oracle_encode)
    ref_p = RefPredictor(d, first_sequence="text", second_sequence="image", sequence_length=20)
    my_p = CLIPPredictor(d, first_sequence="text", second_sequence="image", sequence_length=20)
    for make, key in ((lambda r: {"text": r[0]}, "text_feat"), (lambda r: {"image": r[1]}, "image_feat")):
        with torch.no_grad():
            ref_out = ref_p.run([make(r) for r in rows])
        my_out = my_p.run([make(r) for r in rows])
        a = np.array([[float(x) for x in o[key].split("\t")] for o in ref_out], np.float32)
        b = np.array([[float(x) for x in o[key].split("\t")] for o in my_out], np.float32)
        assert a.shape == b.shape == (7, 64) and np.abs(a - b).max() < 2e-6, key


def test_recall_report_both_directions_on_stubbed_device_calls(monkeypatch, capsys):
    """``recall_report(..., both_directions=True)``: the reference's text -> image line and metric first, unchanged; one more printed
    line and an ``("i2t_mean_recall", ...)`` entry after it.  The two device calls of the fused sweep are stood in for by their counting
    definition on the CPU; small integer embeddings make every score exact and ties frequent."""
    import torch
    from oracle import clip_oracle as O
    import easynlp_amd.appzoo.clip.evaluator as EV
    g = torch.Generator().manual_seed(12)
    n = 23
    t = torch.randint(-3, 4, (n, 16), generator=g).float()
    v = t + torch.randint(-2, 3, (n, 16), generator=g).float()

    def block_cpu(t_rows, vv, row0, paired, out, cols):
        sim = t_rows @ vv.t()
        rows = sim.shape[0]
        i = torch.arange(row0, row0 + rows)[:, None]
        j = torch.arange(n)[None, :]
        d = paired[row0:row0 + rows][:, None]
        out[:rows] = ((sim > d) | ((sim == d) & (j < i))).sum(1).to(torch.int32)
        if cols is not None:
            dc = paired[None, :]
            cols += ((sim > dc) | ((sim == dc) & (i < j))).sum(0).to(torch.int32)

    monkeypatch.setattr(EV, "_paired_scores", lambda a, b: (a * b).sum(-1))
    monkeypatch.setattr(EV, "_ranks_block", block_cpu)
    want_r, want_c = O.recall_ranks(t, v)
    res = EV.recall_report(t, v, 0.5, both_directions=True)
    assert [k for k, _ in res] == ["mean_recall", "i2t_mean_recall"]
    mean = lambda r: sum(float((r < k).sum()) / n for k in (1, 5, 10)) / 3
    assert abs(res[0][1] - mean(want_r)) < 1e-12 and abs(res[1][1] - mean(want_c)) < 1e-12
    out = capsys.readouterr().out
    assert "query_num:%d" % n in out and "image->text" in out
    one = EV.recall_report(t, v, 0.5)
    assert one == [("mean_recall", res[0][1])]

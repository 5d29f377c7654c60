#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REAL reference (imported from
/root/reference, build container only) on the oracle's deterministic weights
and inputs.  Re-run:  python tools/make_golden.py

Each fixture stores the reference's outputs (embeddings, logits, loss), the
gradient of every parameter reduced to (L2 norm, 16 strided samples), and for
the small configs the full gradients.  torch / numpy versions are recorded.

It also stores ``bf16dev/<param>``: how far the same algorithm (the oracle
restatement) evaluated in bfloat16 by torch on the CPU lands from the reference fp32 gradient (relative
L2 error where the full gradient is stored, relative norm deviation otherwise).
On these random-init fixtures the towers rank-collapse, a few gradients are
differences of nearly equal terms and carry little relative information in bf16;
the bf16 GPU tests allow max(base tolerance, 1.5 x this measured reference-bf16
deviation) per parameter instead of hand-picked exceptions.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import clip_oracle as O  # noqa: E402
from oracle import ref_harness as R  # noqa: E402

CASES = [
    # name, config, batch, seq_len, weight seed, input seed, full grads?
    ("tiny_b6_l24", "tiny", 6, 24, 1234, 3, True),
    ("small_b5_l40", "small", 5, 40, 99, 7, False),
    ("p14_w256_b16_l32", "p14_w256", 16, 32, 77, 5, False),
    ("vitb16_bertbase_b4_l64", "vitb16_bertbase", 4, 64, 1234, 0, False),
    ("large_text_b24_l40", "large_text", 24, 40, 31, 11, False),      # round 4: text width 1024 / 16 heads / FFN 4096
    # round 6: the full-depth model with every residual branch's output layer at 0.3 of its random-init scale (clip_oracle.make_state_dict
    # residual_gain): NOT rank-collapsed -- the fixture on which bf16 query / key gradients at depth mean something
    ("vitb16_bertbase_rg03_b4_l64", "vitb16_bertbase", 4, 64, 1234, 0, False, 0.3),
]


def grad_digest(g: torch.Tensor):
    flat = g.reshape(-1)
    n = flat.numel()
    idx = torch.linspace(0, n - 1, 16).long()
    return float(flat.double().norm()), flat[idx].numpy().astype(np.float32), idx.numpy()


def run_case(name, cfg_name, B, L, wseed, iseed, full, residual_gain=1.0):
    torch.manual_seed(0)
    cfg = O.CONFIGS[cfg_name]
    sd = O.make_state_dict(cfg, wseed, residual_gain)
    model = R.reference_chinese_clip(cfg, sd)
    px, ids = O.make_inputs(cfg, B, L, iseed)
    img, txt = model(px, ids)                                   # modeling_chineseclip.py:352-365
    lpt = torch.matmul(txt, img.t()) * model.logit_scale.exp()  # appzoo/clip/model.py:148
    ar = torch.arange(B)
    loss = (torch.nn.functional.cross_entropy(lpt, ar)
            + torch.nn.functional.cross_entropy(lpt.T, ar)) / 2.0   # model.py:154-160
    loss.backward()
    out = {
        "meta": np.array([cfg_name, str(B), str(L), str(wseed), str(iseed),
                          torch.__version__, np.__version__]),
        "image_embeds": img.detach().numpy(), "text_embeds": txt.detach().numpy(),
        "logits_per_text": lpt.detach().numpy(), "loss": np.float32(loss.item()),
    }
    if residual_gain != 1.0:
        out["residual_gain"] = np.float64(residual_gain)
    # the same algorithm in bfloat16 (torch CPU): the intrinsic bf16 noise of these gradients.  The reference's
    # own LayerNorm subclass cannot run with bf16 parameters on the CPU ("mixed dtype (CPU)"), so this leg uses
    # the oracle restatement (pinned to the reference at 2e-6 in fp32 by tests/test_oracle.py) in bf16.
    _, loss16, g16 = O.forward_loss_backward(sd, cfg, px, ids, dtype=torch.bfloat16)
    out["loss_bf16_reference"] = np.float32(loss16.item())
    for n, p in model.named_parameters():
        if p.grad is None:
            out["nograd/" + n] = np.zeros(0, np.float32)
            continue
        norm, samp, idx = grad_digest(p.grad)
        out["gnorm/" + n] = np.float64(norm)
        out["gsamp/" + n] = samp
        g = g16[n].double()
        if full:
            out["grad/" + n] = p.grad.numpy()
            dev = float((g - p.grad.double()).norm()) / (norm + 1e-30)
        else:
            dev = abs(float(g.norm()) - norm) / (norm + 1e-30)
        out["bf16dev/" + n] = np.float64(dev)
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "loss", loss.item(), "->", path, os.path.getsize(path) // 1024, "KiB")


def run_dropout_case(name="tiny_dropout_b6_l24", cfg_name="tiny", B=6, L=24, wseed=1234, iseed=3, p_hidden=0.1, p_attn=0.15):
    """Train-mode pass of the reference with BERT dropout on.  ``torch.nn.functional.dropout`` is wrapped so that the
    keep mask of every active call is recorded (mask = output != 0; positions whose input is exactly 0 are irrelevant
    to values and gradients alike) -- the arithmetic stays the reference's own.  The oracle must reproduce outputs
    and all gradients when it replays those masks (tests/test_oracle.py)."""
    torch.manual_seed(0)
    cfg = dict(O.CONFIGS[cfg_name], text_hidden_dropout_prob=p_hidden, text_attention_probs_dropout_prob=p_attn)
    sd = O.make_state_dict(cfg, wseed)
    model = R.reference_chinese_clip(cfg, sd)
    model.train()
    px, ids = O.make_inputs(cfg, B, L, iseed)
    calls = []
    real = torch.nn.functional.dropout

    def recording_dropout(input, p=0.5, training=True, inplace=False):
        out = real(input, p, training, False)
        if training and p > 0:
            calls.append((float(p), ((out != 0) | (input == 0)).detach().clone()))
        return out

    torch.nn.functional.dropout = recording_dropout
    try:
        torch.manual_seed(4321)
        img, txt = model(px, ids)
    finally:
        torch.nn.functional.dropout = real
    nl = cfg["text_num_hidden_layers"]
    assert len(calls) == 1 + 3 * nl, len(calls)                 # emb, then (attn, self_out, out) per layer
    lpt = torch.matmul(txt, img.t()) * model.logit_scale.exp()
    ar = torch.arange(B)
    loss = (torch.nn.functional.cross_entropy(lpt, ar) + torch.nn.functional.cross_entropy(lpt.T, ar)) / 2.0
    loss.backward()
    out = {
        "meta": np.array([cfg_name, str(B), str(L), str(wseed), str(iseed), torch.__version__, np.__version__]),
        "p_hidden": np.float32(p_hidden), "p_attn": np.float32(p_attn),
        "image_embeds": img.detach().numpy(), "text_embeds": txt.detach().numpy(),
        "logits_per_text": lpt.detach().numpy(), "loss": np.float32(loss.item()),
    }
    names = ["emb"] + [f"{i}.{k}" for i in range(nl) for k in ("attn", "self_out", "out")]
    for n, (p, m) in zip(names, calls):
        assert abs(p - (p_attn if n.endswith("attn") else p_hidden)) < 1e-9, (n, p)
        out["mask/" + n] = np.packbits(m.numpy().reshape(-1))
        out["mshape/" + n] = np.array(m.shape, np.int64)
    for n, p in model.named_parameters():
        if p.grad is None:
            out["nograd/" + n] = np.zeros(0, np.float32)
        else:
            out["grad/" + n] = p.grad.numpy()
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "loss", loss.item(), "->", path, os.path.getsize(path) // 1024, "KiB")


def run_hf_case(name, cfg_name, B, L, wseed, iseed, full):
    """The REAL reference CLIPApp in huggingface_clip mode (appzoo/clip/model.py:73-104,128-150: RobertaModel pooled
    output + CLIPVisionModel pooled output detached + biased projections), loaded from a synthetic checkpoint directory:
    outputs, loss (CLIPApp.compute_loss) and the gradient of every parameter."""
    import tempfile
    from oracle import hf_clip_oracle as H
    R.install_shims()
    from easynlp.appzoo.clip.model import CLIPApp
    torch.manual_seed(0)
    cfg = H.HF_CONFIGS[cfg_name]
    sd = H.make_state_dict(cfg, wseed)
    d = tempfile.mkdtemp()
    R.write_hf_checkpoint_dir(d, cfg, sd)
    app = CLIPApp(d)
    assert app.model_type == "huggingface_clip"
    app.eval()
    px, ids, tt, am = H.make_inputs(cfg, B, L, iseed)
    fo = app({"pixel_values": px, "input_ids": ids, "token_type_ids": tt, "attention_mask": am})
    loss = app.compute_loss(fo, [])["loss"]
    loss.backward()
    out = {"meta": np.array([cfg_name, str(B), str(L), str(wseed), str(iseed), torch.__version__, np.__version__]),
           "image_embeds": fo["image_embeds"].detach().numpy(), "text_embeds": fo["text_embeds"].detach().numpy(),
           "logits_per_text": fo["logits_per_text"].detach().numpy(), "loss": np.float32(loss.item())}
    for n, p in app.named_parameters():
        if p.grad is None:
            out["nograd/" + n] = np.zeros(0, np.float32)
        elif full:
            out["grad/" + n] = p.grad.numpy()
        else:
            norm, samp, idx = grad_digest(p.grad)
            out["gnorm/" + n] = np.float64(norm)
            out["gsamp/" + n] = samp
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "loss", loss.item(), "->", path, os.path.getsize(path) // 1024, "KiB")


def run_openclip_case(name, cfg_name, B, wseed, iseed, full):
    """The REAL reference CLIPApp in open_clip mode (appzoo/clip/model.py:56-64,124-125: OPEN_CLIP with the causal text
    transformer), loaded from a synthetic checkpoint directory."""
    import json
    import tempfile
    from oracle import open_clip_oracle as OC
    R.install_shims()
    from easynlp.appzoo.clip.model import CLIPApp
    torch.manual_seed(0)
    cfg = OC.OPENCLIP_CONFIGS[cfg_name]
    sd = OC.make_state_dict(cfg, wseed)
    d = tempfile.mkdtemp()
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg, f)
    torch.save({"open_clip." + k: v for k, v in sd.items()}, os.path.join(d, "pytorch_model.bin"))
    app = CLIPApp(d)
    assert app.model_type == "open_clip"
    app.eval()
    px, ids = OC.make_inputs(cfg, B, iseed)
    fo = app({"pixel_values": px, "input_ids": ids})
    loss = app.compute_loss(fo, [])["loss"]
    loss.backward()
    out = {"meta": np.array([cfg_name, str(B), str(cfg["context_length"]), str(wseed), str(iseed), torch.__version__, np.__version__]),
           "image_embeds": fo["image_embeds"].detach().numpy(), "text_embeds": fo["text_embeds"].detach().numpy(),
           "logits_per_text": fo["logits_per_text"].detach().numpy(), "loss": np.float32(loss.item())}
    for n, p in app.named_parameters():
        n = n.replace("open_clip.", "", 1)
        if p.grad is None:
            out["nograd/" + n] = np.zeros(0, np.float32)
        elif full:
            out["grad/" + n] = p.grad.numpy()
        else:
            norm, samp, idx = grad_digest(p.grad)
            out["gnorm/" + n] = np.float64(norm)
            out["gsamp/" + n] = samp
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "loss", loss.item(), "->", path, os.path.getsize(path) // 1024, "KiB")


OPENCLIP_CASES = [("openclip_tiny_b6", "oc_tiny", 6, 1234, 3, True), ("openclip_small_b5", "oc_small", 5, 99, 7, False)]

HF_CASES = [("hf_tiny_b6_l24", "hf_tiny", 6, 24, 1234, 3, True), ("hf_small_b5_l40", "hf_small", 5, 40, 99, 7, False),
            ("hf_large_text_b24_l40", "hf_large_text", 24, 40, 41, 13, False)]   # round 4: the LARGE RoBERTa text tower's widths


def run_wukong_case(name, cfg_name, B, wseed, iseed, full):
    """The REAL reference WukongCLIP (appzoo/wukong_clip/model.py:8-73; WukongModel, LayerNorm eps 1e-7, feature of the
    token 102), loaded from a synthetic checkpoint directory."""
    import json
    import tempfile
    from oracle import wukong_oracle as WK
    R.install_shims()
    from easynlp.appzoo.wukong_clip.model import WukongCLIP
    torch.manual_seed(0)
    cfg = WK.WUKONG_CONFIGS[cfg_name]
    sd = WK.make_state_dict(cfg, wseed)
    d = tempfile.mkdtemp()
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg, f)
    torch.save(sd, os.path.join(d, "pytorch_model.bin"))
    app = WukongCLIP(d)
    app.eval()
    px, ids = WK.make_inputs(cfg, B, iseed)
    fo, _ = app({"pixel_values": px, "input_ids": ids})
    loss = app.compute_loss(fo, [])["loss"]
    loss.backward()
    out = {"meta": np.array([cfg_name, str(B), str(wseed), str(iseed), torch.__version__, np.__version__]),
           "image_features": fo["image_features"].detach().numpy(), "text_features": fo["text_features"].detach().numpy(),
           "logit_scale": fo["logit_scale"].detach().numpy(), "loss": np.float32(loss.item())}
    names = [n for n, _ in app.named_parameters()]
    assert sorted(names) == sorted(sd), (set(names) ^ set(sd))
    for n, p in app.named_parameters():
        if p.grad is None:
            out["nograd/" + n] = np.zeros(0, np.float32)
        elif full:
            out["grad/" + n] = p.grad.numpy()
        else:
            norm, samp, idx = grad_digest(p.grad)
            out["gnorm/" + n] = np.float64(norm)
            out["gsamp/" + n] = samp
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "loss", loss.item(), "->", path, os.path.getsize(path) // 1024, "KiB")


WUKONG_CASES = [("wukong_tiny_b6", "wk_tiny", 6, 1234, 3, True), ("wukong_small_b5", "wk_small", 5, 99, 7, False)]


def run_t2v_case(name, cfg_name, B, T, wseed, iseed, full):
    """The REAL reference Text2VideoRetrieval (appzoo/text2video_retrieval/model.py:39-121): OPEN_CLIP per frame + masked
    mean pooling, loaded from a synthetic open_clip checkpoint directory."""
    import json
    import tempfile
    from oracle import open_clip_oracle as OC
    from oracle import text2video_oracle as TV
    R.install_shims()
    from easynlp.appzoo.text2video_retrieval.model import Text2VideoRetrieval
    torch.manual_seed(0)
    cfg = OC.OPENCLIP_CONFIGS[cfg_name]
    sd = OC.make_state_dict(cfg, wseed)
    d = tempfile.mkdtemp()
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg, f)
    torch.save({"open_clip." + k: v for k, v in sd.items()}, os.path.join(d, "pytorch_model.bin"))
    app = Text2VideoRetrieval(d)
    app.eval()
    px, masks, ids = TV.make_inputs(cfg, B, T, iseed)
    fo = app({"pixel_values": px.clone(), "video_masks": masks.clone(), "input_ids": ids.clone()})
    loss = app.compute_loss(fo, [])["loss"]
    loss.backward()
    out = {"meta": np.array([cfg_name, str(B), str(T), str(wseed), str(iseed), torch.__version__, np.__version__]),
           "video_embeds": fo["video_embeds"].detach().numpy(), "text_embeds": fo["text_embeds"].detach().numpy(),
           "logits_per_text": fo["logits_per_text"].detach().numpy(), "loss": np.float32(loss.item())}
    for n, p in app.named_parameters():
        n = n.replace("open_clip.", "", 1)
        if p.grad is None:
            out["nograd/" + n] = np.zeros(0, np.float32)
        elif full:
            out["grad/" + n] = p.grad.numpy()
        else:
            norm, samp, idx = grad_digest(p.grad)
            out["gnorm/" + n] = np.float64(norm)
            out["gsamp/" + n] = samp
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "loss", loss.item(), "->", path, os.path.getsize(path) // 1024, "KiB")


T2V_CASES = [("t2v_tiny_b4_t3", "oc_tiny", 4, 3, 1234, 3, False), ("t2v_small_b3_t5", "oc_small", 3, 5, 99, 7, False)]


def run_dataset_case(name="dataset_tsv_b7"):
    """The reference's own CLIPDataset (appzoo/clip/data.py:152-295) over a small TSV: text \\t urlsafe-base64(PNG).
    Fixture: the TSV, vocab.txt, the token tensors of batch_fn, and per-image SHA-256 of the float32 pixel_values (the
    tensors themselves would be 4 MB of resampled noise) plus a strided sample for diagnostics."""
    import base64, hashlib, io, tempfile
    import numpy as np
    from PIL import Image
    R.install_shims()
    from easynlp.appzoo.clip.data import CLIPDataset
    rng = np.random.RandomState(77)
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list("abcdefghijklmnopqrstuvwxyz") + \
            ["##" + c for c in "abcdefghijklmnopqrstuvwxyz"] + ["cat", "dog", "photo", "of", "a", "the", "##s", "red", "中", "文", "猫", "狗", "图", ",", "."]
    texts = ["a photo of a cat", "the red dogs.", "中文猫图", "zzz unknownword " * 12, "", "photo, of THE Dog", "狗 a 图"]
    sizes = [(40, 56, "RGB"), (64, 33, "RGB"), (224, 224, "RGB"), (300, 231, "L"), (17, 90, "RGB"), (250, 224, "RGB"), (96, 96, "L")]
    rows = []
    for t, (w, h, mode) in zip(texts, sizes):
        shape = (h, w, 3) if mode == "RGB" else (h, w)
        img = Image.fromarray(rng.randint(0, 256, size=shape).astype(np.uint8), mode)
        buf = io.BytesIO()
        img.save(buf, format="PNG")
        rows.append(t + "\t" + base64.urlsafe_b64encode(buf.getvalue()).decode("ascii"))
    tsv = "\n".join(rows) + "\n"
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "config.json"), "w") as f:
            json.dump({"model_type": "chinese_clip"}, f)
        with open(os.path.join(d, "vocab.txt"), "w") as f:
            f.write("\n".join(vocab) + "\n")
        with open(os.path.join(d, "data.tsv"), "w") as f:
            f.write(tsv)
        ds = CLIPDataset(d, os.path.join(d, "data.tsv"), 20, input_schema="text:str:1,image:str:1",
                         first_sequence="text", second_sequence="image")
        batch = ds.batch_fn([ds[i] for i in range(len(ds))])
    px = batch["pixel_values"].numpy()
    assert px.dtype == np.float32 and px.shape == (len(rows), 3, 224, 224)
    out = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(out, tsv=np.frombuffer(tsv.encode("utf-8"), dtype=np.uint8), vocab=np.frombuffer("\n".join(vocab).encode("utf-8"), dtype=np.uint8),
                        input_ids=batch["input_ids"].numpy(), token_type_ids=batch["token_type_ids"].numpy(),
                        attention_mask=batch["attention_mask"].numpy(),
                        pixel_sha256=np.array([hashlib.sha256(np.ascontiguousarray(px[i]).tobytes()).hexdigest() for i in range(len(rows))]),
                        pixel_sample=px[:, :, ::16, ::16].copy())
    print("wrote", out, os.path.getsize(out))


WUKONG_CORPUS = ["a photo of a cat", "the red dogs.", "中文猫图", "zzz unknownword " * 12, "", "photo, of THE Dog", "狗 a 图",
                 "Café naïve ÀB", "a\tb\nc  d", "[CLS] a [SEP]", "don't stop-me", "ａｂｃ full-width", "a\u200bb\ufffdc", "x" * 150,
                 "y" * 201, "dogs,cats.the;a", "abc中def文", "  leading and trailing  ", "«quoted» „text“ …", "1+1=2 & 50% off!",
                 "\U00020000 extension-B ideograph", "a\u0301 combining", "MiXeD CaSe ÉCOLE", "tab\tsep", "a\x00b\x07c", "　ideographic　space"]


def run_wukong_dataset_case(name="wukong_dataset_b5"):
    """The reference's own WukongCLIPDataset / FullTokenizer (appzoo/wukong_clip/data.py:136-241, bert_tokenizer.py) over
    the images of dataset_tsv_b7 with a BERT-layout vocabulary ([CLS] = 101, [SEP] = 102): token tensors of batch_fn,
    per-image SHA-256 of pixel_values, and the wordpiece ids of a corpus of edge-case strings."""
    import hashlib
    import tempfile
    R.install_shims()
    from easynlp.appzoo.wukong_clip.bert_tokenizer import FullTokenizer
    from easynlp.appzoo.wukong_clip.data import WukongCLIPDataset
    base = np.load(os.path.join(ROOT, "tests", "golden", "dataset_tsv_b7.npz"))
    # the reference's Wukong pipeline has no convert('RGB') (data.py:82-83): greyscale rows raise there -- keep the RGB rows
    import base64, io
    from PIL import Image
    rows = [r for r in base["tsv"].tobytes().decode("utf-8").split("\n") if r]
    rows = [r for r in rows if Image.open(io.BytesIO(base64.urlsafe_b64decode(r.split("\t")[1]))).mode == "RGB"]
    tsv = "\n".join(rows) + "\n"
    words = [w for w in base["vocab"].tobytes().decode("utf-8").split("\n") if not w.startswith("[")]
    vocab = ["[PAD]"] + ["[unused%d]" % i for i in range(1, 100)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + words + \
            ["é", "cafe", "naive", "ab", "+", "=", "&", "%", "!", "1", "2", "50", "off", "mixed", "case", "ecole", "«", "»", "…", "-", "'"]
    assert vocab.index("[SEP]") == 102 and vocab.index("[CLS]") == 101
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "vocab.txt"), "w") as f:
            f.write("\n".join(vocab) + "\n")
        with open(os.path.join(d, "data.tsv"), "w") as f:
            f.write(tsv)
        ds = WukongCLIPDataset(d, os.path.join(d, "data.tsv"), 32, input_schema="text:str:1,image:str:1",
                               first_sequence="text", second_sequence="image")
        batch = ds.batch_fn([ds[i] for i in range(len(ds))])
        tok = FullTokenizer(vocab_file=os.path.join(d, "vocab.txt"))
        corpus_ids = [tok.convert_tokens_to_ids(tok.tokenize(t)) for t in WUKONG_CORPUS]
    px = batch["pixel_values"].numpy()
    assert px.dtype == np.float32 and px.shape[1:] == (3, 224, 224) and sorted(batch) == ["input_ids", "pixel_values"]
    out = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(out, vocab=np.frombuffer("\n".join(vocab).encode("utf-8"), dtype=np.uint8),
                        tsv=np.frombuffer(tsv.encode("utf-8"), dtype=np.uint8), input_ids=batch["input_ids"].numpy(),
                        pixel_sha256=np.array([hashlib.sha256(np.ascontiguousarray(px[i]).tobytes()).hexdigest() for i in range(len(px))]),
                        corpus=np.frombuffer("\x1e".join(WUKONG_CORPUS).encode("utf-8"), dtype=np.uint8),
                        corpus_ids=np.array([",".join(map(str, ids)) for ids in corpus_ids]))
    print("wrote", out, os.path.getsize(out))


BPE_TRAIN_TEXT = ("a photo of a cat . a photo of a dog . the red dogs and the red cats are running in the garden . "
                  "two people riding bikes near the river , it's a sunny day ! she'll say they've done it ; i'm here , we're there . "
                  "photograph photography photographer telephoto 2023 1999 100% caf\u00e9 na\u00efve fa\u00e7ade "
                  "\u4e2d\u6587 \u732b \u56fe \u4e2d\u6587\u732b\u56fe running runner runs ran the then there these those ") * 3

BPE_CORPUS = ["a photo of a cat", "The RED dogs.", "two people riding bikes near the river", "it's a sunny day! she'll say they've done it",
              "photographers' telephoto lens, 2023", "", "   spaced    out \t text \n", "caf\u00e9 na\u00efve fa\u00e7ade", "\u4e2d\u6587\u732b\u56fe",
              "&amp;lt;b&amp;gt; html &quot;entities&quot; &amp;amp;", "<start_of_text> inside <end_of_text> text", "emoji \U0001F600 and symbols #$%^&*()",
              "word " * 60, "x", "'s 't 're don't I'M", "100% of 1,234.56", "under_score-and-dash", "\u00bd \u2460 \u0663"]


def _train_bpe(text, n_merges):
    """a minimal BPE trainer (most frequent adjacent pair, ties by first occurrence) -- only to obtain a realistic merges
    file for the tokenizer fixtures; the tokenizers under test only READ merges"""
    import collections
    import regex
    R.install_shims()
    from easynlp.modelzoo.models.clip.openclip_tokenizer import bytes_to_unicode
    enc = bytes_to_unicode()
    pat = regex.compile(r"""'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""", regex.IGNORECASE)
    words = collections.Counter()
    for tok in pat.findall(text.lower()):
        sym = [enc[b] for b in tok.encode("utf-8")]
        sym[-1] += "</w>"
        words[tuple(sym)] += 1
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for w, c in words.items():
            for a, b in zip(w[:-1], w[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        best = max(pairs.items(), key=lambda kv: kv[1])[0]
        merges.append(best)
        new = collections.Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i + 1 < len(w) and (w[i], w[i + 1]) == best:
                    out.append(w[i] + w[i + 1]); i += 2
                else:
                    out.append(w[i]); i += 1
            new[tuple(out)] += c
        words = new
    return merges


def run_bpe_case(name="openclip_bpe_corpus"):
    """The reference's SimpleTokenizer / openclip_tokenize (modelzoo/models/clip/openclip_tokenizer.py, appzoo/clip/data.py:
    137-161) over a gzip merges file trained here: ids of an edge-case corpus and the [n, 77] / [n, 16] token tensors."""
    import gzip
    import tempfile
    R.install_shims()
    from easynlp.appzoo.clip.data import openclip_tokenize
    from easynlp.modelzoo.models.clip.openclip_tokenizer import SimpleTokenizer
    merges = _train_bpe(BPE_TRAIN_TEXT, 400)
    blob = "#version: synthetic fixture\n" + "\n".join(a + " " + b for a, b in merges) + "\n"
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "vocab.txt")
        with gzip.open(path, "wb") as f:
            f.write(blob.encode("utf-8"))
        tok = SimpleTokenizer(bpe_path=path)
        ids = [tok.encode(t) for t in BPE_CORPUS]
        t77 = openclip_tokenize(BPE_CORPUS, context_length=77, _tokenizer=tok).numpy()
        t16 = openclip_tokenize(BPE_CORPUS, context_length=16, _tokenizer=tok).numpy()
        meta = np.array([str(tok.vocab_size), str(tok.encoder["<start_of_text>"]), str(tok.encoder["<end_of_text>"]), str(len(merges))])
    out = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(out, merges=np.frombuffer(blob.encode("utf-8"), dtype=np.uint8),
                        corpus=np.frombuffer("\x1e".join(BPE_CORPUS).encode("utf-8"), dtype=np.uint8),
                        corpus_ids=np.array([",".join(map(str, x)) for x in ids]), tokens77=t77, tokens16=t16, meta=meta)
    print("wrote", out, os.path.getsize(out), "merges", len(merges))


def run_t2v_dataset_case(name="t2v_dataset_b3"):
    """The reference's own Text2VideoRetrievalDataset (appzoo/text2video_retrieval/data.py:163-279) over three clips stored as
    frame directories (12 / 5 / 1 frames, several sizes, one greyscale frame) with the BPE merges of openclip_bpe_corpus:
    token tensor, video masks, and the SHA-256 of every frame's float32 pixel_values keyed by frame file name (the frame
    ORDER is os.listdir's and belongs to the directory, not to the fixture)."""
    import gzip
    import hashlib
    import io
    import tempfile
    from PIL import Image
    R.install_shims()
    from easynlp.appzoo.text2video_retrieval.data import Text2VideoRetrievalDataset
    bpe = np.load(os.path.join(ROOT, "tests", "golden", "openclip_bpe_corpus.npz"))
    rng = np.random.RandomState(5)
    captions = ["two people riding bikes near the river", "a photo of a cat", "The RED dogs."]
    specs = [[(64, 48, "RGB")] * 12, [(224, 224, "RGB"), (300, 200, "RGB"), (100, 160, "L"), (224, 250, "RGB"), (17, 33, "RGB")], [(90, 90, "RGB")]]
    frames = {}
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "config.json"), "w") as f:
            json.dump({"model_type": "open_clip"}, f)
        with gzip.open(os.path.join(d, "vocab.txt"), "wb") as f:
            f.write(bpe["merges"].tobytes())
        rows = []
        for ci, (cap, spec) in enumerate(zip(captions, specs)):
            cdir = os.path.join(d, "clip%d" % ci)
            os.makedirs(cdir)
            for fi, (w, h, mode) in enumerate(spec):
                arr = rng.randint(0, 256, size=(h, w, 3) if mode == "RGB" else (h, w)).astype(np.uint8)
                buf = io.BytesIO()
                Image.fromarray(arr, mode).save(buf, format="PNG")
                fname = "f%02d.png" % fi
                with open(os.path.join(cdir, fname), "wb") as f:
                    f.write(buf.getvalue())
                frames["clip%d/%s" % (ci, fname)] = np.frombuffer(buf.getvalue(), dtype=np.uint8)
            rows.append(cap + "\t" + cdir)
        with open(os.path.join(d, "data.tsv"), "w") as f:
            f.write("\n".join(rows) + "\n")
        ds = Text2VideoRetrievalDataset(d, os.path.join(d, "data.tsv"), 77, input_schema="text:str:1,image:str:1",
                                        first_sequence="text", second_sequence="image")
        batch = ds.batch_fn([ds[i] for i in range(len(ds))])
        order = [os.listdir(os.path.join(d, "clip%d" % ci)) for ci in range(3)]
    px = batch["pixel_values"].numpy()
    assert px.shape == (3, 12, 3, 224, 224) and px.dtype == np.float32
    sha = {}
    for ci in range(3):
        for fi, fname in enumerate(order[ci]):
            sha["clip%d/%s" % (ci, fname)] = hashlib.sha256(np.ascontiguousarray(px[ci, fi]).tobytes()).hexdigest()
    pad_sha = hashlib.sha256(np.ascontiguousarray(px[2, 11]).tobytes()).hexdigest()
    assert all(hashlib.sha256(np.ascontiguousarray(px[2, k]).tobytes()).hexdigest() == pad_sha for k in range(1, 12))
    out = {"captions": np.array(captions), "input_ids": batch["input_ids"].numpy(), "video_masks": batch["video_masks"].numpy(),
           "frame_names": np.array(sorted(frames)), "frame_sha256": np.array([sha[k] for k in sorted(frames)]), "pad_sha256": np.array(pad_sha)}
    for k in sorted(frames):
        out["png/" + k] = frames[k]
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    only = sys.argv[1:]
    for case in CASES:
        if only and case[0] not in only:
            continue
        run_case(*case)
    if not only or "tiny_dropout_b6_l24" in only:
        run_dropout_case()
    for case in HF_CASES:
        if not only or case[0] in only:
            run_hf_case(*case)
    for case in OPENCLIP_CASES:
        if not only or case[0] in only:
            run_openclip_case(*case)
    if not only or "dataset_tsv_b7" in only:
        run_dataset_case()
    for case in WUKONG_CASES:
        if not only or case[0] in only:
            run_wukong_case(*case)
    if not only or "wukong_dataset_b5" in only:
        run_wukong_dataset_case()
    if not only or "openclip_bpe_corpus" in only:
        run_bpe_case()
    if not only or "t2v_dataset_b3" in only:
        run_t2v_dataset_case()
    for case in T2V_CASES:
        if not only or case[0] in only:
            run_t2v_case(*case)

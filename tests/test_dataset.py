"""Input side of the path (SURVEY.md 8f "next": the caller upstream of CLIPApp.forward) -- the drop-in CLIPDataset against
the reference's own ``easynlp.appzoo.clip.data.CLIPDataset``.

Fixture tests/golden/dataset_tsv_b7.npz (tools/make_golden.py: run_dataset_case) holds a 7-row TSV in the reference's
wire format (text \\t urlsafe-base64(PNG)), the vocab, and what the REFERENCE dataset's batch_fn produced from it:
token tensors verbatim, pixel_values as per-image SHA-256 (+ a strided sample).  CPU tests pin rows / tokens / decode;
the GPU tests pin the float32 pixel_values bit for bit and run them through CLIPApp.forward."""
import base64
import hashlib
import io
import json
import os

import numpy as np
import pytest
import torch

from easynlp_amd import lib as L
from easynlp_amd.appzoo.clip.data import CLIPDataset, parse_row_by_schema
from oracle import ref_harness as R

PIL = pytest.importorskip("PIL.Image")
GOLD = os.path.join(os.path.dirname(__file__), "golden", "dataset_tsv_b7.npz")
SCHEMA = "text:str:1,image:str:1"


def _materialise(tmp_path, model_type="chinese_clip"):
    g = np.load(GOLD)
    d = str(tmp_path)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump({"model_type": model_type} if model_type else {"text_config": {}, "vision_config": {}}, f)
    with open(os.path.join(d, "vocab.txt"), "wb") as f:
        f.write(g["vocab"].tobytes() + b"\n")
    with open(os.path.join(d, "data.tsv"), "wb") as f:
        f.write(g["tsv"].tobytes())
    return g, d


def _dataset(d, **kw):
    return CLIPDataset(d, os.path.join(d, "data.tsv"), 20, input_schema=SCHEMA, first_sequence="text",
                       second_sequence="image", **kw)


def test_parse_row_by_schema_follows_the_reference():
    assert parse_row_by_schema("hello\tQUJD\n", SCHEMA) == {"text": "hello", "image": "QUJD"}
    # columns are zipped with the schema: surplus columns are dropped, missing ones simply absent
    assert parse_row_by_schema("a\tb\tc", SCHEMA) == {"text": "a", "image": "b"}
    assert parse_row_by_schema("a", SCHEMA) == {"text": "a"}
    assert parse_row_by_schema("3\t1,2,3\t0.5", "n:int:1,v:int:3,x:float:1") == {"n": 3, "v": [1, 2, 3], "x": 0.5}
    with pytest.raises(RuntimeError):
        parse_row_by_schema("a", "text:bytes:1")


def test_tokens_equal_the_reference_batch(tmp_path):
    g, d = _materialise(tmp_path)
    ds = _dataset(d)
    assert len(ds) == 7
    batch = ds.batch_fn([ds[i] for i in range(len(ds))])
    for k in ("input_ids", "token_type_ids", "attention_mask"):
        assert batch[k].dtype == torch.int64
        assert np.array_equal(batch[k].numpy(), g[k]), k
    assert batch["label_ids"] == [] and "pixel_values" not in batch
    assert ds.label_enumerate_values == ["0", "1"]              # Trainer.save_checkpoint reads it (trainer.py:429)
    # rows: truncated to max_seq_length (row 3), empty text = [CLS][SEP] (row 4)
    assert int(batch["attention_mask"][3].sum()) == 20 and int(batch["attention_mask"][4].sum()) == 2


def test_images_are_the_decoded_pixels(tmp_path):
    g, d = _materialise(tmp_path)
    ds = _dataset(d)
    rows = g["tsv"].tobytes().decode("utf-8").split("\n")[:-1]
    batch = ds.batch_fn([ds[i] for i in range(len(ds))])
    assert batch["image_size"] == 224 and len(batch["images"]) == 7
    for row, img in zip(rows, batch["images"]):
        ref = PIL.open(io.BytesIO(base64.urlsafe_b64decode(row.split("\t")[1])))
        assert img.dtype == np.uint8 and np.array_equal(img, np.asarray(ref))
    assert batch["images"][3].ndim == 2 and batch["images"][0].shape == (56, 40, 3)      # 'L' stays one channel


def test_dataloader_with_workers_collates_through_batch_fn(tmp_path):
    g, d = _materialise(tmp_path)
    ds = _dataset(d)
    dl = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, collate_fn=ds.batch_fn, num_workers=2)
    batches = list(dl)
    assert [len(b["images"]) for b in batches] == [4, 3]
    assert np.array_equal(torch.cat([b["input_ids"] for b in batches]).numpy(), g["input_ids"])


def test_pack_batches_moves_the_image_copies_into_the_workers(tmp_path):
    """pack_batches=True: batch_fn (run by the DataLoader workers) returns ONE uint8 tensor + descriptors instead of a list
    of arrays -- the same bytes L.pack_images makes of the list, through worker IPC as well"""
    g, d = _materialise(tmp_path)
    plain, packed = _dataset(d), _dataset(d, pack_batches=True)
    want = L.pack_images(plain.batch_fn([plain[i] for i in range(7)])["images"])
    dl = torch.utils.data.DataLoader(packed, batch_size=7, shuffle=False, collate_fn=packed.batch_fn, num_workers=2)
    batch = next(iter(dl))
    assert L.is_packed_images(batch["images"]) and batch["image_size"] == 224
    assert torch.equal(batch["images"]["desc"], want["desc"])
    for o, w, h in want["desc"].tolist():
        assert torch.equal(batch["images"]["data"][o:o + w * h * 3], want["data"][o:o + w * h * 3])
    assert np.array_equal(batch["input_ids"].numpy(), g["input_ids"])


def test_huggingface_flavour_and_what_is_not_covered(tmp_path):
    g, d = _materialise(tmp_path, model_type=None)
    ds = _dataset(d)
    assert ds.model_type == "huggingface_clip"
    b = ds.batch_fn([ds[0], ds[1]])
    assert np.array_equal(b["input_ids"].numpy(), g["input_ids"][:2]) and "token_type_ids" in b and "attention_mask" in b
    with pytest.raises(FileNotFoundError):              # a tar shard that does not exist
        CLIPDataset(d, os.path.join(d, "shard-000.tar"), 20, input_schema=SCHEMA, first_sequence="text", second_sequence="image")
    # palette / alpha / CMYK images take the reference's own CPU steps for what depends on the mode (resize IN that mode,
    # crop, convert('RGB')) and reach the GPU as 224 x 224 RGB: /255 and the normalisation of exactly those bytes are the
    # reference's pixel_values (checked against the reference pipeline itself when the checkout is present)
    rng = np.random.RandomState(5)
    rows = []
    for mode, shape in (("P", (300, 417)), ("RGBA", (260, 231, 4)), ("CMYK", (224, 500, 4)), ("LA", (231, 260, 2))):
        im = PIL.fromarray(rng.randint(0, 255, shape, dtype=np.uint8), mode)
        if mode == "P":
            im.putpalette([int(v) for v in rng.randint(0, 255, 768)])
        buf = io.BytesIO()
        im.save(buf, format="TIFF" if mode == "CMYK" else "PNG")
        rows.append((im, base64.urlsafe_b64encode(buf.getvalue()).decode()))
    with open(os.path.join(d, "p.tsv"), "w") as f:
        for i, (_, b64) in enumerate(rows):
            f.write("caption %d\t%s\n" % (i, b64))
    dp = CLIPDataset(d, os.path.join(d, "p.tsv"), 20, input_schema=SCHEMA, first_sequence="text", second_sequence="image")
    with pytest.warns(UserWarning, match="resize / crop run on the CPU"):
        got = [dp[i]["image"] for i in range(len(rows))]
    assert all(a.shape == (224, 224, 3) and a.dtype == np.uint8 for a in got)
    if R.reference_available():
        R.install_shims()
        from easynlp.appzoo.clip import data as RD
        mean, std = np.array(L.CLIP_MEAN, dtype=np.float32), np.array(L.CLIP_STD, dtype=np.float32)
        for (im, b64), a in zip(rows, got):
            src = PIL.open(io.BytesIO(base64.urlsafe_b64decode(b64)))
            ref = RD._normalize(RD._center_crop(RD._resize(src, 224), 224))              # data.py:256-262
            mine = ((a.astype(np.float32) / 255.0).transpose(2, 0, 1) - mean[:, None, None]) / std[:, None, None]
            assert np.array_equal(mine.astype(np.float32), ref.astype(np.float32)), src.mode
    # a corrupt row surfaces as the reference's RuntimeError (dataset.py:196-201)
    with open(os.path.join(d, "bad.tsv"), "w") as f:
        f.write("a\tnot-an-image\n")
    db = CLIPDataset(d, os.path.join(d, "bad.tsv"), 20, input_schema=SCHEMA, first_sequence="text", second_sequence="image")
    with pytest.raises(RuntimeError):
        db[0]


def test_open_clip_flavour_tokenises_with_the_bpe_merges_file(tmp_path):
    """model_type open_clip (data.py:225-227,246-249): vocab.txt is the gzip merges file, captions become 77 BPE ids; the
    ids of the fixture captions are the reference tokenizer's (tests/golden/openclip_bpe_corpus.npz)."""
    import gzip
    g, d = _materialise(tmp_path, model_type="open_clip")
    b = np.load(os.path.join(os.path.dirname(GOLD), "openclip_bpe_corpus.npz"))
    with gzip.open(os.path.join(d, "vocab.txt"), "wb") as f:
        f.write(b["merges"].tobytes())
    corpus = b["corpus"].tobytes().decode("utf-8").split("\x1e")
    rows = g["tsv"].tobytes().decode("utf-8").split("\n")[:-1]
    keep = [i for i, t in enumerate(corpus) if "\t" not in t and "\n" not in t][:len(rows)]
    with open(os.path.join(d, "oc.tsv"), "w") as f:
        for i, r in zip(keep, rows):
            f.write(corpus[i] + "\t" + r.split("\t")[1] + "\n")
    ds = CLIPDataset(d, os.path.join(d, "oc.tsv"), 32, input_schema=SCHEMA, first_sequence="text", second_sequence="image")
    assert ds.model_type == "open_clip"
    batch = ds.batch_fn([ds[i] for i in range(len(ds))])
    assert tuple(batch["input_ids"].shape) == (len(rows), 77) and "attention_mask" not in batch and "token_type_ids" not in batch
    # a TSV row is stripped of its newline only, the tokenizer cleans the rest: same ids as the reference's tokens77 rows
    assert np.array_equal(batch["input_ids"].numpy(), b["tokens77"][keep])


@pytest.mark.gpu
def test_gpu_pixel_values_equal_the_reference_dataset_bit_for_bit(tmp_path):
    g, d = _materialise(tmp_path)
    ds = _dataset(d)
    batch = ds.batch_fn([ds[i] for i in range(len(ds))])
    px = L.preprocess_images(batch["images"], size=224, crop=224).cpu().numpy()
    assert px.shape == (7, 3, 224, 224) and px.dtype == np.float32
    assert np.array_equal(px[:, :, ::16, ::16].view(np.uint32), g["pixel_sample"].view(np.uint32))
    for i in range(7):
        assert hashlib.sha256(np.ascontiguousarray(px[i]).tobytes()).hexdigest() == str(g["pixel_sha256"][i]), i


@pytest.mark.gpu
def test_clipapp_forward_takes_dataset_batches(tmp_path):
    """DataLoader(CLIPDataset, collate_fn=batch_fn) -> CLIPApp.forward -> compute_loss, as Trainer drives it
    (core/trainer.py); 'images' and the equivalent 'pixel_values' give identical outputs."""
    from easynlp_amd.appzoo.clip import CLIPApp
    from oracle import clip_oracle as O
    from oracle import ref_harness as R
    g = np.load(GOLD)
    cfg = dict(O.CONFIGS["tiny"])
    vocab = g["vocab"].tobytes().decode("utf-8").split("\n")
    cfg["vocab_size"] = len(vocab)
    R.write_checkpoint_dir(str(tmp_path), cfg, O.make_state_dict(cfg, 5))
    with open(os.path.join(str(tmp_path), "vocab.txt"), "wb") as f:
        f.write(g["vocab"].tobytes() + b"\n")
    with open(os.path.join(str(tmp_path), "data.tsv"), "wb") as f:
        f.write(g["tsv"].tobytes())
    res = int(cfg["image_resolution"])
    ds = _dataset(str(tmp_path), image_size=res)
    app = CLIPApp(str(tmp_path)).cuda()
    dl = torch.utils.data.DataLoader(ds, batch_size=7, shuffle=False, collate_fn=ds.batch_fn)
    batch = next(iter(dl))
    images = list(batch["images"])
    out = app(batch)
    assert tuple(out["logits_per_text"].shape) == (7, 7) and tuple(batch["pixel_values"].shape) == (7, 3, res, res)
    loss = app.compute_loss(out, batch["label_ids"])["loss"]
    loss.backward()
    # (every parameter but the BertPooler, which chinese_clip computes and never uses: .grad stays None as in the reference)
    assert torch.isfinite(loss) and all(p.grad is not None for n, p in app.named_parameters() if p.requires_grad and "pooler" not in n)
    px = L.preprocess_images(images, size=res, crop=res)
    app.eval()
    with torch.no_grad():
        a = app({"images": images, "image_size": res, "input_ids": torch.from_numpy(g["input_ids"])})
        b = app({"pixel_values": px, "input_ids": torch.from_numpy(g["input_ids"])})
    assert torch.equal(a["logits_per_text"], b["logits_per_text"]) and torch.equal(a["image_embeds"], b["image_embeds"])


def test_parse_row_by_schema_fuzz_against_the_reference():
    from oracle import ref_harness as R
    if not R.reference_available():
        pytest.skip("reference checkout not present")
    R.install_shims()
    import random
    from easynlp.utils import parse_row_by_schema as ref_parse
    rnd = random.Random(4)
    for _ in range(500):
        ncol = rnd.randint(1, 4)
        schema, fields = [], []
        for c in range(ncol):
            typ, length = rnd.choice(["str", "int", "float"]), rnd.choice([1, 1, 3])
            schema.append("c%d:%s:%d" % (c, typ, length))
            if typ == "str":
                fields.append("".join(rnd.choice("ab ,:中x") for _ in range(rnd.randint(0, 6))))
            else:
                vals = [str(rnd.randint(-9, 9)) if typ == "int" else "%.3f" % rnd.uniform(-2, 2) for _ in range(length)]
                fields.append(",".join(vals))
        extra = rnd.choice([0, 0, 1, -1])                       # a surplus column, or one column short
        row_fields = fields + ["zz"] if extra == 1 else (fields[:-1] if extra == -1 and ncol > 1 else fields)
        row = "\t".join(row_fields) + rnd.choice(["", "\n"])
        assert parse_row_by_schema(row, ",".join(schema)) == ref_parse(row, ",".join(schema)), (row, schema)


def test_wordpiece_loader_fuzz_against_the_reference_tokenizer(tmp_path):
    """load_wordpiece_tokenizer (the installed transformers' BertTokenizer, 4.x or 5.x) vs the reference's vendored
    BertTokenizer.from_pretrained(vocab.txt) (data.py:229) with the dataset's call (padding / truncation / max_length):
    control characters, CJK, accents, full-width forms, 100-character words, special tokens inside the text."""
    from oracle import ref_harness as R
    if not R.reference_available():
        pytest.skip("reference checkout not present")
    R.install_shims()
    import random
    from easynlp.modelzoo import BertTokenizer as RefTok
    from easynlp_amd.appzoo.clip.data import load_wordpiece_tokenizer
    g = np.load(os.path.join(os.path.dirname(GOLD), "wukong_dataset_b5.npz"))          # BERT-layout vocabulary
    vp = os.path.join(str(tmp_path), "vocab.txt")
    with open(vp, "wb") as f:
        f.write(g["vocab"].tobytes() + b"\n")
    ref, mine = RefTok.from_pretrained(vp), load_wordpiece_tokenizer(vp)
    rnd = random.Random(11)
    alphabet = list("abcdefghijklmnopqrstuvwxyzABCXYZ  \t,.;:!?'\"-_()[]{}<>@#$%^&*+=~`|\\/0123456789") + \
        list("中文猫狗图的了，。！？「」·—…éÀüñçøßÆ") + ["​", "�", "\x07", " ", "　", "́", "\U0001F600", "\U00020000", "ａ", "①", "ｶ", "ﾞ", "ก", "ั"]
    for it in range(1500):
        t = "".join(rnd.choice(alphabet) for _ in range(rnd.randint(0, 40)))
        if it % 40 == 0:
            t += "x" * rnd.choice([99, 100, 101, 150])
        if it % 55 == 0:
            t += " [SEP] [CLS] [MASK] [UNK] [PAD]"
        a = ref([t], padding="max_length", truncation=True, max_length=24, return_tensors="pt")
        b = mine([t], padding="max_length", truncation=True, max_length=24, return_tensors="pt")
        for k in ("input_ids", "token_type_ids", "attention_mask"):
            assert a[k].tolist() == b[k].tolist(), (k, t)


def test_webdataset_tar_input(tmp_path):
    """data_file ending in 'tar' (data.py:203-217): samples = members sharing a key, image from jpg / png, caption from the
    json; brace-expanded shard lists dealt to ranks as urls[rank::world].  webdataset itself is not installed, so this branch
    is checked against its documented conventions only (PARITY UNPINNED)."""
    import tarfile
    from easynlp_amd.appzoo.clip.data import expand_braces, read_webdataset_tar
    g, d = _materialise(tmp_path)
    rows = [r.split("\t") for r in g["tsv"].tobytes().decode("utf-8").split("\n")[:-1]]

    def add(tf, name, payload):
        info = tarfile.TarInfo(name)
        info.size = len(payload)
        tf.addfile(info, io.BytesIO(payload))
    shards = [os.path.join(d, "shard-%03d.tar" % i) for i in range(2)]
    for si, path in enumerate(shards):
        with tarfile.open(path, "w") as tf:
            for ri, (text, b64) in enumerate(rows):
                if ri % 2 != si:
                    continue
                key = "pairs/%05d" % ri
                add(tf, key + ".json", json.dumps({"caption": text, "id": ri}).encode("utf-8"))
                add(tf, key + ".png", base64.urlsafe_b64decode(b64))
    assert expand_braces(os.path.join(d, "shard-{000..001}.tar")) == shards
    ds = CLIPDataset(d, os.path.join(d, "shard-{000..001}.tar"), 20, first_sequence="text", second_sequence="image")
    assert ds.data_source == "tar" and len(ds) == 7
    order = [0, 2, 4, 6, 1, 3, 5]                                  # shard 0 then shard 1
    batch = ds.batch_fn([ds[i] for i in range(7)])
    assert np.array_equal(batch["input_ids"].numpy(), g["input_ids"][order])
    for k, ri in enumerate(order):
        ref = PIL.open(io.BytesIO(base64.urlsafe_b64decode(rows[ri][1]))).convert("RGB")       # decode("pil") -> RGB
        assert np.array_equal(batch["images"][k], np.asarray(ref))
    got = [r["text"] for r in read_webdataset_tar(os.path.join(d, "shard-{000..001}.tar"), rank=1, world=2)]
    assert got == [rows[i][0] for i in (1, 3, 5)]
    with tarfile.open(os.path.join(d, "broken.tar"), "w") as tf:
        add(tf, "a.json", b'{"caption": "x"}')
    with pytest.raises(L.EzclipError):
        CLIPDataset(d, os.path.join(d, "broken.tar"), 20, first_sequence="text", second_sequence="image")

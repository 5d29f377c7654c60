"""Data side of the path -- mirror of ``easynlp.appzoo.clip.data.CLIPDataset`` (easynlp/appzoo/clip/data.py:152-295)
for the local TSV format ``text \\t urlsafe-base64(encoded image)`` the tutorials use, and for webdataset tar shards
(``read_webdataset_tar``) (``input_schema`` such as
``"text:str:1,image:str:1"``; base class contract easynlp/appzoo/dataset.py:39-215: ``__getitem__`` parses the row by the
schema and calls ``convert_single_row_to_example``; the DataLoader collates with ``batch_fn``).

What stays where the reference has it (CPU, DataLoader workers): reading the file, WordPiece tokenisation with the
checkpoint's ``vocab.txt`` (padding='max_length', truncation, max_length=max_seq_length, data.py:250-253) -- or, for
open_clip checkpoints, the byte-level BPE of bpe_tokenizer.py over the gzip merges file (77 ids, data.py:246-249) --
base64 + image decode (data.py:240-244).

What moves to the GPU: the per-image ``_resize`` / ``_center_crop`` / ``_normalize`` (data.py:256-262).  ``batch_fn`` emits
the decoded images under ``'images'`` (plus ``'image_size'``) instead of ``'pixel_values'`` and the drop-in
``CLIPApp.forward`` turns them into the same float32 ``pixel_values`` with ``ezclip_preprocess_images`` (bit-identical to the
reference's PIL pipeline) -- no GPU call happens in a DataLoader worker.  There is no CPU resize here: a
caller who wants CPU-side ``pixel_values`` uses the reference's own CLIPDataset, whose batches ``CLIPApp.forward`` takes
unchanged.
"""
from __future__ import annotations

import base64
import io
import json
import os
from typing import Dict, List

import numpy as np
import torch

from ... import lib as L


def parse_row_by_schema(row: str, input_schema: str) -> Dict[str, object]:
    """``"text:str:1,image:str:1"`` + one tab-separated line -> {column: value} (easynlp/utils/__init__.py:77-98: columns are
    zipped with the schema, so surplus columns are ignored; int / float columns of length > 1 are comma lists)."""
    out: Dict[str, object] = {}
    for schema, content in zip(input_schema.split(","), row.strip("\n").split("\t")):
        name, typ, length = schema.split(":")
        if typ == "str":
            out[name] = content
        elif typ in ("int", "float"):
            conv = int if typ == "int" else float
            out[name] = conv(content) if int(length) == 1 else [conv(t) for t in content.split(",")]
        else:
            raise RuntimeError("Invalid schema: %s" % schema)
    return out


def expand_braces(pattern: str) -> List[str]:
    """``"shard-{000..012}.tar"`` / ``"{a,b}.tar"`` -> file names (the shell-style brace notation webdataset accepts for
    its ``urls``; numeric ranges keep their zero padding)"""
    i = pattern.find("{")
    if i < 0:
        return [pattern]
    j = pattern.find("}", i)
    if j < 0:
        return [pattern]
    head, body, tail = pattern[:i], pattern[i + 1:j], pattern[j + 1:]
    if ".." in body:
        lo, hi = body.split("..", 1)
        width = len(lo) if lo.startswith("0") or hi.startswith("0") else 0
        alts = [str(v).zfill(width) for v in range(int(lo), int(hi) + 1)]
    else:
        alts = body.split(",")
    return [name for a in alts for name in expand_braces(head + a + tail)]


def read_webdataset_tar(urls: str, rank: int = 0, world: int = 1) -> List[Dict[str, object]]:
    """The tar branch of the reference dataset (data.py:203-217: ``wds.WebDataset(data_file, nodesplitter=...)
    .decode("pil").to_tuple("jpg;png", "json")``, every sample kept in memory as ``{'text': json['caption'], 'image': PIL}``)
    restated with ``tarfile`` (webdataset is not part of this image, so this branch cannot be compared with the reference
    here: PARITY UNPINNED for it).  webdataset's conventions: consecutive members sharing the path up to the first dot of
    the base name form one sample, the rest of the name is the field; ``jpg;png`` takes the first of the two present;
    ``decode("pil")`` yields an RGB image; shards are dealt to ranks as ``urls[rank::world]`` (the reference's nodesplitter)."""
    import tarfile
    from PIL import Image
    rows: List[Dict[str, object]] = []
    for path in expand_braces(urls)[rank::world]:
        with tarfile.open(path, "r:*") as tf:
            key, fields = None, {}

            def flush():
                if key is None:
                    return
                img = next((fields[e] for e in ("jpg", "png") if e in fields), None)
                if img is None or "json" not in fields:       # to_tuple("jpg;png", "json") fails on an incomplete sample too
                    raise L.EzclipError("%s: sample %r lacks %s" % (path, key, "an image (jpg / png)" if img is None else "its json"))
                meta = json.loads(fields["json"].decode("utf-8"))
                rows.append({"text": meta["caption"], "image": Image.open(io.BytesIO(img)).convert("RGB")})
            for m in tf:
                if not m.isfile():
                    continue
                dirname, base = os.path.split(m.name)
                stem, _, ext = base.partition(".")
                k = os.path.join(dirname, stem)
                if k != key:
                    flush()
                    key, fields = k, {}
                fields[ext.lower()] = tf.extractfile(m).read()
            flush()
    return rows


def load_wordpiece_tokenizer(vocab_path: str):
    """``BertTokenizer.from_pretrained(dir + '/vocab.txt')`` of the reference (data.py:229, predictor.py:52) with the installed
    ``transformers``: 4.x takes ``vocab_file=``, 5.x takes the token -> id table and silently ignores ``vocab_file`` (every
    token would come out as [UNK]) -- so build by signature and verify the table landed."""
    import inspect
    from transformers import BertTokenizer
    with io.open(vocab_path, encoding="utf-8") as f:
        tokens = [t.rstrip("\n") for t in f.readlines()]
    while tokens and tokens[-1] == "":
        tokens.pop()
    table = {t: i for i, t in enumerate(tokens)}
    if "vocab" in inspect.signature(BertTokenizer.__init__).parameters:
        tok = BertTokenizer(vocab=table)
    else:
        tok = BertTokenizer(vocab_file=vocab_path)
    probe = tokens[len(tokens) // 2]
    if tok.convert_tokens_to_ids(probe) != table[probe] or tok.convert_tokens_to_ids("[CLS]") != table.get("[CLS]"):
        raise L.EzclipError("the installed transformers BertTokenizer did not take the vocabulary of %s" % vocab_path)
    return tok


_warned_modes = set()


def decoded_pixels(image, size: int, crop: int) -> "np.ndarray":
    """Decoded PIL image -> the uint8 array that travels to ``ezclip_preprocess_images``.

    RGB and greyscale images go to the GPU as they are (its resampler is Pillow's 8-bit path, bit for bit).  Every other
    mode -- palette and alpha PNGs, CMYK JPEGs, 16-bit / float images, GIF frames, all common in scraped image-text data --
    takes the reference's own CPU path for the steps that depend on the mode: ``_resize`` (PIL BICUBIC on the shorter
    side, IN THE IMAGE'S OWN MODE: nearest for 'P', premultiplied for alpha, 4 channels for CMYK), ``_center_crop``, and
    the ``convert('RGB')`` that ``_normalize`` starts with (appzoo/clip/data.py:29-72,113-115).  What remains for the
    GPU is /255 and the normalisation: the resampler sees an image that already has the target size and passes it through
    unchanged (unit windows), so ``pixel_values`` equal the reference's for these images too."""
    if image.mode in ("RGB", "L"):
        return np.asarray(image)
    if image.mode not in _warned_modes:
        _warned_modes.add(image.mode)
        import warnings
        warnings.warn("image mode %r: resize / crop run on the CPU in that mode (as the reference does) before the GPU "
                      "pre-processing; RGB / L images are resized on the GPU" % image.mode)
    from PIL import Image
    width, height = image.size
    short, long = (width, height) if width <= height else (height, width)
    if short != size:                                                              # _resize, data.py:52-72
        new_short, new_long = size, int(size * long / short)
        new_w, new_h = (new_short, new_long) if width <= height else (new_long, new_short)
        image = image.resize((new_w, new_h), Image.BICUBIC)
    width, height = image.size                                                     # _center_crop, data.py:29-50
    top, left = int((height - crop + 1) * 0.5), int((width - crop + 1) * 0.5)
    image = image.crop((left, top, left + crop, top + crop))
    return np.asarray(image.convert("RGB"))                                        # _normalize's first step, data.py:113-115


class CLIPDataset(torch.utils.data.Dataset):

    def __init__(self, pretrained_model_name_or_path, data_file, max_seq_length, input_schema=None, first_sequence=None,
                 label_name=None, second_sequence=None, label_enumerate_values=None, user_defined_parameters=None,
                 skip_first_line: bool = False, image_size: int = 224, pack_batches: bool = False, *args, **kwargs):
        path = pretrained_model_name_or_path
        with open(os.path.join(path, "config.json"), "r") as f:
            self.raw_config = json.load(f)
        mt = self.raw_config.get("model_type")
        self.model_type = mt if mt in ("open_clip", "chinese_clip") else "huggingface_clip"          # data.py:196-201
        self.data_source = "tar" if str(data_file)[-3:] == "tar" else "local"
        if self.data_source == "tar":                                                         # data.py:203-217
            import torch.distributed as dist
            on = dist.is_available() and dist.is_initialized()
            self.input_schema, self.column_names = "tar", []
            self.data_rows = read_webdataset_tar(str(data_file), dist.get_rank() if on else 0, dist.get_world_size() if on else 1)
        else:
            if not input_schema:
                raise L.EzclipError("CLIPDataset needs input_schema, e.g. 'text:str:1,image:str:1'")
            self.input_schema = input_schema
            self.column_names = [t.split(":")[0] for t in input_schema.split(",")]
            with io.open(data_file) as f:
                if skip_first_line:
                    f.readline()
                self.data_rows = f.readlines()
        self.text_col = first_sequence
        self.image_col = second_sequence
        if self.model_type == "open_clip":                                                         # data.py:225-229
            from .bpe_tokenizer import SimpleTokenizer
            self.openclip_tokenizer = SimpleTokenizer(bpe_path=os.path.join(path, "vocab.txt"))   # (a gzip merges file)
        else:
            self.tokenizer = load_wordpiece_tokenizer(os.path.join(path, "vocab.txt"))
        self.max_text_length = max_seq_length
        self.pack_batches = bool(pack_batches)      # batch_fn packs the images into one uint8 tensor (in the DataLoader worker)
        self.size = self.crop_size = int(image_size)             # data.py:231-236 fixes 224; other resolutions by keyword

    def __len__(self):
        return len(self.data_rows)

    @property
    def label_enumerate_values(self):
        """read by Trainer.save_checkpoint (core/trainer.py:429-438); BaseDataset's default (appzoo/dataset.py:261-263)"""
        return ["0", "1"]

    def __getitem__(self, item):
        if self.data_source == "tar":
            row = self.data_rows[item]                                                           # dataset.py:191-192
        else:
            row = parse_row_by_schema(self.data_rows[item].strip("\n"), self.input_schema)      # dataset.py:160-196
        try:
            return self.convert_single_row_to_example(row)
        except L.EzclipError:
            raise
        except Exception as e:
            raise RuntimeError("Failed row %d: %s" % (item, e)) from e

    def convert_single_row_to_example(self, row: Dict[str, str]):
        from PIL import Image
        if self.data_source == "tar":                                                            # data.py:236-238
            text, image = row["text"], row["image"]
        else:
            text = row[self.text_col]
            image = Image.open(io.BytesIO(base64.urlsafe_b64decode(row[self.image_col])))         # data.py:242
        if self.model_type == "open_clip":                                                         # data.py:246-249: always 77
            from .bpe_tokenizer import openclip_tokenize
            tk = {"input_ids": openclip_tokenize([text], context_length=77, _tokenizer=self.openclip_tokenizer)}
        else:
            tk = self.tokenizer([text], padding="max_length", truncation=True, max_length=self.max_text_length,
                                return_tensors="pt")                                                # data.py:250-253
        return {"text": tk, "image": decoded_pixels(image, self.size, self.crop_size)}

    def batch_fn(self, features):
        """data.py:275-295; the decoded images travel as ``'images'`` (list of uint8 HWC / HW arrays)."""
        out = {"input_ids": torch.cat([f["text"]["input_ids"] for f in features], dim=0), "label_ids": []}
        for k in ("token_type_ids", "attention_mask"):
            if all(k in f["text"] for f in features):
                out[k] = torch.cat([f["text"][k] for f in features], dim=0)
        out["images"] = [f["image"] for f in features]
        if self.pack_batches:
            out["images"] = L.pack_images(out["images"])
        out["image_size"] = self.size
        return out


class DevicePrefetcher:
    """Wraps a DataLoader of dict batches and hands out batches whose tensors are ALREADY on the device: batch k + 1 crosses PCIe on a
    copy stream while the consumer computes batch k.  The reference's loop gives ``CLIPApp.forward`` host tensors and the copy
    (``inputs['pixel_values'].to(device)``, appzoo/clip/model.py:116-123: 617 MB of float32 pixels per 1 024 pairs) is serial with
    the step: 57.3 instead of 42.6 ms per forward step on one MI355X; behind this wrapper 43.9 (PCIe-inclusive rate).
    It goes where the reference puts its own device loader (``pl.MpDeviceLoader(self._train_loader, self._device)``,
    core/trainer.py:215-218):  ``loader = DevicePrefetcher(loader, device)``.

    A background thread pulls from the wrapped loader and issues the copies (``non_blocking`` from pinned memory; from pageable
    memory the runtime's staged copy blocks only that thread), records an event per batch on the copy stream, and queues at most
    ``depth`` batches; ``__next__`` makes the consumer's current stream wait for the batch's event and registers the tensors with it
    (``record_stream``: their memory is not recycled under the consumer's kernels).  Everything that is not a tensor -- ``label_ids``
    lists, image lists, sizes -- passes through untouched, as do tensors already on the device; of a packed image batch
    (``batch_fn`` with ``pack_batches=True``) the byte buffer is copied and the descriptor table stays on the host.  On a CPU device it is the identity.  An exception raised by the wrapped loader is re-raised in the consumer."""

    def __init__(self, loader, device, depth: int = 2):
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, int(depth))

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch, stream):
        if not isinstance(batch, dict):
            return batch, None
        out = {}
        with torch.cuda.stream(stream):
            for k, v in batch.items():
                if torch.is_tensor(v) and not v.is_cuda:
                    v = v.to(self.device, non_blocking=True)
                elif L.is_packed_images(v) and not v["data"].is_cuda:      # batch_fn(pack_batches=True): the decoded RGB8 bytes
                    v = dict(v, data=v["data"].to(self.device, non_blocking=True))
                out[k] = v
            ev = torch.cuda.Event()
            ev.record(stream)
        return out, ev

    def __iter__(self):
        if self.device.type != "cuda":
            yield from self.loader
            return
        import queue
        import threading
        q = queue.Queue(maxsize=self.depth)
        stream = torch.cuda.Stream(device=self.device)
        stop = threading.Event()
        END = object()

        def put(item):                      # (gives up when the consumer went away: a generator closed mid-epoch)
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def work():
            try:
                torch.cuda.set_device(self.device)
                for batch in self.loader:
                    if not put(self._stage(batch, stream)):
                        return
                put((END, None))
            except BaseException as e:      # noqa: BLE001  (handed to the consumer)
                put((e, "error"))

        t = threading.Thread(target=work, name="ezclip-device-prefetch", daemon=True)
        t.start()
        try:
            while True:
                batch, ev = q.get()
                if batch is END:
                    return
                if isinstance(ev, str):
                    raise batch
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    for v in batch.values():
                        v = v["data"] if L.is_packed_images(v) else v
                        if torch.is_tensor(v) and v.is_cuda:
                            v.record_stream(cur)
                yield batch
        finally:
            stop.set()

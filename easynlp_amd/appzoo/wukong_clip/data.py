"""Mirror of ``easynlp.appzoo.wukong_clip.data.WukongCLIPDataset`` (wukong_clip/data.py:136-241; base contract
appzoo/dataset.py:39-215): TSV rows ``text \\t urlsafe-base64(image)`` parsed by ``input_schema``; the caption becomes
``[CLS] + wordpieces[:30] + [SEP]`` zero-padded to 32 ids (``tokenize`` :181-203 -- always 32, whatever ``max_seq_length``
says, as in the reference); ``batch_fn`` collates ``input_ids`` and the image side.

As in the clip mirror (appzoo/clip/data.py) the decode stays on the CPU and the per-image ``_resize`` / ``_center_crop`` /
``_normalize`` (:222-228) move to the GPU: the batch carries the decoded images under ``'images'`` and
``WukongCLIP.forward`` produces the bit-identical float32 ``pixel_values`` with ``ezclip_preprocess_images``.
"""
from __future__ import annotations

import base64
import io
import os

import numpy as np
import torch

from ... import lib as L
from ..clip.data import parse_row_by_schema
from .tokenizer import FullTokenizer


class WukongCLIPDataset(torch.utils.data.Dataset):

    def __init__(self, pretrained_model_name_or_path, data_file, max_seq_length=32, input_schema=None, first_sequence=None,
                 label_name=None, second_sequence=None, label_enumerate_values=None, user_defined_parameters=None,
                 skip_first_line: bool = False, image_size: int = 224, pack_batches: bool = False, *args, **kwargs):
        if not input_schema:
            raise L.EzclipError("WukongCLIPDataset needs input_schema, e.g. 'text:str:1,image:str:1'")
        self.input_schema = input_schema
        self.column_names = [t.split(":")[0] for t in input_schema.split(",")]
        with io.open(data_file) as f:
            if skip_first_line:
                f.readline()
            self.data_rows = f.readlines()
        self.text_col = first_sequence
        self.image_col = second_sequence
        self.tokenizer = FullTokenizer(vocab_file=os.path.join(pretrained_model_name_or_path, "vocab.txt"))    # :171
        self.max_text_length = max_seq_length
        self.pack_batches = bool(pack_batches)      # batch_fn packs the images into one uint8 tensor (in the DataLoader worker)
        self.size = self.crop_size = int(image_size)                                                          # :174-178: 224

    def __len__(self):
        return len(self.data_rows)

    @property
    def label_enumerate_values(self):
        """read by Trainer.save_checkpoint (core/trainer.py:429-438); BaseDataset's default (appzoo/dataset.py:261-263)"""
        return ["0", "1"]

    def tokenize(self, texts, context_length: int = 32) -> torch.Tensor:
        return self.tokenizer.tokenize_batch(texts, context_length)

    def __getitem__(self, item):
        row = parse_row_by_schema(self.data_rows[item].strip("\n"), self.input_schema)
        try:
            return self.convert_single_row_to_example(row)
        except L.EzclipError:
            raise
        except Exception as e:
            raise RuntimeError("Failed row %d: %s" % (item, e)) from e

    def convert_single_row_to_example(self, row):
        from PIL import Image
        tk = {"input_ids": self.tokenize(row[self.text_col])}                                                  # :217-218
        image = Image.open(io.BytesIO(base64.urlsafe_b64decode(row[self.image_col])))                        # :219
        if image.mode != "RGB":
            # the reference's Wukong pipeline has no convert('RGB') (data.py:82-83): anything but RGB fails in _normalize there
            raise L.EzclipError("WukongCLIPDataset: image mode %r -- only RGB images are defined by the reference "
                                "pipeline; convert('RGB') upstream" % image.mode)
        return {"text": tk, "image": np.asarray(image)}

    def batch_fn(self, features):
        """:231-241 -- 'images' (decoded uint8 arrays) + 'image_size' stand in for 'pixel_values'"""
        images = [f["image"] for f in features]
        return {"input_ids": torch.cat([f["text"]["input_ids"] for f in features], dim=0),
                "images": L.pack_images(images) if self.pack_batches else images, "image_size": self.size}

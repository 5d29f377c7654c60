"""Mirror of ``easynlp.appzoo.text2video_retrieval.data.Text2VideoRetrievalDataset`` (text2video_retrieval/data.py:163-279):
TSV rows ``caption \\t directory of frame images`` parsed by ``input_schema``; the caption becomes 77 BPE ids
(``openclip_tokenize``, appzoo/clip/bpe_tokenizer.py); the frames of a clip are every file of the directory in
``os.listdir`` order (as the reference reads them), padded with black 224x224 frames to ``max_frames`` = 12, with
``video_masks`` [1, 12] marking the real ones.

Decode stays on the CPU; the per-frame ``_resize`` / ``_center_crop`` / ``_normalize`` (data.py:246-252) run on the GPU:
``batch_fn`` ships the decoded frames under ``'images'`` (one list of 12 arrays per clip) and
``Text2VideoRetrieval.forward`` builds the bit-identical float32 ``pixel_values`` [B, 12, 3, 224, 224] with
``ezclip_preprocess_images``.
"""
from __future__ import annotations

import io
import json
import os
from typing import List

import numpy as np
import torch

from ... import lib as L
from ..clip.bpe_tokenizer import SimpleTokenizer, openclip_tokenize
from ..clip.data import parse_row_by_schema

MAX_FRAMES = 12       # data.py:215


def load_clip_frames(directory: str, size: int = 224, max_frames: int = MAX_FRAMES):
    """(frames as uint8 arrays, number of real frames): data.py:230-238 / predictor.py:83-89"""
    from PIL import Image
    frames: List[np.ndarray] = []
    for name in os.listdir(directory):
        img = Image.open(os.path.join(directory, name))
        if img.mode not in ("RGB", "L"):
            raise L.EzclipError("frame %s: image mode %r is not on the GPU pre-processing path (the reference resizes palette / "
                                "alpha images in their own mode); convert('RGB') upstream" % (name, img.mode))
        frames.append(np.asarray(img))
    n = len(frames)
    if n > max_frames:
        raise L.EzclipError("%s holds %d frames; the reference pads to %d and its mask has %d slots -- longer clips break "
                            "there (pixel_values / video_masks disagree): sample %d frames upstream"
                            % (directory, n, max_frames, max_frames, max_frames))
    frames += [np.zeros((size, size, 3), np.uint8)] * (max_frames - n)       # Image.new('RGB', (size, size), (0, 0, 0))
    return frames, n


def video_mask(n_real: int, max_frames: int = MAX_FRAMES) -> torch.Tensor:
    m = torch.zeros((1, max_frames), dtype=torch.int64)
    m[0, :n_real] = 1
    return m


class Text2VideoRetrievalDataset(torch.utils.data.Dataset):

    def __init__(self, pretrained_model_name_or_path, data_file, max_seq_length, input_schema=None, first_sequence=None,
                 label_name=None, second_sequence=None, label_enumerate_values=None, user_defined_parameters=None,
                 skip_first_line: bool = False, pack_batches: bool = False, *args, **kwargs):
        path = pretrained_model_name_or_path
        with open(os.path.join(path, "config.json"), "r") as f:
            self.raw_config = json.load(f)
        if self.raw_config.get("model_type") != "open_clip":
            raise L.EzclipError("Text2VideoRetrievalDataset: only open_clip checkpoints are defined by the reference (data.py:194-195)")
        self.model_type = "open_clip"
        if not input_schema:
            raise L.EzclipError("Text2VideoRetrievalDataset needs input_schema, e.g. 'text:str:1,image:str:1'")
        self.input_schema = input_schema
        self.column_names = [t.split(":")[0] for t in input_schema.split(",")]
        with io.open(data_file) as f:
            if skip_first_line:
                f.readline()
            self.data_rows = f.readlines()
        self.text_col = first_sequence
        self.image_col = second_sequence
        self.openclip_tokenizer = SimpleTokenizer(bpe_path=os.path.join(path, "vocab.txt"))            # data.py:205
        self.max_text_length = max_seq_length
        self.size = self.crop_size = 224
        self.max_frames = MAX_FRAMES
        self.pack_batches = bool(pack_batches)      # batch_fn packs all frames into one uint8 tensor (in the DataLoader worker)

    def __len__(self):
        return len(self.data_rows)

    @property
    def label_enumerate_values(self):
        """read by Trainer.save_checkpoint (core/trainer.py:429-438); BaseDataset's default (appzoo/dataset.py:261-263)"""
        return ["0", "1"]

    def __getitem__(self, item):
        row = parse_row_by_schema(self.data_rows[item].strip("\n"), self.input_schema)
        try:
            return self.convert_single_row_to_example(row)
        except L.EzclipError:
            raise
        except Exception as e:
            raise RuntimeError("Failed row %d: %s" % (item, e)) from e

    def convert_single_row_to_example(self, row):
        frames, n = load_clip_frames(row[self.image_col], self.size, self.max_frames)
        tk = {"input_ids": openclip_tokenize([row[self.text_col]], context_length=77, _tokenizer=self.openclip_tokenizer)}
        return {"text": tk, "frames": frames, "video_masks": video_mask(n, self.max_frames)}

    def batch_fn(self, features):
        """data.py:257-279; 'images' (per clip: 12 decoded uint8 frames) + 'image_size' stand in for 'pixel_values'"""
        clips = [f["frames"] for f in features]
        if self.pack_batches:
            clips = dict(L.pack_images([fr for c in clips for fr in c]), clips=len(clips))
        return {"input_ids": torch.cat([f["text"]["input_ids"] for f in features], dim=0),
                "video_masks": torch.cat([f["video_masks"] for f in features], dim=0),
                "images": clips, "image_size": self.size, "label_ids": []}

"""CPU oracle for the image half of ``CLIPDataset.convert_single_row_to_example``
(easynlp/appzoo/clip/data.py:256-262): ``_resize`` (:52-72, PIL BICUBIC, shorter side -> 224) ->
``_center_crop`` (:29-50) -> ``_normalize`` (:101-135: convert('RGB'), /255 in float32, (x - mean) / std).

TEST INFRASTRUCTURE ONLY (see clip_oracle.py).  The resampling itself lives in a third-party dependency of the
reference, Pillow (``PIL.Image.resize`` -> libImaging/Resample.c; the image ships Pillow 12.2.0, the reference does not
pin a version).  It is restated here in numpy integer arithmetic exactly as Pillow's 8-bit path does it -- a separable
two-pass convolution (horizontal, then vertical, the intermediate rounded to uint8) with per-output-pixel windows of
fixed-point coefficients (22 fractional bits) of the Keys bicubic kernel (a = -0.5) stretched by the downscale factor
(antialiasing) -- and pinned bit for bit against Pillow itself (tests/test_preprocess.py; Pillow is present in the
build container and on the GPU box, so the pin runs in both places).
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c: coefficients are ints with 22 fractional bits
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # data.py:101
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def bicubic_filter(x: float) -> float:
    """Resample.c bicubic_filter (Keys, a = -0.5), support 2."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full box (0, in_size):
    returns ksize, bounds [out_size, 2] (first source index, count), kk [out_size, ksize] int32."""
    in0, in1 = np.float32(0.0), np.float32(in_size)
    scale = float(np.float32(in1 - in0)) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = float(in0) + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _clip8(acc: np.ndarray) -> np.ndarray:
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resample_axis0(img: np.ndarray, out_size: int, first: int = 0, count: int = None) -> np.ndarray:
    """One pass along axis 0 of a uint8 array [n, ...]: output rows first .. first+count."""
    n = img.shape[0]
    _, bounds, kk = precompute_coeffs(n, out_size)
    count = out_size - first if count is None else count
    out = np.empty((count,) + img.shape[1:], np.uint8)
    for o in range(count):
        xmin, xmax = bounds[first + o]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(xmax):
            acc += img[xmin + x].astype(np.int64) * int(kk[first + o, x])
        out[o] = _clip8(acc)
    return out


def resize_bicubic(rgb: np.ndarray, new_w: int, new_h: int) -> np.ndarray:
    """``Image.resize((new_w, new_h), Image.BICUBIC)`` of an RGB uint8 [H, W, 3] array (ImagingResample: horizontal
    pass over the source rows the vertical pass needs, then the vertical pass)."""
    h, w, _ = rgb.shape
    out = rgb
    if new_w != w:
        out = np.ascontiguousarray(resample_axis0(np.ascontiguousarray(out.transpose(1, 0, 2)), new_w).transpose(1, 0, 2))
    if new_h != h:
        out = resample_axis0(out, new_h)
    return out


def resized_size(w: int, h: int, size: int):
    """data.py:62-71 -- shorter side -> size, the other int(size * long / short); unchanged if already there."""
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return w, h
    new_short, new_long = size, int(size * long / short)
    return (new_short, new_long) if w <= h else (new_long, new_short)


def crop_origin(w: int, h: int, crop: int):
    """data.py:43-48"""
    return int((w - crop + 1) * 0.5), int((h - crop + 1) * 0.5)      # left, top


def normalize_lut(mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
    """[3, 256] float32: (np.float32(v) / 255.0 - mean_c) / std_c evaluated as the reference does (data.py:93-94,119-122,
    131-132: float32 array / 255.0, mean/std cast to the array's dtype)."""
    v = np.arange(256).astype(np.float32) / 255.0
    m = np.array(mean).astype(np.float32)
    s = np.array(std).astype(np.float32)
    return ((v[None, :] - m[:, None]) / s[:, None]).astype(np.float32)


def preprocess(rgb: np.ndarray, size: int = 224, crop: int = 224, mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
    """RGB uint8 [H, W, 3] -> float32 [3, crop, crop], the ``pixel_values`` of one example."""
    h, w, _ = rgb.shape
    nw, nh = resized_size(w, h, size)
    img = resize_bicubic(rgb, nw, nh)
    left, top = crop_origin(nw, nh, crop)
    assert left >= 0 and top >= 0 and left + crop <= nw and top + crop <= nh, "crop larger than the resized image"
    img = img[top:top + crop, left:left + crop]
    lut = normalize_lut(mean, std)
    return np.stack([lut[c][img[:, :, c]] for c in range(3)], axis=0)


def reference_pipeline_pil(pil_image, size: int = 224, crop: int = 224):
    """The reference's own sequence on a PIL image (restating data.py:29-135 call for call, PIL doing the resampling):
    used to pin everything above."""
    from PIL import Image
    w, h = pil_image.size
    nw, nh = resized_size(w, h, size)
    img = pil_image if (nw, nh) == (w, h) else pil_image.resize((nw, nh), Image.BICUBIC)      # data.py:72
    left, top = crop_origin(nw, nh, crop)
    img = img.crop((left, top, left + crop, top + crop))                                      # data.py:50
    arr = np.array(img.convert("RGB")).astype(np.float32) / 255.0                             # data.py:119-120,93-94
    arr = arr.transpose(2, 0, 1)
    m = np.array(CLIP_MEAN).astype(arr.dtype)
    s = np.array(CLIP_STD).astype(arr.dtype)
    return (arr - m[:, None, None]) / s[:, None, None]                                        # data.py:131-132

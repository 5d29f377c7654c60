"""Image pre-processing (input pipeline) -- CLIPDataset.convert_single_row_to_example's image branch
(easynlp/appzoo/clip/data.py:29-135,256-262) as a HIP kernel pair.

Pin chain: Pillow itself (the reference's resampler; present here and on the GPU box) == numpy restatement
(oracle/preprocess_oracle.py) == host window tables of the library (CPU tests, bit for bit) == device output (GPU tests,
bit for bit: the integer resize is exact and the float normalisation is a 256-entry table evaluated as numpy does)."""
import ctypes as C

import numpy as np
import pytest
import torch

from easynlp_amd import lib as L
from oracle import preprocess_oracle as P

PIL = pytest.importorskip("PIL.Image")

SIZES = [(500, 375), (375, 500), (640, 480), (224, 224), (300, 224), (224, 300), (100, 80), (31, 200), (1024, 768),
         (225, 224), (223, 400), (257, 255), (2000, 300), (17, 17), (448, 448), (3, 1000), (4032, 3024)]


def _img(w, h, seed):
    rs = np.random.RandomState(seed)
    if seed % 2:
        return rs.randint(0, 256, (h, w, 3)).astype(np.uint8)          # white noise: every rounding case
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy) % 256], axis=2)
    return ((base + rs.randint(0, 32, (h, w, 3))) % 256).astype(np.uint8)   # structured + saturating edges


@pytest.mark.parametrize("w,h", SIZES[:-1])
def test_oracle_resize_and_pipeline_equal_pillow(w, h):
    a = _img(w, h, w + h)
    nw, nh = P.resized_size(w, h, 224)
    want = a if (nw, nh) == (w, h) else np.array(PIL.fromarray(a).resize((nw, nh), PIL.BICUBIC))
    assert np.array_equal(P.resize_bicubic(a, nw, nh), want)
    got, ref = P.preprocess(a), P.reference_pipeline_pil(PIL.fromarray(a))
    assert ref.dtype == np.float32 and np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_greyscale_resize_commutes_with_rgb_conversion():
    """the reference resizes in the image's own mode and converts to RGB afterwards (data.py:119); for 'L' that equals
    replicating first (what lib.preprocess_images does)"""
    g = _img(300, 200, 5)[:, :, 0]
    ref = P.reference_pipeline_pil(PIL.fromarray(g, mode="L"))
    got = P.preprocess(np.repeat(g[:, :, None], 3, axis=2))
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("in_size,out_size", [(375, 224), (500, 298), (80, 224), (17, 224), (3024, 224), (224, 224), (2000, 1493),
                                              (255, 224), (1000, 74666)])
def test_library_window_tables_equal_oracle(in_size, out_size):
    lib = L.load()
    first = max(0, (out_size - 224 + 1) // 2)
    count = min(224, out_size - first)
    ks_o, b_o, k_o = P.precompute_coeffs(in_size, out_size) if in_size != out_size else (1, None, None)
    ksize = C.c_int()
    bounds = (C.c_int * (2 * count))()
    cap = count * (ks_o + 2)
    kk = (C.c_int * cap)()
    L.check(lib.ezclip_op_resample_table(in_size, out_size, first, count, C.byref(ksize), bounds, kk, cap))
    b = np.array(bounds[:]).reshape(count, 2)
    if in_size == out_size:
        assert ksize.value == 1 and np.array_equal(b[:, 0], np.arange(first, first + count)) and (b[:, 1] == 1).all()
        assert (np.array(kk[:count]) == 1 << 22).all()
        return
    assert ksize.value == ks_o
    assert np.array_equal(b, b_o[first:first + count])
    assert np.array_equal(np.array(kk[:count * ks_o]).reshape(count, ks_o), k_o[first:first + count])


def test_rejects_what_the_path_does_not_cover():
    lib = L.load()
    d = (L.EzclipImageDesc * 1)()
    d[0].offset, d[0].width, d[0].height = 0, 0, 10
    assert lib.ezclip_preprocess_workspace_bytes(d, 1, 224, 224) == 0 and "size 0 x 10" in L.last_error()
    d[0].width = 100
    assert lib.ezclip_preprocess_workspace_bytes(d, 1, 100, 224) == 0 and "smaller than the 224 crop" in L.last_error()
    with pytest.raises(L.EzclipError):
        L.preprocess_images([np.zeros((10, 10, 4), np.uint8)])
    with pytest.raises(L.EzclipError):
        L.preprocess_images([np.zeros((10, 10, 3), np.float32)])


# ------------------------------------------------------------------------------------------------ GPU

def test_pack_images_layout():
    """host side of ezclip_preprocess_images' input: HWC RGB bytes, 16-byte aligned starts, (offset, width, height) rows;
    greyscale replicated, non-contiguous views made contiguous -- checked against a direct restatement"""
    rs = np.random.RandomState(0)
    imgs = [rs.randint(0, 256, (37, 50, 3), dtype=np.uint8), rs.randint(0, 256, (20, 31), dtype=np.uint8),
            rs.randint(0, 256, (5, 7, 1), dtype=np.uint8), rs.randint(0, 256, (64, 64, 3), dtype=np.uint8)[:, ::2]]
    p = L.pack_images(imgs)
    assert L.is_packed_images(p) and not L.is_packed_images(imgs) and p["data"].dtype == torch.uint8 and p["desc"].dtype == torch.int64
    off = 0
    for (o, w, h), im in zip(p["desc"].tolist(), imgs):
        a = im if im.ndim == 3 else im[:, :, None]
        a = np.repeat(a, 3, axis=2) if a.shape[2] == 1 else a
        assert (o, w, h) == (off, a.shape[1], a.shape[0]) and o % 16 == 0
        assert np.array_equal(p["data"].numpy()[o:o + a.size].reshape(a.shape), a)
        off += (a.size + 15) // 16 * 16
    assert p["data"].numel() == off + 16
    with pytest.raises(L.EzclipError):
        L.pack_images([])
    with pytest.raises(L.EzclipError):
        L.pack_images([np.zeros((4, 4, 4), np.uint8)])
    with pytest.raises(L.EzclipError):
        L.pack_images([np.zeros((4, 4, 3), np.float32)])


@pytest.mark.gpu
def test_device_preprocess_is_bit_identical_to_the_reference_pipeline():
    imgs = [_img(w, h, i) for i, (w, h) in enumerate(SIZES)]
    out = L.preprocess_images(imgs).cpu().numpy()
    assert out.shape == (len(imgs), 3, 224, 224) and out.dtype == np.float32
    for i, a in enumerate(imgs):
        ref = P.reference_pipeline_pil(PIL.fromarray(a))
        assert np.array_equal(out[i].view(np.uint32), ref.view(np.uint32)), (SIZES[i], np.abs(out[i] - ref).max())
    # a second call reuses the library's pinned staging buffer; single image; greyscale
    one = L.preprocess_images([imgs[2]]).cpu().numpy()
    assert np.array_equal(one[0], out[2])
    g = imgs[0][:, :, 0]
    ref = P.reference_pipeline_pil(PIL.fromarray(g, mode="L"))
    assert np.array_equal(L.preprocess_images([g]).cpu().numpy()[0].view(np.uint32), ref.view(np.uint32))


@pytest.mark.gpu
def test_device_preprocess_other_size_and_statistics():
    imgs = [_img(320, 240, 11), _img(240, 320, 12)]
    out = L.preprocess_images(imgs, size=160, crop=128).cpu().numpy()
    for i, a in enumerate(imgs):
        want = P.preprocess(a, size=160, crop=128)
        assert np.array_equal(out[i].view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_predictor_preprocess_decodes_and_uses_the_gpu_pipeline(tmp_path):
    """CLIPPredictor.preprocess on the reference's wire format (urlsafe base64 of an encoded image, predictor.py:100-110)"""
    import base64
    import io
    from easynlp_amd.appzoo.clip import CLIPPredictor
    from oracle import clip_oracle as O
    from oracle import ref_harness as R
    cfg = O.CONFIGS["tiny"]
    R.write_checkpoint_dir(str(tmp_path), cfg, O.make_state_dict(cfg, 3))
    pred = CLIPPredictor(str(tmp_path), first_sequence="text", second_sequence="image", sequence_length=16)
    res = int(cfg["image_resolution"])
    recs = []
    for i, (w, h) in enumerate([(90, 60), (64, 100)]):
        buf = io.BytesIO()
        PIL.fromarray(_img(w, h, i)).save(buf, format="PNG")
        recs.append({"image": base64.urlsafe_b64encode(buf.getvalue()).decode()})
    out = pred.preprocess(recs)
    for r, (w, h) in zip(out, [(90, 60), (64, 100)]):
        img = PIL.open(io.BytesIO(base64.urlsafe_b64decode(r["image"])))
        ref = P.reference_pipeline_pil(img, size=res, crop=res)
        assert tuple(r["pixel_values"].shape) == (1, 3, res, res)
        assert np.array_equal(r["pixel_values"].cpu().numpy()[0].view(np.uint32), ref.view(np.uint32))
    feats = pred.run(recs)
    assert len(feats) == 2 and "image_feat" in feats[0]
    # text records: WordPiece ids from the checkpoint's vocab.txt, padded to sequence_length (predictor.py:95-101)
    trecs = pred.preprocess([{"text": "tok7 tok9 tok11"}, {"text": "tok8"}])
    assert tuple(trecs[0]["input_ids"].shape) == (1, 16) and int(trecs[0]["attention_mask"].sum()) == 5
    assert trecs[0]["input_ids"][0, :6].tolist() == [2, 12, 14, 16, 3, 0]      # [CLS] tok7 tok9 tok11 [SEP] [PAD]: real ids, not [UNK]
    tf = pred.run([{"text": "tok7 tok9 tok11"}, {"text": "tok8"}])
    assert len(tf) == 2 and "text_feat" in tf[0]


@pytest.mark.gpu
def test_device_built_window_tables_equal_host_tables():
    """ezclip_preprocess_images builds its resampling windows on the device by default (explicitly rounded double
    arithmetic in Pillow's operation order); they must equal the host tables -- themselves equal to the Pillow-pinned
    oracle (CPU test above) -- bit for bit, over many size pairs including extreme ratios."""
    lib = L.load()
    rs = np.random.RandomState(5)
    pairs = [(375, 224), (500, 298), (80, 224), (17, 224), (3024, 224), (224, 224), (2000, 1493), (255, 224), (1000, 74666),
             (4032, 298), (7, 224), (225, 224), (223, 224)]
    pairs += [(int(rs.randint(8, 5000)), int(rs.randint(224, 2000))) for _ in range(120)]
    for in_size, out_size in pairs:
        first = max(0, (out_size - 224 + 1) // 2)
        count = min(224, out_size - first)
        ksize = C.c_int()
        cap = count * 400
        hb = (C.c_int * (2 * count))()
        hk = (C.c_int * cap)()
        L.check(lib.ezclip_op_resample_table(in_size, out_size, first, count, C.byref(ksize), hb, hk, cap))
        ks = ksize.value
        db = torch.full((count, 2), -7, dtype=torch.int32, device="cuda")
        dk = torch.full((count, ks), -7, dtype=torch.int32, device="cuda")
        L.check(lib.ezclip_op_resample_table_device(in_size, out_size, first, count, L.ptr(db), L.ptr(dk), L.stream_ptr()))
        assert np.array_equal(db.cpu().numpy(), np.array(hb[:]).reshape(count, 2)), (in_size, out_size)
        assert np.array_equal(dk.cpu().numpy(), np.array(hk[:count * ks]).reshape(count, ks)), (in_size, out_size)


@pytest.mark.gpu
def test_device_preprocess_with_host_tables_switch():
    imgs = [_img(w, h, i) for i, (w, h) in enumerate(SIZES[:8])]
    lib = L.load()
    dev = L.preprocess_images(imgs).cpu().numpy()
    L.check(lib.ezclip_debug_set(5, 0))
    try:
        host = L.preprocess_images(imgs).cpu().numpy()
    finally:
        L.check(lib.ezclip_debug_set(5, 1))
    assert np.array_equal(dev.view(np.uint32), host.view(np.uint32))

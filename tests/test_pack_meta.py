"""Host side of packed text batches: which tokens the text tower is given, in which order, and when a packing may
be used under train-mode dropout (CPU only: `pack_meta` works on host tensors)."""
import torch

from easynlp_amd.appzoo.clip.model import HipClipEngine


def _engine():
    eng = HipClipEngine.__new__(HipClipEngine)      # (no device, no library: only the packing logic)
    eng._pack_cache = None
    eng._drop = (0.0, 0.0)
    eng.pack_hf_dropout = True
    return eng


def _expected_rowmap(keep):
    B, S = keep.shape
    return [b * S + t for b in range(B) for t in range(S) if keep[b, t]]


def test_prefix_batches_are_packed_sample_by_sample():
    eng = _engine()
    S = 16
    lens = [16, 1, 5, 9]
    ids = torch.zeros(4, S, dtype=torch.int64)
    for b, n in enumerate(lens):
        ids[b, :n] = torch.arange(1, n + 1)
    m = eng.pack_meta(ids, device="cpu")
    assert m is not None and m["rows"] == sum(lens) and m["longest"] == 16 and m["shape"] == (4, S)
    assert m["lens"].tolist() == lens and m["cu"].tolist() == [0, 16, 17, 22]
    assert m["rowmap"].tolist() == _expected_rowmap(ids.ne(0))
    assert m["prefix"] is True
    for drop in ((0.0, 0.0), (0.1, 0.0), (0.0, 0.1)):
        eng._drop = drop
        assert eng.usable(m)
    # the huggingface_clip branch (explicit position / type / mask tensors): packed under dropout too since round 3
    # (tests/test_hf_gpu.py::test_hf_packed_text_tower_with_dropout_equals_the_padded_one); EZCLIP_PACK_HF_DROPOUT=0 switches it off
    extras = (None, None, torch.ones_like(ids))
    eng._drop = (0.0, 0.0)
    assert eng.usable(m, extras)
    eng._drop = (0.1, 0.1)
    assert eng.usable(m, extras)
    eng.pack_hf_dropout = False
    assert not eng.usable(m, extras)


def test_holes_masked_cls_and_empty_sentences():
    eng = _engine()
    S = 12
    ids = torch.zeros(5, S, dtype=torch.int64)
    ids[0, :6] = 7
    ids[1, :8] = 7
    ids[1, 3] = 0            # a pad id inside the sentence: kept tokens are no prefix
    ids[2, :4] = 7
    ids[2, 0] = 0            # masked CLS token: still kept (row cu[b] must be t = 0)
    # sentence 3: no unmasked key at all -> kept whole (its softmax is uniform over ALL positions in the reference)
    ids[4, :2] = 7
    m = eng.pack_meta(ids, device="cpu")
    keep = ids.ne(0)
    keep[2, 0] = True
    keep[3, :] = True
    assert m["rowmap"].tolist() == _expected_rowmap(keep)
    assert m["lens"].tolist() == [6, 7, 4, S, 2]
    assert m["prefix"] is False
    eng._drop = (0.0, 0.0)
    assert eng.usable(m)
    eng._drop = (0.1, 0.1)   # a packed position would not be the padded one: the masks could not be regenerated
    assert not eng.usable(m)
    # an explicit attention mask decides instead of the ids
    am = torch.ones_like(ids)
    am[:, 6:] = 0
    m2 = eng.pack_meta(ids, am, device="cpu")
    assert m2["lens"].tolist() == [6] * 5 and m2["prefix"] is True


def test_packing_that_does_not_pay_is_declined_and_results_are_cached_per_tensor():
    eng = _engine()
    full = torch.ones(4, 32, dtype=torch.int64)
    assert eng.pack_meta(full, device="cpu") is None                  # > 90 % of the rows kept
    long = torch.zeros(2, 400, dtype=torch.int64)
    long[0, :300] = 1
    long[1, :3] = 1
    assert eng.pack_meta(long, device="cpu") is None                  # longest sentence > 256 tokens
    ids = torch.zeros(3, 20, dtype=torch.int64)
    ids[:, :4] = 5
    a = eng.pack_meta(ids, device="cpu")
    assert eng.pack_meta(ids, device="cpu") is a                      # same tensor object, unmodified: cached
    ids[0, 5] = 9                                                     # in-place change bumps the version counter
    b = eng.pack_meta(ids, device="cpu")
    assert b is not a and b["prefix"] is False

"""DevicePrefetcher (easynlp_amd/appzoo/clip/data.py): the device loader that goes where the reference puts its own
(`pl.MpDeviceLoader(self._train_loader, self._device)`, core/trainer.py:215-218).  CPU: pass-through contract; GPU: values, order,
stream ordering under a consumer that overwrites what it received, exceptions, early exit."""
import pytest
import torch

from easynlp_amd.appzoo.clip import DevicePrefetcher


def _batches(n, seed=0, rows=5):
    g = torch.Generator().manual_seed(seed)
    return [{"pixel_values": torch.randn(rows + k, 3, 8, 8, generator=g), "input_ids": torch.randint(0, 99, (rows + k, 12), generator=g),
             "label_ids": [], "image_size": 224, "note": "batch %d" % k} for k in range(n)]


def test_cpu_device_is_the_identity_and_len_is_the_loaders():
    bs = _batches(4)
    pf = DevicePrefetcher(bs, "cpu")
    assert len(pf) == 4
    out = list(pf)
    assert all(a is b for a, b in zip(out, bs))
    assert list(pf) == out                      # a second epoch


def test_works_on_a_real_dataloader_with_a_collate_fn():
    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 10

        def __getitem__(self, i):
            return {"x": torch.full((3,), float(i)), "i": i}

    def collate(fs):
        return {"pixel_values": torch.stack([f["x"] for f in fs]), "label_ids": [], "ids": [f["i"] for f in fs]}

    dl = torch.utils.data.DataLoader(DS(), batch_size=4, collate_fn=collate)
    got = list(DevicePrefetcher(dl, "cpu"))
    assert [b["ids"] for b in got] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]] and len(DevicePrefetcher(dl, "cpu")) == 3


@pytest.mark.gpu
def test_batches_arrive_on_the_device_in_order_with_the_same_values():
    bs = _batches(9, seed=3)
    bs[4]["pixel_values"] = bs[4]["pixel_values"].pin_memory()          # pinned and pageable batches mixed
    bs[6]["input_ids"] = bs[6]["input_ids"].cuda()                      # already on the device: passes through
    seen = 0
    for k, b in enumerate(DevicePrefetcher(bs, "cuda:0", depth=2)):
        assert b["pixel_values"].is_cuda and b["input_ids"].is_cuda
        assert b["label_ids"] == [] and b["image_size"] == 224 and b["note"] == "batch %d" % k
        assert torch.equal(b["pixel_values"].cpu(), bs[k]["pixel_values"]) and torch.equal(b["input_ids"].cpu(), bs[k]["input_ids"].cpu())
        # a consumer that keeps the device busy and scribbles over what it got: later batches must not be affected
        x = b["pixel_values"]
        for _ in range(20):
            x = x * 1.0001 + 1.0
        b["pixel_values"].fill_(float("nan"))
        seen += 1
    assert seen == 9
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_large_batches_overlap_safely_with_compute_on_the_current_stream():
    """64 MB batches, a long-running consumer kernel sequence per batch, depth 1: every batch's checksum is its own (the copy of batch
    k + 1 runs while batch k is being read; record_stream keeps batch k's memory from being recycled under the reader)."""
    g = torch.Generator().manual_seed(5)
    bs = [{"pixel_values": torch.randn(16, 1024, 1024, generator=g).pin_memory()} for _ in range(6)]
    want = [float(b["pixel_values"].double().sum()) for b in bs]
    sums = []
    w = torch.randn(1024, 1024, device="cuda:0")
    for b in DevicePrefetcher(bs, "cuda:0", depth=1):
        x = b["pixel_values"]
        y = x
        for _ in range(8):
            y = (y @ w) * 1e-3                                              # keeps the stream busy while the next copy runs
        sums.append((x.double().sum(), y.abs().mean()))
        del b, x, y
    torch.cuda.synchronize()
    for (s, m), wv in zip(sums, want):
        assert abs(float(s) - wv) < 1e-6 * max(1.0, abs(wv)) and torch.isfinite(m)


@pytest.mark.gpu
def test_loader_exception_reaches_the_consumer_and_early_exit_stops_the_thread():
    def gen():
        yield from _batches(2)
        raise ValueError("bad shard")

    class L:
        def __iter__(self):
            return gen()

        def __len__(self):
            return 3

    it = iter(DevicePrefetcher(L(), "cuda:0"))
    assert next(it)["pixel_values"].is_cuda and next(it)["pixel_values"].is_cuda
    with pytest.raises(ValueError, match="bad shard"):
        next(it)
    import threading
    n0 = sum(t.name == "ezclip-device-prefetch" and t.is_alive() for t in threading.enumerate())
    for k, b in enumerate(DevicePrefetcher(_batches(50), "cuda:0", depth=1)):
        if k == 2:
            break                                                            # generator closed mid-epoch
    import time
    time.sleep(0.5)
    n1 = sum(t.name == "ezclip-device-prefetch" and t.is_alive() for t in threading.enumerate())
    assert n1 <= n0, (n0, n1)


@pytest.mark.gpu
def test_forward_takes_prefetched_batches_like_host_batches(tmp_path):
    from easynlp_amd.appzoo.clip import CLIPApp
    from oracle import clip_oracle as O
    from oracle import ref_harness as R
    cfg = O.CONFIGS["tiny"]
    R.write_checkpoint_dir(str(tmp_path), cfg, O.make_state_dict(cfg, 3))
    app = CLIPApp(str(tmp_path), user_defined_parameters={"clip_compute_dtype": "fp32"}).cuda()
    app.eval()
    host = []
    for k in range(3):
        px, ids = O.make_inputs(cfg, 4 + k, 24, 10 + k)
        host.append({"pixel_values": px, "input_ids": ids, "label_ids": []})
    with torch.no_grad():
        want = [app({k: v for k, v in b.items()})["logits_per_text"].cpu() for b in host]
        got = [app(b)["logits_per_text"].cpu() for b in DevicePrefetcher(host, "cuda:0")]
    for a, b in zip(got, want):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_packed_image_batches_cross_on_the_copy_stream_and_preprocess_to_the_same_pixels():
    import numpy as np
    from easynlp_amd import lib as L
    rng = np.random.RandomState(4)
    raw = [[rng.randint(0, 256, (h, w, 3), dtype=np.uint8) for h, w in ((240, 320), (300, 224), (224, 224), (97, 180))] for _ in range(3)]
    host = [{"images": L.pack_images(ims), "image_size": 224, "label_ids": []} for ims in raw]
    want = [L.preprocess_images(ims, size=224, crop=224, device="cuda:0").cpu() for ims in raw]
    for b, w in zip(DevicePrefetcher(host, "cuda:0"), want):
        assert b["images"]["data"].is_cuda and not b["images"]["desc"].is_cuda
        assert torch.equal(L.preprocess_images(b["images"], size=224, crop=224, device="cuda:0").cpu(), w)
    assert not host[0]["images"]["data"].is_cuda                           # the wrapped loader's batch is not modified

#!/usr/bin/env python
"""Golden fixtures of the reference's ModifiedResNet tower for oracle/resnet_oracle.py:
    python tools/make_golden_resnet.py        (build container only: imports /root/reference)
writes tests/golden/rn_tiny_b3.npz = {pixels, image_features, stem, layer4, meta}  (eval mode) and
tests/golden/rn_tiny_train_b4.npz = {pixels, probe, image_features, grad:<name> for every parameter, stat:<name> for every updated
running statistic, meta}  (train mode: BatchNorm batch statistics, loss = sum(features * probe), torch autograd of the REFERENCE module)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_harness as R            # noqa: E402
from oracle import resnet_oracle as RO         # noqa: E402

CFG = dict(layers=(1, 2, 1, 1), width=16, output_dim=24, resolution=64, wseed=7, iseed=3, batch=3)


def reference_tower(cfg, sd):
    R.install_shims()
    from easynlp.modelzoo.models.clip.modeling_chineseclip import ModifiedResNet
    m = ModifiedResNet(cfg["layers"], cfg["output_dim"], cfg["width"] * 32 // 64, cfg["resolution"], cfg["width"])
    own = {k[len("visual."):]: v for k, v in sd.items()}
    missing, unexpected = m.load_state_dict(own, strict=False)
    assert all(k.endswith("num_batches_tracked") for k in missing), missing
    assert not unexpected, unexpected
    return m.eval()


TRAIN_CFG = dict(layers=(1, 2, 1, 1), width=16, output_dim=24, resolution=64, wseed=9, iseed=5, batch=4)


def reference_train_step(cfg, sd, px, probe):
    """(features, {visual.<name>: grad}, {visual.<name>: updated running statistic}) of the reference module in train() mode"""
    m = reference_tower(cfg, sd).train()
    out = m(px)
    (out * probe).sum().backward()
    grads = {"visual." + k: p.grad.detach().clone() for k, p in m.named_parameters()}
    stats = {"visual." + k: v.detach().clone() for k, v in m.state_dict().items() if k.endswith(("running_mean", "running_var"))}
    return out.detach(), grads, stats


def sample_index(name, numel, k=64):
    """64 positions of a flattened tensor, seeded by its name (shared with tests/test_resnet_oracle.py)"""
    import zlib
    return np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff).choice(numel, size=k, replace=False).astype(np.int64)


def main_train():
    c = TRAIN_CFG
    sd = RO.make_state_dict(c["layers"], c["width"], c["output_dim"], c["resolution"], c["wseed"])
    g = torch.Generator().manual_seed(c["iseed"])
    px = torch.randn(c["batch"], 3, c["resolution"], c["resolution"], generator=g)
    probe = torch.randn(c["batch"], c["output_dim"], generator=g)
    out, grads, stats = reference_train_step(c, sd, px, probe)
    path = os.path.join(ROOT, "tests", "golden", "rn_tiny_train_b4.npz")
    arrays = dict(pixels=px.numpy(), probe=probe.numpy(), image_features=out.numpy(),
                  meta=np.frombuffer(json.dumps(c).encode(), dtype=np.uint8))
    for k, v in grads.items():             # small tensors whole; large ones as (L2 norm, 64 entries at seeded positions)
        if v.numel() <= 4096:
            arrays["grad:" + k] = v.numpy()
        else:
            idx = sample_index(k, v.numel())
            arrays["gnorm:" + k] = np.float64(v.double().norm().item())
            arrays["gsamp:" + k] = v.reshape(-1)[torch.from_numpy(idx)].numpy()
    arrays.update({"stat:" + k: v.numpy() for k, v in stats.items()})
    np.savez_compressed(path, **arrays)
    print("wrote", path, out.shape, len(grads), "gradients", len(stats), "running statistics", os.path.getsize(path), "bytes")


def main():
    c = CFG
    sd = RO.make_state_dict(c["layers"], c["width"], c["output_dim"], c["resolution"], c["wseed"])
    m = reference_tower(c, sd)
    g = torch.Generator().manual_seed(c["iseed"])
    px = torch.randn(c["batch"], 3, c["resolution"], c["resolution"], generator=g)
    taps = {}
    hooks = [m.avgpool.register_forward_hook(lambda _m, _i, o: taps.setdefault("stem", o.detach().clone())),
             m.layer4.register_forward_hook(lambda _m, _i, o: taps.setdefault("layer4", o.detach().clone()))]
    with torch.no_grad():
        out = m(px)
    for h in hooks:
        h.remove()
    path = os.path.join(ROOT, "tests", "golden", "rn_tiny_b3.npz")
    np.savez_compressed(path, pixels=px.numpy(), image_features=out.numpy(), stem=taps["stem"].numpy(),
                        layer4=taps["layer4"].numpy(), meta=np.frombuffer(json.dumps(c).encode(), dtype=np.uint8))
    print("wrote", path, out.shape, float(out.abs().max()))


if __name__ == "__main__":
    main()
    main_train()

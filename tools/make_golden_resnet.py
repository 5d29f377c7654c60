#!/usr/bin/env python
"""Golden fixture of the reference's ModifiedResNet tower (eval mode) for oracle/resnet_oracle.py:
    python tools/make_golden_resnet.py        (build container only: imports /root/reference)
writes tests/golden/rn_tiny_b3.npz = {pixels, image_features, stem, layer4, meta}."""
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

#!/usr/bin/env python
"""Where (which rows / channels) does the first wrong intermediate of the ModifiedResNet training backward differ from the oracle?
Width 48, batch 4, 64 x 64 (tools/rn_train_stage_diff.py: layer4's conv2 BatchNorm backward is the first stage off).  The device dumps
its first intermediates (EZCLIP_RN_DEBUG_DUMP), the oracle's are recorded by wrapping its helpers."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import resnet_oracle as RO  # noqa: E402

layers, width, e, res, B = (1, 1, 1, 1), 48, 128, 64, 4
child = os.path.join(ROOT, "tools", "rn_train_stage_diff.py")
r = subprocess.run([sys.executable, child, "child", "1"], capture_output=True, text=True,
                   env=dict(os.environ, EZCLIP_RN_DEBUG="1", EZCLIP_RN_DEBUG_DUMP="6"), timeout=600)
print("device rc", r.returncode)
lines = [ln for ln in r.stderr.splitlines() if ln.startswith("[rn-dbg]")][:8]
print("\n".join(lines))
sd = {k: v.double() for k, v in RO.make_state_dict(layers, width, e, res, 17).items()}
g = torch.Generator().manual_seed(6)
px, probe = torch.randn(B, 3, res, res, generator=g).double(), torch.randn(B, e, generator=g).double()
with torch.no_grad():
    raw = RO.modified_resnet_forward(sd, layers, width, px, train=True, new_stats={})
nrm = raw.norm(dim=-1, keepdim=True)
out = raw / nrm
d_raw = (probe - out * (out * probe).sum(dim=-1, keepdim=True)) / nrm
rec = []
bn_bwd, pool_bwd = RO._bn_train_bwd, RO._avgpool_bwd
RO._bn_train_bwd = lambda dy, xh, rstd, gamma: (lambda r_: (rec.append(("bn", dy.clone(), r_[0].clone())), r_)[1])(bn_bwd(dy, xh, rstd, gamma))
RO._avgpool_bwd = lambda dy, s: (lambda r_: (rec.append(("pool", dy.clone(), r_.clone())), r_)[1])(pool_bwd(dy, s))
try:
    RO.train_step_grads_by_steps(sd, layers, width, px, d_raw)
finally:
    RO._bn_train_bwd, RO._avgpool_bwd = bn_bwd, pool_bwd


def nhwc(t, cp):
    Bn, C, H, W = t.shape
    o = torch.zeros(Bn * H * W, cp, dtype=torch.float64)
    o[:, :C] = t.permute(0, 2, 3, 1).reshape(-1, C)
    return o


def load(no, rows, cp):
    return torch.from_numpy(np.fromfile("/tmp/rn_dbg_%d.bin" % no, dtype=np.float32).astype(np.float64)).reshape(rows, cp)


def report(name, dev, ref):
    d = (dev - ref).abs()
    print("%s: shape %s  max |diff| %.3e (max |ref| %.3e)  rel-L2 %.3e" % (name, tuple(ref.shape), float(d.max()), float(ref.abs().max()),
                                                                       float(d.norm() / ref.norm())))
    bad = (d > 1e-4 * float(ref.abs().max())).nonzero()
    print("   elements off by > 1e-4 of the max: %d of %d" % (bad.shape[0], d.numel()))
    if bad.shape[0]:
        rows_, cols_ = bad[:, 0], bad[:, 1]
        print("   rows: min %d max %d, distinct %d;  channels: min %d max %d, distinct %d" % (int(rows_.min()), int(rows_.max()), len(set(rows_.tolist())),
                                                                                           int(cols_.min()), int(cols_.max()), len(set(cols_.tolist()))))
        print("   first few (row, channel, device, oracle):", [(int(a), int(b), float(dev[a, b]), float(ref[a, b])) for a, b in bad[:6].tolist()])
        print("   channel histogram (per 64):", np.bincount((cols_.numpy() // 64), minlength=ref.shape[1] // 64).tolist())
        print("   row histogram (per 16):", np.bincount((rows_.numpy() // 16), minlength=(ref.shape[0] + 15) // 16).tolist())


# oracle record order: bn(c3 of layer4) [0], pool (d_o2 -> d_y2) [1], bn(c2) [2], ...
assert rec[0][0] == "bn" and rec[1][0] == "pool" and rec[2][0] == "bn"
mask_in_c2 = rec[2][1]            # the oracle's dy of conv2's BatchNorm: pooled gradient, masked by [o2 > 0]
report("line 3: dx of conv3 (d o2 pooled grid)", load(3, B * 2 * 2, 384), nhwc(rec[1][1], 384))
report("line 4: dy into conv2's BatchNorm (after the pool backward, unmasked)", load(4, B * 4 * 4, 384), nhwc(rec[1][2], 384))
report("line 5: dz of conv2", load(5, B * 4 * 4, 384), nhwc(rec[2][2], 384))

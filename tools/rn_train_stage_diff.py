#!/usr/bin/env python
"""ModifiedResNet training tower: WHICH stage of the backward pass first disagrees with the oracle?  The device pass prints the sum of
squares of every dz / dx (EZCLIP_RN_DEBUG=1, csrc/resnet.hip); the oracle's explicit backward (train_step_grads_by_steps) is run with its
BatchNorm / convolution backward helpers wrapped so that it records the same quantities in the same order (c3, c2, c1, downsample per
block, last block first; then the stem).  Layout does not matter to a sum of squares (padded channels are exact zeros -- if they are not,
that shows here too)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

CFG = [((1, 1, 1, 1), 64, 128, 64, 4), ((1, 1, 1, 1), 48, 128, 64, 4), ((1, 1, 1, 1), 64, 128, 64, 3)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    from easynlp_amd import lib as L
    from easynlp_amd.appzoo.clip.rn_tower import RnEngine
    from oracle import resnet_oracle as RO
    layers, width, e, res, B = CFG[int(sys.argv[2])]
    sd = RO.make_state_dict(layers, width, e, res, 17)
    g = torch.Generator().manual_seed(6)
    px, probe = torch.randn(B, 3, res, res, generator=g), torch.randn(B, e, generator=g)
    dev = torch.device("cuda", 0)
    eng = RnEngine(layers, width, e, res, L.DTYPE_F32)
    tensors = {n: sd[n].to(dev).contiguous() for n in eng.names}
    eng.sync_train(tensors)
    out = eng.encode_image_train(px.to(dev))
    grads = {n: torch.zeros(eng.shapes[n], dtype=torch.float32, device=dev) for n in eng.names if not eng.is_statistic(n)}
    eng.backward(out, probe.to(dev), grads)
    torch.cuda.synchronize()
    sys.exit(0)

from oracle import resnet_oracle as RO  # noqa: E402

for ci, (layers, width, e, res, B) in enumerate(CFG):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(ci)], capture_output=True, text=True,
                       env=dict(os.environ, EZCLIP_RN_DEBUG="1"), timeout=600)
    dev_rows = []
    for ln in r.stderr.splitlines():
        if ln.startswith("[rn-dbg]"):
            f = ln.split()
            what = " ".join(f[2:f.index("n")])
            if what in ("dz", "dx"):
                dev_rows.append((what, float(f[-1])))
    print("== layers %s width %d res %d B %d: device rc %d, %d dz/dx stages" % (layers, width, res, B, r.returncode, len(dev_rows)))
    if r.returncode != 0:
        print(r.stderr[-1200:])
        continue
    # the oracle, double precision, same loss: sum(normalise(raw) * probe)  ->  d raw = (probe - out <out, probe>) / |raw|
    sd = {k: v.double() for k, v in RO.make_state_dict(layers, width, e, res, 17).items()}
    g = torch.Generator().manual_seed(6)
    px, probe = torch.randn(B, 3, res, res, generator=g).double(), torch.randn(B, e, generator=g).double()
    with torch.no_grad():
        raw = RO.modified_resnet_forward(sd, layers, width, px, train=True, new_stats={})
    nrm = raw.norm(dim=-1, keepdim=True)
    out = raw / nrm
    d_raw = (probe - out * (out * probe).sum(dim=-1, keepdim=True)) / nrm
    rows = []
    bn_bwd, conv_bwd = RO._bn_train_bwd, RO._conv_bwd

    def bn_wrap(dy, xh, rstd, gamma):
        res_ = bn_bwd(dy, xh, rstd, gamma)
        rows.append(("dz", float((res_[0] ** 2).sum()), tuple(dy.shape)))
        return res_

    def conv_wrap(x, w, dz, stride, padding):
        res_ = conv_bwd(x, w, dz, stride, padding)
        rows.append(("dx", float((res_[0] ** 2).sum()), tuple(res_[0].shape)))
        return res_
    RO._bn_train_bwd, RO._conv_bwd = bn_wrap, conv_wrap
    try:
        RO.train_step_grads_by_steps(sd, layers, width, px, d_raw)
    finally:
        RO._bn_train_bwd, RO._conv_bwd = bn_bwd, conv_bwd
    # the device does not form the input gradient of the stem's first convolution (its input is the pixels): drop the oracle's last dx
    j = 0
    shown = 0
    for i, (what, v, shp) in enumerate(rows):
        if j >= len(dev_rows):
            break
        dw, dv = dev_rows[j]
        if dw != what:
            if what == "dx":          # (oracle-only stage)
                continue
            print("   order mismatch at oracle %d (%s) / device %d (%s)" % (i, what, j, dw))
            break
        rel = abs(dv - v) / (abs(v) + 1e-300)
        flag = "  <-- FIRST" if rel > 1e-4 and shown == 0 else ""
        if rel > 1e-4 or i < 3:
            print("   stage %2d %-3s shape %-20s oracle %.9e device %.9e rel %.2e%s" % (i, what, shp, v, dv, rel, flag))
        if rel > 1e-4:
            shown += 1
            if shown > 6:
                break
        j += 1

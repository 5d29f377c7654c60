"""CPU restatement of the reference's ModifiedResNet image tower (TEST INFRASTRUCTURE ONLY: nothing on the product path may
import this module).

Follows easynlp/modelzoo/models/clip/modeling_chineseclip.py: ``Bottleneck`` :27-74, ``AttentionPool2d`` :77-108,
``ModifiedResNet`` :110-167 (the tower CHINESE_CLIP builds when ``vision_layers`` is a tuple, :279-287).  EVAL mode --
BatchNorm with its running statistics -- is what evaluation and prediction run and what the HIP tower implements; TRAIN mode
(``train=True``: BatchNorm normalises with the statistics of the batch and moves the running statistics towards them,
nn.BatchNorm2d defaults momentum 0.1, biased variance for the normalisation, unbiased for the running update) is the oracle-first
step of the tower's training row (no HIP kernel consumes it yet: DESIGN.md section 7).  Gradients come from torch autograd over
this restatement.  Written as plain functions over a
state dict (names as in the reference checkpoint, prefix ``visual.``) so that every step can be compared with a kernel:

    stem:   3 x [conv3x3 (the first with stride 2) -> BN -> ReLU], AvgPool2d(2)
    layerN: Bottleneck blocks: conv1x1-BN-ReLU, conv3x3-BN-ReLU, AvgPool2d(stride) (anti-aliased stride: only the first block of
            layers 2..4), conv1x1-BN, + identity (avgpool + conv1x1 + BN when the shape changes), ReLU
    attnpool: tokens = [mean over positions; the H*W positions] + positional embedding; one multi-head attention whose only
            QUERY is the mean token; output projection -> [B, output_dim]

Pinned against the real reference by tests/test_resnet_oracle.py: live (when /root/reference is importable) and through
tests/golden/rn_tiny_b3.npz / rn_tiny_train_b4.npz (generated from the reference by tools/make_golden_resnet.py: eval-mode
outputs; train-mode outputs, every parameter gradient and the updated running statistics)."""
from __future__ import annotations

import math
from typing import Dict, Sequence

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # nn.BatchNorm2d default
BN_MOMENTUM = 0.1      # nn.BatchNorm2d default


def param_shapes(layers: Sequence[int], width: int, output_dim: int, resolution: int) -> Dict[str, tuple]:
    """Names and shapes of ``ModifiedResNet(layers, output_dim, heads, resolution, width).state_dict()`` (running statistics
    included, ``num_batches_tracked`` left out), prefixed with ``visual.``."""
    s: Dict[str, tuple] = {}

    def bn(name, c):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            s["%s.%s" % (name, leaf)] = (c,)

    s["visual.conv1.weight"] = (width // 2, 3, 3, 3); bn("visual.bn1", width // 2)
    s["visual.conv2.weight"] = (width // 2, width // 2, 3, 3); bn("visual.bn2", width // 2)
    s["visual.conv3.weight"] = (width, width // 2, 3, 3); bn("visual.bn3", width)
    inplanes = width
    for li, (planes, nblocks) in enumerate(zip((width, width * 2, width * 4, width * 8), layers), start=1):
        for bi in range(nblocks):
            stride = 2 if (li > 1 and bi == 0) else 1
            p = "visual.layer%d.%d" % (li, bi)
            s[p + ".conv1.weight"] = (planes, inplanes, 1, 1); bn(p + ".bn1", planes)
            s[p + ".conv2.weight"] = (planes, planes, 3, 3); bn(p + ".bn2", planes)
            s[p + ".conv3.weight"] = (planes * 4, planes, 1, 1); bn(p + ".bn3", planes * 4)
            if stride > 1 or inplanes != planes * 4:
                s[p + ".downsample.0.weight"] = (planes * 4, inplanes, 1, 1); bn(p + ".downsample.1", planes * 4)
            inplanes = planes * 4
    e = width * 32
    sp = resolution // 32
    s["visual.attnpool.positional_embedding"] = (sp * sp + 1, e)
    for n in ("k_proj", "q_proj", "v_proj"):
        s["visual.attnpool.%s.weight" % n] = (e, e)
        s["visual.attnpool.%s.bias" % n] = (e,)
    s["visual.attnpool.c_proj.weight"] = (output_dim, e)
    s["visual.attnpool.c_proj.bias"] = (output_dim,)
    return s


def make_state_dict(layers, width, output_dim, resolution, seed=7) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic weights: every BatchNorm gets non-trivial gain, shift AND running statistics (a kernel that
    folds them wrongly cannot pass), convolutions ~ fan_in ** -0.5."""
    rs = np.random.RandomState(seed)
    sd = {}
    for name, shape in param_shapes(layers, width, output_dim, resolution).items():
        if name.endswith("running_var"):
            v = 0.5 + rs.rand(*shape)
        elif name.endswith("running_mean"):
            v = 0.2 * rs.standard_normal(shape)
        elif ".bn" in name or "downsample.1" in name:
            v = (1.0 + 0.1 * rs.standard_normal(shape)) if name.endswith("weight") else 0.05 * rs.standard_normal(shape)
        elif name.endswith("positional_embedding"):
            v = (shape[1] ** -0.5) * rs.standard_normal(shape)
        elif name.endswith(".bias"):
            v = 0.02 * rs.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = (fan_in ** -0.5) * rs.standard_normal(shape)
        sd[name] = torch.from_numpy(np.asarray(v, dtype=np.float32).copy())
    return sd


def _bn(sd, name, x, new_stats=None):
    """BatchNorm2d.  ``new_stats is None`` -- eval(): (x - running_mean) / sqrt(running_var + eps) * weight + bias, per channel.
    ``new_stats`` a dict -- train(): the mean and the BIASED variance of this batch over (N, H, W) normalise it, and the entries
    ``name.running_mean`` / ``name.running_var`` receive (1 - momentum) * running + momentum * (batch mean / UNBIASED batch variance),
    torch/nn/modules/batchnorm.py + aten batch_norm semantics."""
    if new_stats is None:
        scale = sd[name + ".weight"] / torch.sqrt(sd[name + ".running_var"] + BN_EPS)
        shift = sd[name + ".bias"] - sd[name + ".running_mean"] * scale
        return x * scale[None, :, None, None] + shift[None, :, None, None]
    n = x.shape[0] * x.shape[2] * x.shape[3]
    mean = x.mean(dim=(0, 2, 3))
    var = ((x - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
    with torch.no_grad():
        new_stats[name + ".running_mean"] = (1 - BN_MOMENTUM) * sd[name + ".running_mean"] + BN_MOMENTUM * mean.detach()
        new_stats[name + ".running_var"] = (1 - BN_MOMENTUM) * sd[name + ".running_var"] + BN_MOMENTUM * var.detach() * (n / (n - 1.0))
    xh = (x - mean[None, :, None, None]) / torch.sqrt(var + BN_EPS)[None, :, None, None]
    return xh * sd[name + ".weight"][None, :, None, None] + sd[name + ".bias"][None, :, None, None]


def _keep(x):
    return x


def bottleneck(sd, p, x, stride, new_stats=None, store=_keep):
    """Bottleneck.forward (:59-74).  ``store`` (here and below): applied to every tensor an implementation keeps in memory between two
    kernels -- each convolution output, each BatchNorm (+ identity) (+ ReLU) output, each pooled map, the attention pool's tokens /
    projections / context.  Identity in the restatement proper; the bf16 tests pass a round-to-bf16 with a straight-through gradient
    to measure what STORING activations in bf16 costs any implementation (tests/test_resnet_train_gpu.py)."""
    out = store(F.relu(_bn(sd, p + ".bn1", store(F.conv2d(x, sd[p + ".conv1.weight"])), new_stats)))
    out = store(F.relu(_bn(sd, p + ".bn2", store(F.conv2d(out, sd[p + ".conv2.weight"], padding=1)), new_stats)))
    if stride > 1:
        out = store(F.avg_pool2d(out, stride))
    out = _bn(sd, p + ".bn3", store(F.conv2d(out, sd[p + ".conv3.weight"])), new_stats)
    identity = x
    if (p + ".downsample.0.weight") in sd:
        identity = store(F.avg_pool2d(x, stride)) if stride > 1 else x
        identity = store(_bn(sd, p + ".downsample.1", store(F.conv2d(identity, sd[p + ".downsample.0.weight"])), new_stats))
    return store(F.relu(out + identity))


def attention_pool(sd, x, heads, store=_keep):
    """AttentionPool2d.forward (:87-108): only the mean token queries; scaling head_dim ** -0.5 as in
    F.multi_head_attention_forward."""
    p = "visual.attnpool."
    B, C, Hh, Ww = x.shape
    t = x.reshape(B, C, Hh * Ww).permute(0, 2, 1)                       # [B, HW, C]
    t = torch.cat([t.mean(dim=1, keepdim=True), t], dim=1)               # [B, HW + 1, C]
    t = store(t + sd[p + "positional_embedding"][None])
    q = store(t[:, :1] @ sd[p + "q_proj.weight"].t() + sd[p + "q_proj.bias"])   # [B, 1, C]
    k = store(t @ sd[p + "k_proj.weight"].t() + sd[p + "k_proj.bias"])
    v = store(t @ sd[p + "v_proj.weight"].t() + sd[p + "v_proj.bias"])
    hd = C // heads
    q = q.reshape(B, 1, heads, hd).transpose(1, 2) * (hd ** -0.5)
    k = k.reshape(B, -1, heads, hd).transpose(1, 2)
    v = v.reshape(B, -1, heads, hd).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(-1, -2), dim=-1)                   # [B, heads, 1, HW + 1]
    o = store((a @ v).transpose(1, 2).reshape(B, C))
    return o @ sd[p + "c_proj.weight"].t() + sd[p + "c_proj.bias"]


def modified_resnet_forward(sd, layers, width, x, taps=None, train=False, new_stats=None, store=_keep):
    """ModifiedResNet.forward (:153-167).  x: [B, 3, R, R] float.  ``taps``: optional dict filled with the stem output and every
    layer's output (NCHW).  ``train=True``: BatchNorm in training mode (see ``_bn``); ``new_stats`` (a dict, optional) receives the
    updated running statistics of every BatchNorm -- ``sd`` itself is never written.  ``store``: see ``bottleneck``."""
    heads = width * 32 // 64                                              # CHINESE_CLIP.__init__ :280
    ns = (new_stats if new_stats is not None else {}) if train else None
    for i, stride in ((1, 2), (2, 1), (3, 1)):
        x = store(F.relu(_bn(sd, "visual.bn%d" % i, store(F.conv2d(x, sd["visual.conv%d.weight" % i], stride=stride, padding=1)), ns)))
    x = store(F.avg_pool2d(x, 2))
    if taps is not None:
        taps["stem"] = x
    for li, nblocks in enumerate(layers, start=1):
        for bi in range(nblocks):
            x = bottleneck(sd, "visual.layer%d.%d" % (li, bi), x, 2 if (li > 1 and bi == 0) else 1, ns, store)
        if taps is not None:
            taps["layer%d" % li] = x
    return attention_pool(sd, x, heads, store)


def train_step_grads(sd, layers, width, x, probe):
    """One training-mode pass and its backward: loss = sum(features * probe) (``probe``: a fixed [B, output_dim] tensor, so that
    every output element has its own weight).  Returns (features, {parameter name: gradient} for every tensor of ``sd`` that is
    not a running statistic, {running statistic name: updated value})."""
    leaves = {k: (v.detach().clone().requires_grad_(True) if not k.endswith(("running_mean", "running_var")) else v.detach().clone())
              for k, v in sd.items()}
    new_stats = {}
    out = modified_resnet_forward(leaves, layers, width, x, train=True, new_stats=new_stats)
    (out * probe).sum().backward()
    grads = {k: v.grad.detach() for k, v in leaves.items() if v.requires_grad}
    return out.detach(), grads, new_stats


# ---- the backward pass, step by step ---------------------------------------------------------------------------------------------
# The same gradients as ``train_step_grads`` WITHOUT autograd: every step is the formula a kernel (or a GEMM call) of the HIP tower's
# training path will implement, in the decomposition it will use -- BatchNorm backward from two per-channel column moments, the
# input gradient of a 3x3 convolution as a 3x3 convolution of the output gradient with the tap-flipped, in/out-transposed weights,
# the weight gradient as im2col(x)^T . dz (one TN product), the anti-aliasing average pool as a broadcast, the attention pool as a
# one-query attention.  tests/test_resnet_oracle.py checks it against autograd of the restatement above and, through it, against
# the reference module.  (NCHW torch tensors here; the formulas do not depend on the layout.)

def _bn_train_fwd(x, gamma, beta):
    mean = x.mean(dim=(0, 2, 3))
    var = ((x - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
    rstd = 1.0 / torch.sqrt(var + BN_EPS)
    xh = (x - mean[None, :, None, None]) * rstd[None, :, None, None]
    return xh * gamma[None, :, None, None] + beta[None, :, None, None], (xh, rstd)


def _bn_train_bwd(dy, xh, rstd, gamma):
    """dgamma = sum dy * xhat, dbeta = sum dy (per channel, over N, H, W: the two column moments);
    dx = gamma * rstd * (dy - dbeta / n - xhat * dgamma / n)"""
    n = dy.shape[0] * dy.shape[2] * dy.shape[3]
    dgamma = (dy * xh).sum(dim=(0, 2, 3))
    dbeta = dy.sum(dim=(0, 2, 3))
    dx = (gamma * rstd)[None, :, None, None] * (dy - dbeta[None, :, None, None] / n - xh * dgamma[None, :, None, None] / n)
    return dx, dgamma, dbeta


def _conv_bwd(x, w, dz, stride, padding):
    """(dx, dw) of z = conv2d(x, w, stride, padding), bias-free.
    dw[co, ci, ky, kx] = sum_p dz[p, co] * x[p * stride + (ky, kx) - padding, ci]  =  dz^T . im2col(x)   (one TN product over pixels)
    dx: stride 1 -- a convolution of dz (same padding) with w flipped in both taps and transposed in (co, ci);
        stride 2 (the stem's first conv only) -- the same after dz has been spread onto the stride grid (zeros in between)."""
    B, Ci, H, W = x.shape
    Co, _, kh, kw = w.shape
    cols = F.unfold(x, (kh, kw), padding=padding, stride=stride)                    # [B, Ci*kh*kw, P]
    dzf = dz.reshape(B, Co, -1)                                                     # [B, Co, P]
    dw = torch.einsum("bop,bkp->ok", dzf, cols).reshape(Co, Ci, kh, kw)
    if stride > 1:
        up = torch.zeros(B, Co, (dz.shape[2] - 1) * stride + 1, (dz.shape[3] - 1) * stride + 1, dtype=dz.dtype)
        up[:, :, ::stride, ::stride] = dz
        # the forward read x at rows oy * stride + ky - padding: the spread gradient needs kh - 1 - padding of padding on the low side
        # and whatever restores H on the high side
        lo = kh - 1 - padding
        hi_h = H - (up.shape[2] + lo - (kh - 1))
        hi_w = W - (up.shape[3] + lo - (kw - 1))
        up = F.pad(up, (lo, hi_w, lo, hi_h))
        dx = F.conv2d(up, w.flip(2, 3).transpose(0, 1))
    else:
        dx = F.conv2d(dz, w.flip(2, 3).transpose(0, 1), padding=kh - 1 - padding)
    return dx, dw


def _avgpool_bwd(dy, s):
    """AvgPool2d(s) backward: every input pixel of an s x s window receives dy / s^2"""
    return dy.repeat_interleave(s, dim=2).repeat_interleave(s, dim=3) / float(s * s)


def _attention_pool_fwd_bwd(sd, x, heads, d_out):
    """AttentionPool2d forward + backward by hand.  Returns (out, dx, {param: grad})."""
    p = "visual.attnpool."
    B, C, Hh, Ww = x.shape
    hw = Hh * Ww
    t0 = x.reshape(B, C, hw).permute(0, 2, 1)
    tok = torch.cat([t0.mean(dim=1, keepdim=True), t0], dim=1)
    t = tok + sd[p + "positional_embedding"][None]
    Wq, Wk, Wv, Wc = (sd[p + n + ".weight"] for n in ("q_proj", "k_proj", "v_proj", "c_proj"))
    q = t[:, 0] @ Wq.t() + sd[p + "q_proj.bias"]                                     # [B, C]
    k = t @ Wk.t() + sd[p + "k_proj.bias"]                                           # [B, L, C]
    v = t @ Wv.t() + sd[p + "v_proj.bias"]
    hd = C // heads
    scale = hd ** -0.5
    qh = q.reshape(B, heads, hd) * scale
    kh = k.reshape(B, -1, heads, hd)
    vh = v.reshape(B, -1, heads, hd)
    s = torch.einsum("bhd,blhd->bhl", qh, kh)
    a = torch.softmax(s, dim=-1)
    o = torch.einsum("bhl,blhd->bhd", a, vh).reshape(B, C)
    out = o @ Wc.t() + sd[p + "c_proj.bias"]
    # ---- backward
    g = {}
    g[p + "c_proj.bias"] = d_out.sum(0)
    g[p + "c_proj.weight"] = d_out.t() @ o
    do = (d_out @ Wc).reshape(B, heads, hd)
    da = torch.einsum("bhd,blhd->bhl", do, vh)
    dvh = torch.einsum("bhl,bhd->blhd", a, do)
    ds = a * (da - (a * da).sum(-1, keepdim=True))                                   # softmax backward, one query row per head
    dqh = torch.einsum("bhl,blhd->bhd", ds, kh)
    dkh = torch.einsum("bhl,bhd->blhd", ds, qh)
    dq = (dqh * scale).reshape(B, C)
    dk = dkh.reshape(B, -1, C)
    dv = dvh.reshape(B, -1, C)
    g[p + "q_proj.bias"] = dq.sum(0); g[p + "q_proj.weight"] = dq.t() @ t[:, 0]
    g[p + "k_proj.bias"] = dk.sum((0, 1)); g[p + "k_proj.weight"] = torch.einsum("blo,bli->oi", dk, t)
    g[p + "v_proj.bias"] = dv.sum((0, 1)); g[p + "v_proj.weight"] = torch.einsum("blo,bli->oi", dv, t)
    dt = dk @ Wk + dv @ Wv
    dt[:, 0] += dq @ Wq
    g[p + "positional_embedding"] = dt.sum(0)
    dt0 = dt[:, 1:] + dt[:, :1] / hw                                                 # the mean token spreads its gradient over the positions
    dx = dt0.permute(0, 2, 1).reshape(B, C, Hh, Ww)
    return out, dx, g


def train_step_grads_by_steps(sd, layers, width, x, probe, flips=None, near_zero=None, delta=0.0):
    """``train_step_grads`` by explicit forward and backward formulas (no autograd).  Returns (features, grads) with the same keys.

    ReLU decisions (round 5).  A pre-activation within float32 rounding of zero is a coin toss for any float32 implementation: the
    device may take the other branch than this float64 evaluation, and ONE such decision moves every gradient upstream of it by
    0.3-1 % (the BatchNorm behind it spreads it over the whole channel).  ``near_zero`` (a list, filled) receives
    (site, flat index, value) of every pre-activation with |value| < ``delta``; ``flips`` (a set of (site, flat index)) inverts
    those decisions in the backward pass -- the tests accept a device gradient that matches the exact gradient of ONE of the
    decision patterns float32 cannot tell apart (tests/test_resnet_train_gpu.py)."""
    heads = width * 32 // 64
    g = {}
    tape = []                         # (kind, saved...) in forward order
    masks = {}

    def relu_site(site, pre):
        m = pre > 0
        if near_zero is not None and delta > 0:
            idx = (pre.abs() < delta).reshape(-1).nonzero().reshape(-1).tolist()
            near_zero.extend((site, i, float(pre.reshape(-1)[i])) for i in idx)
        if flips:
            mf = m.reshape(-1).clone()
            for s_, i in flips:
                if s_ == site:
                    mf[i] = ~mf[i]
            m = mf.reshape(m.shape)
        masks[site] = m
        return F.relu(pre)

    def conv_bn(x_in, conv, bn, stride, padding, relu, residual=None):
        z = F.conv2d(x_in, sd[conv + ".weight"], stride=stride, padding=padding)
        y, (xh, rstd) = _bn_train_fwd(z, sd[bn + ".weight"], sd[bn + ".bias"])
        if residual is not None:
            y = y + residual
        if relu:
            y = relu_site(conv, y)
        tape.append(("conv_bn", conv, bn, stride, padding, relu, x_in, xh, rstd, y))
        return y

    with torch.no_grad():
        h = x
        for i, stride in ((1, 2), (2, 1), (3, 1)):
            h = conv_bn(h, "visual.conv%d" % i, "visual.bn%d" % i, stride, 1, True)
        h = F.avg_pool2d(h, 2); tape.append(("pool", 2))
        blocks = []
        for li, nblocks in enumerate(layers, start=1):
            for bi in range(nblocks):
                blocks.append(("visual.layer%d.%d" % (li, bi), 2 if (li > 1 and bi == 0) else 1))
        # the blocks are unrolled by hand below (two branches meet at the residual add), so their tape is a list of dicts
        btape = []
        for p_, stride in blocks:
            rec = {"p": p_, "stride": stride, "x": h}
            t_ = []
            def cb(x_in, conv, bn, padding, relu):
                z = F.conv2d(x_in, sd[conv + ".weight"], padding=padding)
                y, (xh, rstd) = _bn_train_fwd(z, sd[bn + ".weight"], sd[bn + ".bias"])
                t_.append((conv, bn, padding, x_in, xh, rstd))
                return relu_site(conv, y) if relu else y
            o1 = cb(h, p_ + ".conv1", p_ + ".bn1", 0, True)
            o2 = cb(o1, p_ + ".conv2", p_ + ".bn2", 1, True)
            o2p = F.avg_pool2d(o2, stride) if stride > 1 else o2
            o3 = cb(o2p, p_ + ".conv3", p_ + ".bn3", 0, False)
            rec.update(o1=o1, o2=o2, main=t_[:])
            if (p_ + ".downsample.0.weight") in sd:
                xi = F.avg_pool2d(h, stride) if stride > 1 else h
                t_.clear()
                ident = cb(xi, p_ + ".downsample.0", p_ + ".downsample.1", 0, False)
                rec["down"] = t_[0]
            else:
                ident = h
            h = relu_site(p_ + ".out", o3 + ident)
            rec["y"] = h
            btape.append(rec)
        out, dx, ga = _attention_pool_fwd_bwd(sd, h, heads, probe)
        g.update(ga)

        def cb_bwd(dy, rec_):
            conv, bn, padding, x_in, xh, rstd = rec_
            dz, g[bn + ".weight"], g[bn + ".bias"] = _bn_train_bwd(dy, xh, rstd, sd[bn + ".weight"])
            dxi, g[conv + ".weight"] = _conv_bwd(x_in, sd[conv + ".weight"], dz, 1, padding)
            return dxi

        for rec in reversed(btape):
            dsum = dx * masks[rec["p"] + ".out"]                           # ReLU after the residual add
            stride = rec["stride"]
            m1, m2, m3 = rec["main"]
            d = cb_bwd(dsum, m3)
            if stride > 1:
                d = _avgpool_bwd(d, stride)
            d = d * masks[rec["p"] + ".conv2"]
            d = cb_bwd(d, m2)
            d = d * masks[rec["p"] + ".conv1"]
            d = cb_bwd(d, m1)
            if "down" in rec:
                di = cb_bwd(dsum, rec["down"])
                if stride > 1:
                    di = _avgpool_bwd(di, stride)
            else:
                di = dsum
            dx = d + di
        dx = _avgpool_bwd(dx, 2)
        for kind, conv, bn, stride, padding, relu, x_in, xh, rstd, y in reversed([t for t in tape if t[0] == "conv_bn"]):
            dx = dx * masks[conv]
            dz, g[bn + ".weight"], g[bn + ".bias"] = _bn_train_bwd(dx, xh, rstd, sd[bn + ".weight"])
            dx, g[conv + ".weight"] = _conv_bwd(x_in, sd[conv + ".weight"], dz, stride, padding)
    return out, g


# ---- the same steps in the HIP tower's data layout -----------------------------------------------------------------------------------
# NHWC activations [B * H * W, Cp] (channels zero-padded to Cp), packed weights [Opad][9 * Cp] with K index (ky * 3 + kx) * Cp + c
# (csrc/resnet.hip: rn_pack_conv_kernel), the implicit 3x3 convolution of csrc/gemm.hip (GemmArgs::conv_*: output row m = pixel
# (b, y, x), the A tile of tap (ky, kx) is the channel slice of the pixel shifted by (ky - 1, kx - 1), zeros outside the image).
# These pin the index conventions of the training path's packing kernels before any of them exists:
#   input gradient  = the same implicit convolution of dz with  Wd[c][(ky * 3 + kx) * Opad + o] = W[o][c][2 - ky][2 - kx]
#   weight gradient = dz^T . im2col(x)  ->  dWp[o][(ky * 3 + kx) * Cp + c]  (the packed layout again), unpacked to [O, I, 3, 3]

def to_nhwc(x, cp):
    """[B, C, H, W] -> [B * H * W, cp] (zero-padded channels)"""
    B, C, H, W = x.shape
    out = torch.zeros(B * H * W, cp, dtype=x.dtype)
    out[:, :C] = x.permute(0, 2, 3, 1).reshape(B * H * W, C)
    return out


def from_nhwc(m, B, C, H, W):
    return m[:, :C].reshape(B, H, W, C).permute(0, 3, 1, 2)


def pack_conv3x3(w, cp, opad):
    """rn_pack_conv_kernel without a BatchNorm fold: [O, I, 3, 3] -> [opad, 9 * cp], K index (ky * 3 + kx) * cp + c"""
    O, I = w.shape[:2]
    p = torch.zeros(opad, 9, cp, dtype=w.dtype)
    p[:O, :, :I] = w.reshape(O, I, 9).permute(0, 2, 1)
    return p.reshape(opad, 9 * cp)


def pack_conv3x3_dgrad(w, cp, opad):
    """weights of the input-gradient convolution: [cp, 9 * opad] with Wd[c][(ky * 3 + kx) * opad + o] = W[o][c][2 - ky][2 - kx]"""
    O, I = w.shape[:2]
    p = torch.zeros(cp, 9, opad, dtype=w.dtype)
    p[:I, :, :O] = w.flip(2, 3).reshape(O, I, 9).permute(1, 2, 0)
    return p.reshape(cp, 9 * opad)


def im2col3x3_nhwc(a, B, H, W):
    """[B * H * W, cp] -> [B * H * W, 9 * cp]: column block (ky * 3 + kx) holds the pixel shifted by (ky - 1, kx - 1), zeros outside"""
    cp = a.shape[1]
    img = a.reshape(B, H, W, cp)
    pad = F.pad(img, (0, 0, 1, 1, 1, 1))
    cols = [pad[:, ky:ky + H, kx:kx + W, :] for ky in range(3) for kx in range(3)]
    return torch.cat(cols, dim=-1).reshape(B * H * W, 9 * cp)


def implicit_conv3x3_nhwc(a, wp, B, H, W):
    """what gemm_nt computes in convolution mode: out[m][o] = sum_k im2col(a)[m][k] * wp[o][k]"""
    return im2col3x3_nhwc(a, B, H, W) @ wp.t()


def unpack_wgrad3x3(dwp, O, I, cp):
    """[opad, 9 * cp] (the packed K order) -> [O, I, 3, 3]"""
    return dwp.reshape(dwp.shape[0], 9, cp)[:O, :, :I].permute(0, 2, 1).reshape(O, I, 3, 3)

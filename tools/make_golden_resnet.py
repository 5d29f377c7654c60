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


# ReLU decisions and seeds (round 6).  A pre-activation within float32 rounding (~1e-6 after a few layers) of zero is a coin toss for
# ANY float32 implementation -- the reference's own CPU evaluation included -- and one decision taken the other way moves every gradient
# upstream by 0.3-1 % (the BatchNorm behind it spreads it over the channel: profiles/r5_rn_train_where.log).  With N pre-activations of
# roughly unit scale about 0.8 * N * d of them lie within d of zero: 3 x 10^5 of them (this tower, 4 images of 64 x 64) put ~25 inside
# 1e-4 whatever the seed, but ~2.5 inside 1e-5 -- so the INPUT seed of every training fixture is chosen (``margin`` below, float64
# oracle) such that NO pre-activation lies within MARGIN = 1e-5 of zero, ten times the float32 noise.  The tests then assert the
# gradients at 1e-4 with no allowance for decisions.  ``python tools/make_golden_resnet.py --search`` repeats the search.
MARGIN = 1e-5
TRAIN_CFG = dict(layers=(1, 2, 1, 1), width=16, output_dim=24, resolution=64, wseed=9, iseed=18, batch=4)


def margin_of(sd, layers, width, px, delta=MARGIN):
    """(number of ReLU pre-activations of the training pass over ``px`` with |value| < delta, the smallest of them), float64.  The
    pre-activations belong to the FORWARD pass: the loss behind the tower does not enter."""
    sd64 = {k: v.double() for k, v in sd.items() if k.startswith("visual.")}
    e = sd64["visual.attnpool.c_proj.weight"].shape[0]
    near = []
    RO.train_step_grads_by_steps(sd64, tuple(layers), width, px.double(), torch.zeros(px.shape[0], e, dtype=torch.float64), near_zero=near, delta=delta)
    return len(near), (min(abs(v) for _, _, v in near) if near else delta)


def tower_inputs(c, iseed=None):
    g = torch.Generator().manual_seed(c["iseed"] if iseed is None else iseed)
    px = torch.randn(c["batch"], 3, c["resolution"], c["resolution"], generator=g)
    probe = torch.randn(c["batch"], c["output_dim"], generator=g)
    return px, probe


def search_input_seed(c, first=0, last=4000, delta=MARGIN):
    """the first input seed of a tower fixture with no pre-activation within ``delta`` of zero"""
    sd = RO.make_state_dict(c["layers"], c["width"], c["output_dim"], c["resolution"], c["wseed"])
    for iseed in range(first, last):
        n, mn = margin_of(sd, c["layers"], c["width"], tower_inputs(c, iseed)[0], delta)
        if n == 0:
            return iseed
    raise RuntimeError("no input seed in [%d, %d) keeps every pre-activation %g away from zero" % (first, last, delta))


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
    px, probe = tower_inputs(c)
    n_near, smallest = margin_of(sd, c["layers"], c["width"], px)
    assert n_near == 0, "TRAIN_CFG's input seed leaves %d pre-activations within %g of zero (smallest %g): run --search" % (n_near, MARGIN, smallest)
    c = dict(c, relu_margin=MARGIN)
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


# ---- a fixture at which bf16 means something ----------------------------------------------------------------------------------------
# Width 64 (the RN50 family's: no channel padding), 32 images of 64 x 64: every BatchNorm averages over 128 ... 32768 values, so the
# rounding of a bf16 pipeline is noise on top of a well-defined gradient (at batch 4 / width 16 it dominates: torch's own bfloat16
# evaluation is off by ~0.5).  The REFERENCE module is evaluated in float64 here (``module.double()``: the reference's code, no
# float32 coin tosses at the ReLUs), so the float64 oracle must reproduce it to ~1e-9 and the bf16 device path is read against it.
# bn3_gain: the gain of every block's LAST BatchNorm is scaled by it (the reference initialises those gains to ZERO,
# CHINESE_CLIP.initialize_parameters :323-334; a briefly trained tower has small ones).  At gain ~1 a random-init BatchNorm-ReLU tower is
# in its chaotic regime (perturbations grow from layer to layer): the activation-rounding floor of the gradients is then 0.6, at 0.2 it is 0.24.
W64_CFG = dict(layers=(2, 2, 2, 2), width=64, output_dim=128, resolution=64, wseed=23, iseed=2, batch=32, bn3_gain=0.2)


def w64_state_dict(c):
    sd = RO.make_state_dict(c["layers"], c["width"], c["output_dim"], c["resolution"], c["wseed"])
    for k in sd:
        if k.endswith("bn3.weight") and ".layer" in k:
            sd[k] = sd[k] * c["bn3_gain"]
    return sd


def w64_inputs(c):
    rs = np.random.RandomState(c["iseed"])                      # (numpy streams are frozen across versions: the pixels are not stored)
    px = rs.standard_normal((c["batch"], 3, c["resolution"], c["resolution"])).astype(np.float32)
    probe = rs.standard_normal((c["batch"], c["output_dim"])).astype(np.float32)
    return torch.from_numpy(px), torch.from_numpy(probe)


def main_train_w64():
    c = W64_CFG
    sd = w64_state_dict(c)
    px, probe = w64_inputs(c)
    m = reference_tower(c, sd).double().train()
    raw = m(px.double())
    out = raw / raw.norm(dim=-1, keepdim=True)                      # CHINESE_CLIP.forward normalises (modeling_chineseclip.py:360)
    (out * probe.double()).sum().backward()
    arrays = dict(image_features=raw.detach().numpy(), meta=np.frombuffer(json.dumps(dict(c, precision="float64", loss="sum(normalise(features) * probe)")).encode(), dtype=np.uint8))
    for k, p in m.named_parameters():
        k = "visual." + k
        g = p.grad.detach()
        idx = sample_index(k, g.numel(), min(64, g.numel()))
        arrays["gnorm:" + k] = np.float64(g.norm().item())
        arrays["gsamp:" + k] = g.reshape(-1)[torch.from_numpy(idx)].numpy()
    for k, v in m.state_dict().items():
        if k.endswith(("running_mean", "running_var")):
            arrays["stat:visual." + k] = v.detach().numpy()
    path = os.path.join(ROOT, "tests", "golden", "rn_w64_train_b32.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, raw.shape, os.path.getsize(path), "bytes")


# ---- the WHOLE model with this tower, in train() mode ---------------------------------------------------------------------------------
# CHINESE_CLIP(vision_layers=(1, 2, 1, 1), ...) builds ModifiedResNet (modeling_chineseclip.py:279-287); core/trainer.py:658-661 trains it
# through CLIPApp.forward / compute_loss.  Fixture: loss, embeddings, logits, the gradient of EVERY parameter (both towers, logit_scale),
# every running statistic the forward moved and the num_batches_tracked counters, from the REFERENCE module in train() mode.
CLIP_RN_CFG = dict(model_type="chinese_clip", embed_dim=64, image_resolution=64, vision_layers=[1, 2, 1, 1], vision_width=16, vision_patch_size=16,
                   vocab_size=211, text_attention_probs_dropout_prob=0.0, text_hidden_act="gelu", text_hidden_dropout_prob=0.0,
                   text_hidden_size=128, text_initializer_range=0.02, text_intermediate_size=512, text_max_position_embeddings=64,
                   text_num_attention_heads=2, text_num_hidden_layers=2, text_type_vocab_size=2)
CLIP_RN_CASE = dict(batch=6, seq_len=24, wseed=21, rn_wseed=13, iseed=302)


def clip_rn_state_dict(cfg, case):
    """text tower / projections / logit_scale from oracle.clip_oracle's generator, the ModifiedResNet tower from oracle.resnet_oracle's"""
    from oracle import clip_oracle as O
    vit_like = dict(cfg, vision_layers=1, vision_width=64)                                  # (only its non-visual entries are kept)
    sd = {k: v for k, v in O.make_state_dict(vit_like, case["wseed"]).items() if not k.startswith("visual.")}
    sd.update(RO.make_state_dict(tuple(cfg["vision_layers"]), cfg["vision_width"], cfg["embed_dim"], cfg["image_resolution"], case["rn_wseed"]))
    return sd


def clip_rn_inputs(cfg, case, iseed=None):
    from oracle import clip_oracle as O
    return O.make_inputs(cfg, case["batch"], case["seq_len"], case["iseed"] if iseed is None else iseed)


def main_clip_train():
    cfg, case = CLIP_RN_CFG, CLIP_RN_CASE
    sd = clip_rn_state_dict(cfg, case)
    px, ids = clip_rn_inputs(cfg, case)
    n_near, smallest = margin_of(sd, cfg["vision_layers"], cfg["vision_width"], px)
    assert n_near == 0, "CLIP_RN_CASE's input seed leaves %d pre-activations within %g of zero (smallest %g): run --search" % (n_near, MARGIN, smallest)
    torch.manual_seed(0)
    model = R.reference_chinese_clip(cfg, sd)
    model.train()                                                   # (dropout probabilities are 0 in this config: BatchNorm is what changes)
    img, txt = model(px, ids)                                       # modeling_chineseclip.py:352-365
    lpt = torch.matmul(txt, img.t()) * model.logit_scale.exp()      # appzoo/clip/model.py:148
    ar = torch.arange(case["batch"])
    loss = (torch.nn.functional.cross_entropy(lpt, ar) + torch.nn.functional.cross_entropy(lpt.T, ar)) / 2.0      # model.py:154-160
    loss.backward()
    out = {"meta": np.frombuffer(json.dumps(dict(cfg=cfg, case=case, relu_margin=MARGIN, torch=torch.__version__)).encode(), dtype=np.uint8),
           "image_embeds": img.detach().numpy(), "text_embeds": txt.detach().numpy(), "logits_per_text": lpt.detach().numpy(),
           "loss": np.float32(loss.item())}
    n_grads = 0
    for n, p in model.named_parameters():
        if p.grad is None:
            out["nograd/" + n] = np.zeros(0, np.float32)
        elif p.numel() <= 8192:
            out["grad/" + n] = p.grad.numpy(); n_grads += 1
        else:
            idx = sample_index(n, p.numel())
            out["gnorm/" + n] = np.float64(p.grad.double().norm().item())
            out["gsamp/" + n] = p.grad.reshape(-1)[torch.from_numpy(idx)].numpy(); n_grads += 1
    n_stats = 0
    for n, b in model.state_dict().items():
        if n.endswith(("running_mean", "running_var", "num_batches_tracked")):
            out["stat/" + n] = b.detach().numpy(); n_stats += 1
    path = os.path.join(ROOT, "tests", "golden", "clip_rn_tiny_train_b6_l24.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "loss", loss.item(), n_grads, "gradients", n_stats, "statistics / counters", os.path.getsize(path), "bytes")


def search():
    print("TRAIN_CFG iseed ->", search_input_seed(TRAIN_CFG))
    cfg, case = CLIP_RN_CFG, CLIP_RN_CASE
    sd = clip_rn_state_dict(cfg, case)
    for iseed in range(4000):
        n, mn = margin_of(sd, cfg["vision_layers"], cfg["vision_width"], clip_rn_inputs(cfg, case, iseed)[0])
        if n == 0:
            print("CLIP_RN_CASE iseed ->", iseed)
            break


if __name__ == "__main__":
    if "--search" in sys.argv:
        search()
    else:
        main()
        main_train()
        main_train_w64()
        main_clip_train()

#!/usr/bin/env python
"""Soak run of the training path at the headline size: ViT-B/16 + BERT-base, bf16, 1 024 pairs per step, four fixed synthetic batches
visited round-robin, the fused step (forward + InfoNCE + backward) + gradient clipping at 1.0 (core/trainer.py:315-325) + AdamW
(lr 5e-5, eps 1e-6, weight decay 0.01: optimizers.py:381-466) + the re-pack of the library's weight copies, with the text tower's
train-mode dropout (0.1) on.  Prints the loss every 25 steps; asserts that every loss is finite and that the model fits the 4 096
fixed pairs (mean loss of the last 25 steps well under the first's).  usage: soak_train.py [steps] [dropout] [fused|loop]
`loop`: the optimizer as the reference's Trainer runs it -- a Python loop over the ~400 parameters with half a dozen elementwise
launches each (the arithmetic of optimizers.py:381-466 restated: moments, bias correction, addcdiv, decoupled decay through
`p.data`) -- instead of torch's fused AdamW: what the caller's optimizer costs per step next to a 137 ms training step."""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                   # noqa: E402
import bench                                   # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dropout = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
opt_kind = sys.argv[3] if len(sys.argv) > 3 else "fused"
dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS["bf16_b1024_train_opt"])
app, name = bench.build_app(wl, dev, text_dropout=dropout)
app.train()
B, S = wl["batch"], wl["seq"]
batches = [bench.synth_batch(B, S, bench.VITB16_BERTBASE["vocab_size"], dev, seed=1000 + 97 * k) for k in range(4)]
params = [p for p in app.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=5e-5, eps=1e-6, weight_decay=0.01, fused=True)


class LoopAdamW:
    """Per-parameter AdamW written the way the reference's optimizer is (one parameter at a time, updates through p.data)."""

    def __init__(self, named, lr, eps, weight_decay, betas=(0.9, 0.999)):
        no_decay = ("bias", "LayerNorm.bias", "LayerNorm.weight")               # optimizers.py:490
        self.items = [(p, 0.0 if any(nd in n for nd in no_decay) else weight_decay) for n, p in named if p.requires_grad]
        self.lr, self.eps, self.betas, self.t = lr, eps, betas, 0
        self.m = [torch.zeros_like(p) for p, _ in self.items]
        self.v = [torch.zeros_like(p) for p, _ in self.items]

    @torch.no_grad()
    def step(self):
        self.t += 1
        b1, b2 = self.betas
        step_size = self.lr * math.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
        for (p, wd), m, v in zip(self.items, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad
            m.mul_(b1).add_(g, alpha=1.0 - b1)
            v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
            p.data.addcdiv_(m, v.sqrt().add_(self.eps), value=-step_size)
            if wd > 0.0:
                p.data.add_(p.data, alpha=-self.lr * wd)


if opt_kind == "loop":
    opt = LoopAdamW(list(app.named_parameters()), lr=5e-5, eps=1e-6, weight_decay=0.01)
losses, norms = [], []
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(steps):
    px, ids = batches[it % 4]
    loss = app.contrastive_step(px, ids, process_group=False, backward=True, zero_grad=True)
    norms.append(torch.nn.utils.clip_grad_norm_(params, 1.0))
    opt.step()
    app._engine.mark_weights_dirty()
    losses.append(loss.detach())
torch.cuda.synchronize()
dt = time.perf_counter() - t0
losses = [float(x) for x in torch.stack(losses).cpu()]
norms = [float(x) for x in torch.stack(norms).cpu()]
print("%s, dropout %.2f, optimizer %s: %d steps of 1024 pairs in %.1f s = %.0f pairs/s, %.1f ms per step (step + clip + AdamW + re-pack)"
      % (name, dropout, opt_kind, steps, dt, steps * B / dt, dt / steps * 1e3))
for i in range(0, steps, 25):
    seg, ng = losses[i:i + 25], norms[i:i + 25]
    print("steps %4d..%4d  mean loss %.4f  (min %.4f max %.4f)  mean |grad| before clipping %.3f" % (i, i + len(seg) - 1, sum(seg) / len(seg), min(seg), max(seg), sum(ng) / len(ng)))
assert all(math.isfinite(x) for x in losses) and all(math.isfinite(x) for x in norms)
first, last = sum(losses[:25]) / 25, sum(losses[-25:]) / 25
print("first 25: %.4f   last 25: %.4f   ln(1024) = %.4f" % (first, last, math.log(1024)))
assert last < first - 0.5, (first, last)
print("soak OK")

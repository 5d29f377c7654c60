"""contrastive_scope='global' on the autograd path, single GPU: with one rank the global batch IS the local
one, so the loss and every gradient must equal the default ('local') path -- which exercises the fused shard kernel, the
autograd plumbing and compute_loss's 'loss' shortcut.  The two-rank collective logic and the DDP-averaging semantics are
covered on the CPU (tests/test_distributed.py, gloo).  Ordered last: written after the round's GPU minutes were spent."""
import pytest
import torch

from easynlp_amd.appzoo.clip import CLIPApp
from oracle import clip_oracle as O
from oracle import ref_harness as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_global_scope_equals_local_scope_on_one_rank(tmp_path, dtype):
    cfg = O.CONFIGS["small"]
    R.write_checkpoint_dir(str(tmp_path), cfg, O.make_state_dict(cfg, 5))
    px, ids = O.make_inputs(cfg, 6, 24, 3)
    grads, losses = {}, {}
    for scope in ("local", "global"):
        app = CLIPApp(str(tmp_path), user_defined_parameters={"clip_compute_dtype": dtype, "contrastive_scope": scope}).cuda()
        app.train()
        out = app({"pixel_values": px.clone(), "input_ids": ids.clone()})
        if scope == "global":
            # the loss comes with the exchange; the logits are this rank's own block, detached (for logging: model.py:148)
            assert out["loss"] is not None and tuple(out["logits_per_text"].shape) == (6, 6) and not out["logits_per_text"].requires_grad
            assert torch.equal(out["logits_per_image"], out["logits_per_text"].T)
        loss = app.compute_loss(out, [])["loss"]
        loss.backward()
        losses[scope] = loss.item()
        grads[scope] = {n: p.grad.clone() for n, p in app.named_parameters() if p.grad is not None}
        app.eval()                                      # eval keeps the reference contract: logits, no exchange
        with torch.no_grad():
            ev = app({"pixel_values": px.clone(), "input_ids": ids.clone()})
        assert tuple(ev["logits_per_text"].shape) == (6, 6)
    # fp32: both scopes evaluate the loss in exact float32.  bf16: the global scope runs the tiled contrastive kernels (split bf16
    # operands at this size: float32-class loss; its embedding gradients then pass through the bf16 towers' backward)
    assert abs(losses["local"] - losses["global"]) < (1e-5 if dtype == "fp32" else 5e-5)
    assert set(grads["local"]) == set(grads["global"])
    floor = 1e-3 * max(float(g.norm()) for g in grads["local"].values())
    for n, g in grads["local"].items():
        if dtype == "fp32":
            assert float((grads["global"][n] - g).norm()) <= 2e-4 * float(g.norm()) + 1e-7, n
        else:
            assert float((grads["global"][n] - g).norm()) <= 2e-2 * float(g.norm()) + floor, n
    with pytest.raises(Exception):
        CLIPApp(str(tmp_path), user_defined_parameters={"contrastive_scope": "everything"})

"""Host-side state of the drop-in on the GPU: when the packed weights are refreshed, who owns a saved workspace, which
gradients exist, the two-stream tower schedule, the overlapped gradient reduction.

The reference's own optimizers (the CLI default `AdamW` and `BertAdam`, easynlp/core/optimizers.py:367,451,462) update
through ``p.data`` -- which leaves ``p._version`` alone.  The checkout is not on the GPU box, so `DataStepSGD` below
updates the same way; the CPU suite runs the real classes against the same engine logic (tests/test_engine_state.py)."""
import os

import pytest
import torch

from easynlp_amd import lib as L
from easynlp_amd import parallel as P
from easynlp_amd.appzoo.clip import CLIPApp
from oracle import clip_oracle as O
from oracle import ref_harness as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


class DataStepSGD(torch.optim.Optimizer):
    """p.data.add_(-lr * grad): the update style of easynlp/core/optimizers.py (no version-counter bump on p)."""

    def __init__(self, params, lr):
        super().__init__(params, dict(lr=lr))

    def step(self):
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    p.data.add_(p.grad.data, alpha=-g["lr"])


def floor_of(grads):
    """absolute floor for gradient comparisons: BERT key-bias gradients are mathematically zero (softmax does not see a
    per-query constant), i.e. pure rounding noise ~1e-10 that no relative bound applies to"""
    return 1e-6 * max(float(g.norm()) for g in grads.values())


def make_app(tmp_path, dtype, cfg_name="small", seed=3):
    cfg = O.CONFIGS[cfg_name]
    sd = O.make_state_dict(cfg, seed)
    R.write_checkpoint_dir(str(tmp_path), cfg, sd)
    return CLIPApp(str(tmp_path), user_defined_parameters={"clip_compute_dtype": dtype}).cuda(), cfg, sd


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_p_data_updates_reach_the_packed_weights(tmp_path, dtype):
    app, cfg, sd = make_app(tmp_path, dtype)
    app.train()
    px, ids = O.make_inputs(cfg, 6, 24, 1)
    opt = DataStepSGD(app.parameters(), lr=0.05)      # (the oracle's own SGD at this rate: 1.94 -> 0.75 -> 0.37)
    versions = [p._version for p in app.parameters()]
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = app.compute_loss(app({"pixel_values": px, "input_ids": ids}), [])["loss"]
        loss.backward()
        opt.step()
        losses.append(float(loss.item()))
    assert [p._version for p in app.parameters()] == versions, "the optimizer was expected to bypass the version counters"
    assert losses[1] < losses[0] - 0.3 and losses[2] < losses[1] - 0.1, losses      # it learns: the kernels saw the new weights
    # ... and the forward equals the oracle's on the UPDATED weights (bf16 GEMM weights are packed copies; in fp32 the
    # transposed backward copies are)
    app.eval()
    new_sd = {k: v.detach().cpu() for k, v in app._params.items()}
    with torch.no_grad():
        out = app({"pixel_values": px, "input_ids": ids})
        ref = O.clip_forward(new_sd, cfg, px, ids)
        old = O.clip_forward(sd, cfg, px, ids)
    tol = 1e-4 if dtype == "fp32" else 1e-2
    assert float((old["image_embeds"] - ref["image_embeds"]).abs().max()) > 0.1      # (stale weights would miss by this much)
    assert float((out["image_embeds"].cpu() - ref["image_embeds"]).abs().max()) < tol
    assert float((out["text_embeds"].cpu() - ref["text_embeds"]).abs().max()) < tol
    # raw writes without any backward in between: mark_weights_dirty()
    with torch.no_grad():
        before = app({"input_ids": ids}, feat=True)["text_embeds"].clone()
        app._params["text_projection"].data.mul_(-1.0)
        stale = app({"input_ids": ids}, feat=True)["text_embeds"]
        if dtype == "bf16":
            assert torch.equal(stale, before)          # (documented: torch cannot see the write)
        app._engine.mark_weights_dirty()
        fresh = app({"input_ids": ids}, feat=True)["text_embeds"]
    assert float((fresh + before).abs().max()) < 1e-6


def test_two_forwards_before_backward_keep_their_own_activations(tmp_path):
    """Two micro-batches whose losses are summed: the first forward's saved workspace must survive the second forward."""
    app, cfg, sd = make_app(tmp_path, "fp32")
    app.eval()
    pa, ia = O.make_inputs(cfg, 5, 24, 1)
    pb, ib = O.make_inputs(cfg, 5, 24, 2)

    def grads_of(run):
        for p in app.parameters():
            p.grad = None
        run()
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in app._params.items() if p.grad is not None}

    def loss_of(px, ids):
        return app.compute_loss(app({"pixel_values": px, "input_ids": ids}), [])["loss"]

    both = grads_of(lambda: (loss_of(pa, ia) + loss_of(pb, ib)).backward())
    ga = grads_of(lambda: loss_of(pa, ia).backward())
    gb = grads_of(lambda: loss_of(pb, ib).backward())
    assert set(both) == set(ga) == set(gb)
    fl = floor_of(ga)
    for n in both:
        want = ga[n] + gb[n]
        assert float((both[n] - want).norm()) <= 1e-5 * float(want.norm()) + fl, n
    # a grad-enabled feature call between forward and backward does not disturb the pending backward either
    def interleaved():
        loss = loss_of(pa, ia)
        app({"pixel_values": pb, "input_ids": ib}, feat=True)
        loss.backward()
    gi = grads_of(interleaved)
    for n in ga:
        assert float((gi[n] - ga[n]).norm()) <= 1e-5 * float(ga[n].norm()) + fl, n


def test_gradients_that_stay_none_and_gradient_accumulation(tmp_path):
    app, cfg, sd = make_app(tmp_path, "fp32")
    app.eval()
    px, ids = O.make_inputs(cfg, 5, 24, 1)
    _, _, ref_g = O.forward_loss_backward(sd, cfg, px, ids)
    for p in app.parameters():
        p.grad = None
    app.compute_loss(app({"pixel_values": px, "input_ids": ids}), [])["loss"].backward()
    none = sorted(n for n, p in app._params.items() if p.grad is None)
    assert none == ["bert.pooler.dense.bias", "bert.pooler.dense.weight"], none      # as in the reference
    first = {n: p.grad.detach().clone() for n, p in app._params.items() if p.grad is not None}
    fl = floor_of(first)
    for n, g in first.items():
        r = ref_g[n].to(g.device)
        assert float((g - r).norm()) <= 2e-4 * float(r.norm()) + fl, n
    # accumulate a second backward into the live gradients (no zero_grad): autograd adds a separate buffer
    app.compute_loss(app({"pixel_values": px, "input_ids": ids}), [])["loss"].backward()
    for n, g in first.items():
        assert float((app._params[n].grad - 2 * g).norm()) <= 1e-5 * float(g.norm()) + fl, n
    # text only: the image tower's parameters (and logit_scale) get no gradient at all
    for p in app.parameters():
        p.grad = None
    emb = app({"input_ids": ids}, feat=True)["text_embeds"]
    (emb * torch.linspace(-1, 1, emb.shape[1], device=emb.device)).sum().backward()
    assert all(p.grad is None for n, p in app._params.items() if n.startswith("visual.") or n == "logit_scale")
    assert float(app._params["text_projection"].grad.abs().max()) > 0
    # the fused step: same rule, gradients are views of ONE arena in completion order
    for p in app.parameters():
        p.grad = None
    app.contrastive_step(px.cuda(), ids.cuda(), process_group=False, backward=True, zero_grad=True)
    torch.cuda.synchronize()
    arena = app._engine.grad_arena("step", px.cuda().device)
    assert app._params["bert.pooler.dense.weight"].grad is None
    for n, g in first.items():
        assert arena.owns(n, app._params[n].grad), n
        assert float((app._params[n].grad - g).norm()) <= 2e-5 * float(g.norm()) + fl, n
    order = [arena.group_range[g][0] for g in arena.group_order]
    assert order == sorted(order) and arena.group_order[0] == (2, P.STAGE_HEAD) and arena.group_order[1] == (0, P.STAGE_HEAD)


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_two_streams_equal_one_stream(tmp_path, dtype):
    """Image tower on the current stream, text tower on the side stream: bit-identical to running them back to back."""
    app, cfg, sd = make_app(tmp_path, dtype)
    app.eval()
    px, ids = O.make_inputs(cfg, 7, 24, 4)
    px, ids = px.cuda(), ids.cuda()
    res = {}
    for two in (False, True):
        app.two_streams = two
        for p in app.parameters():
            p.grad = None
        with torch.no_grad():
            out = app({"pixel_values": px, "input_ids": ids})
        loss = app.contrastive_step(px, ids, process_group=False, backward=True, zero_grad=True)
        torch.cuda.synchronize()
        res[two] = (out["image_embeds"].clone(), out["text_embeds"].clone(), float(loss.item()),
                    {n: p.grad.detach().clone() for n, p in app._params.items() if p.grad is not None})
    assert torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])
    assert res[False][2] == res[True][2]
    fl = floor_of(res[False][3])
    for n, g in res[False][3].items():
        # (weight gradients are bit-reproducible; LayerNorm / bias column sums use float atomics)
        assert float((g - res[True][3][n]).norm()) <= 1e-5 * float(g.norm()) + fl, n


def test_overlapped_gradient_reduction_single_rank_rccl(tmp_path):
    """The progress hook -> bucketed asynchronous all-reduce path on real hardware (RCCL, one rank: the sum is the
    identity, the machinery -- ctypes callback from inside ezclip_backward_*, buckets on the collective stream, the final
    wait -- is the multi-GPU one)."""
    import torch.distributed as dist
    own = not dist.is_initialized()
    if own:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29613")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        app, cfg, sd = make_app(tmp_path, "fp32")
        app.eval()
        px, ids = O.make_inputs(cfg, 5, 24, 1)
        px, ids = px.cuda(), ids.cuda()
        for p in app.parameters():
            p.grad = None
        app.contrastive_step(px, ids, process_group=True, backward=True, zero_grad=True)
        torch.cuda.synchronize()
        plain = {n: p.grad.detach().clone() for n, p in app._params.items() if p.grad is not None}
        seen = []
        app._engine.set_progress_hook(lambda t, s: seen.append((t, s)))
        app.contrastive_step(px, ids, process_group=True, backward=True, zero_grad=True)
        app._engine.set_progress_hook(None)
        nl_v, nl_t = cfg["vision_layers"], cfg["text_num_hidden_layers"]
        assert seen == [(0, L.STAGE_HEAD)] + [(0, i) for i in range(nl_v - 1, -1, -1)] + [(0, L.STAGE_EMBED)] + \
                       [(1, L.STAGE_HEAD)] + [(1, i) for i in range(nl_t - 1, -1, -1)] + [(1, L.STAGE_EMBED)], seen
        app.contrastive_step(px, ids, process_group=True, backward=True, zero_grad=True, reduce_gradients="force",
                             bucket_bytes=1 << 16)
        torch.cuda.synchronize()
        assert len(app.last_grad_buckets) >= 3
        assert app.last_grad_buckets[0][0] == 0 and app.last_grad_buckets[-1][1] == app._engine.grad_arena("step", px.device).total
        assert all(a[1] == b[0] for a, b in zip(app.last_grad_buckets, app.last_grad_buckets[1:]))
        fl = floor_of(plain)
        for n, g in plain.items():
            assert float((app._params[n].grad - g).norm()) <= 1e-5 * float(g.norm()) + fl, n
    finally:
        if own:
            dist.destroy_process_group()


def test_one_measured_side_stream_per_process(tmp_path):
    """The text tower's stream (model.py pick_side_stream): every engine of a process gets the SAME side stream, and it is one on which a
    trivial kernel finishes while the main stream is busy -- a stream that shares the main stream's hardware queue serialises the towers
    (round 6: 135.9 instead of 130.8 ms per training step, profiles/r6_autograd_side_stream_alias.log)."""
    from easynlp_amd.appzoo.clip import model as M
    burned = [torch.cuda.Stream() for _ in range(5)]          # streams "someone else" created first: candidates may alias the main queue
    for s_ in burned:
        with torch.cuda.stream(s_):
            torch.zeros(8, device="cuda")
    app1, _, _ = make_app(tmp_path / "a", "bf16")
    app2, _, _ = make_app(tmp_path / "b", "bf16")
    dev = torch.device("cuda", torch.cuda.current_device())
    s1, s2 = app1._engine.side_stream(dev), app2._engine.side_stream(dev)
    assert s1.cuda_stream == s2.cuda_stream and s1.cuda_stream != torch.cuda.current_stream().cuda_stream
    assert M._runs_beside(torch.cuda.current_stream(), s1, dev)
    assert len(M._SIDE_REJECTED) <= 7

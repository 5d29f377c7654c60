"""CPU tests of the data-parallel gradient machinery (easynlp_amd/parallel.py): the completion-ordered gradient arena and
the overlapped bucketed all-reduce, driven exactly as the library's progress hook drives them (head, block L-1 .. 0,
embeddings; image tower, then text tower) -- on two gloo ranks.  Also pins the premise of the weight-refresh rule: the
reference's own optimizers step through ``p.data`` and leave ``p._version`` untouched."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from easynlp_amd import lib as L
from easynlp_amd import parallel as P
from oracle import clip_oracle as O
from oracle import ref_harness as R


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _names_shapes():
    shapes = O.param_shapes(O.CONFIGS["tiny"])
    return list(shapes), shapes


def progress_sequence(cfg):
    """The notifications of one training step, in the order ezclip_backward_image / _text issue them (model.hip)."""
    seq = [(2, P.STAGE_HEAD)]
    for tower, nl in ((0, cfg["vision_layers"]), (1, cfg["text_num_hidden_layers"])):
        seq += [(tower, P.STAGE_HEAD)] + [(tower, i) for i in range(nl - 1, -1, -1)] + [(tower, P.STAGE_EMBED)]
    return seq


def test_arena_layout_is_completion_ordered_and_aligned():
    names, shapes = _names_shapes()
    a = P.GradArena(names, shapes, "cpu")
    assert a.group_order == sorted(a.group_order, key=P._completion_key)
    assert a.group_order == progress_sequence(O.CONFIGS["tiny"])
    end = 0
    for g in a.group_order:
        s, e = a.group_range[g]
        assert s == end and s % 64 == 0          # contiguous, 256-byte aligned groups
        end = e
    assert end == a.total
    seen = torch.zeros(a.total, dtype=torch.int32)
    for n in names:
        off, k = a.offsets[n]
        assert off % 4 == 0 and tuple(a.views[n].shape) == tuple(shapes[n]) and a.views[n].numel() == k
        assert a.group_range[P.grad_group(n)][0] <= off and off + k <= a.group_range[P.grad_group(n)][1]
        seen[off:off + k] += 1
    assert int(seen.max()) == 1                  # views do not overlap
    assert P.STAGE_HEAD == L.STAGE_HEAD and P.STAGE_EMBED == L.STAGE_EMBED
    # writing through a view lands in the flat buffer; zero() clears everything in one go
    a.views["text_projection"].fill_(2.0)
    assert float(a.flat.sum()) == 2.0 * a.views["text_projection"].numel()
    a.zero()
    assert float(a.flat.abs().sum()) == 0.0
    # arena.lent(): are views handed to autograd still alive somewhere?
    b = P.GradArena(names, shapes, "cpu", keep_views=False)
    assert not b.lent()
    v = b.make_views(b.flat)
    assert b.lent()
    p = torch.nn.Parameter(torch.zeros(shapes["text_projection"]))
    p.grad = v["text_projection"]
    del v
    assert b.lent()                              # lives on as .grad
    p.grad = None
    assert not b.lent()


def test_buckets_follow_the_progress_notifications():
    names, shapes = _names_shapes()
    a = P.GradArena(names, shapes, "cpu")
    calls = []
    r = P.OverlappedGradReducer(a, bucket_bytes=1, all_reduce=lambda t: calls.append((t.data_ptr(), t.numel())) or None)
    seq = progress_sequence(O.CONFIGS["tiny"])
    # out-of-order completion never sends a range that contains an unfinished group
    r.notify(*seq[2])
    assert r.buckets == []
    r.notify(*seq[0])
    assert r.buckets == [a.group_range[seq[0]]]
    r.notify(*seq[1])
    assert r.buckets[-1] == (a.group_range[seq[1]][0], a.group_range[seq[2]][1])
    for g in seq[3:]:
        r.notify(*g)
    r.finish()
    assert r.buckets[0][0] == 0 and r.buckets[-1][1] == a.total
    assert all(x[1] == y[0] for x, y in zip(r.buckets, r.buckets[1:]))
    assert sum(n for _, n in calls) == a.total
    # one large bucket size: everything goes out in finish()
    r2 = P.OverlappedGradReducer(a, bucket_bytes=1 << 40, all_reduce=lambda t: None)
    for g in seq:
        r2.notify(*g)
    assert r2.buckets == []
    r2.finish()
    assert r2.buckets == [(0, a.total)]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        names, shapes = _names_shapes()
        cfg = O.CONFIGS["tiny"]
        a = P.GradArena(names, shapes, "cpu")
        # (a bucket size no group boundary is a multiple of: buckets end wherever a notification pushes the range over it)
        r = P.OverlappedGradReducer(a, None, bucket_bytes=8 << 10 if world < 8 else 7001)
        g = torch.Generator().manual_seed(7)
        full = [torch.randn(a.total, generator=g) for _ in range(world)]       # every rank knows every rank's gradients
        # the backward pass fills the groups in completion order and notifies after each -- buckets fly in between
        for grp in progress_sequence(cfg):
            s, e = a.group_range[grp]
            a.flat[s:e] = full[rank][s:e]
            r.notify(*grp)
        launched_early = len(r.buckets)
        r.finish()
        want = sum(full)
        assert launched_early >= 2 and len(r.buckets) >= launched_early
        assert torch.allclose(a.flat, want, atol=1e-6)
        for n in names:                                                       # the parameters' .grad views see the sums
            off, k = a.offsets[n]
            assert torch.allclose(a.views[n].reshape(-1), want[off:off + k], atol=1e-6)
        # averaging variant (DDP semantics)
        a.flat.copy_(full[rank])
        r2 = P.OverlappedGradReducer(a, None, bucket_bytes=1 << 30, scale=1.0 / world)
        for grp in progress_sequence(cfg):
            r2.notify(*grp)
        r2.finish()
        assert torch.allclose(a.flat, want / world, atol=1e-6)
        # bf16 buckets (round 4; DDP's bf16_compress_hook): half the wire bytes, every hop of the sum rounded to bf16 -- the result
        # is the float32 sum within bf16 rounding of the partial sums, and every bucket that went out was a bf16 tensor
        a.flat.copy_(full[rank])
        sent = []
        r3 = P.OverlappedGradReducer(a, None, bucket_bytes=8 << 10, bucket_dtype=torch.bfloat16,
                                     all_reduce=lambda t: (sent.append((t.dtype, t.numel())), dist.all_reduce(t, async_op=True))[1])
        waits = []
        for grp in progress_sequence(cfg):
            r3.notify(*grp, wait=lambda grp=grp: waits.append(grp))          # (the event-log route: an explicit wait per group)
        r3.finish()
        assert sent and all(dt == torch.bfloat16 for dt, _ in sent) and sum(n for _, n in sent) == a.total
        # bucket_bytes counts WIRE bytes (round 5): every bucket but the last carries at least 8 KiB of bf16, i.e. >= 4096
        # elements -- with the old float32 accounting they were half that and twice as many
        assert all(n * 2 >= (8 << 10) for _, n in sent[:-1]), sent
        assert set(waits) == set(progress_sequence(cfg))                     # every group's producer was waited for before its bucket left
        scale_ = sum(f.abs() for f in full)
        assert float(((a.flat - want).abs() / (scale_ + 1e-6)).max()) < (world + 1) * 2.0 ** -8      # (one rounding -- half a bf16 ulp, 2^-8 relative -- per input and per hop)
        # the evaluator's sharded recall ranks: every rank fills its slice, one all-gather into a SEPARATE buffer
        from easynlp_amd.appzoo.clip.evaluator import gather_rank_shards
        n_q = 4 * world - 1                                   # (the last rank's slice is ragged)
        per = (n_q + world - 1) // world
        mine = torch.zeros(per * world, dtype=torch.int32)
        mine[rank * per:(rank + 1) * per] = torch.arange(rank * per, (rank + 1) * per, dtype=torch.int32) * 3 + 1
        got = gather_rank_shards(mine, rank, per, n_q)
        assert torch.equal(got, torch.arange(n_q, dtype=torch.int32) * 3 + 1)
        # ... and the whole sharded sweep, both directions (round 4): ranks split the query blocks, all-gather the text -> image
        # ranks, all-reduce the image -> text counts.  The two device calls are stood in for by the counting definition on the CPU.
        import easynlp_amd.appzoo.clip.evaluator as EV
        from oracle import clip_oracle as ORC
        def paired_cpu(t, v):
            return (t * v).sum(-1)
        def block_cpu(t_rows, v, row0, paired, out, cols):
            sim = t_rows @ v.t()
            rows, n = sim.shape
            i = torch.arange(row0, row0 + rows)[:, None]
            j = torch.arange(n)[None, :]
            d = sim[torch.arange(rows), torch.arange(row0, row0 + rows)][:, None]
            out[:rows] = ((sim > d) | ((sim == d) & (j < i))).sum(1).to(torch.int32)
            if cols is not None:
                dc = torch.stack([v[c] @ text_all[c] for c in range(n)])[None, :]
                cols += ((sim > dc) | ((sim == dc) & (i < j))).sum(0).to(torch.int32)
        g = torch.Generator().manual_seed(3)
        n_p = 8 * world + 3
        # small integer embeddings: every score is exact in float32 whatever the summation order, and ties abound
        text_all = torch.randint(-3, 4, (n_p, 16), generator=g).float()
        img_all = text_all + torch.randint(-2, 3, (n_p, 16), generator=g).float()
        img_all[1] = img_all[n_p - 2]
        EV._paired_scores, EV._ranks_block = paired_cpu, block_cpu
        t2i, i2t = EV.recall_ranks(text_all, img_all, block_rows=3, shard=True, both_directions=True)
        want_r, want_c = ORC.recall_ranks(text_all, img_all)
        assert torch.equal(t2i.long(), want_r) and torch.equal(i2t.long(), want_c), (rank, t2i, want_r, i2t, want_c)
        assert torch.equal(EV.recall_ranks(text_all, img_all, block_rows=5, shard=True).long(), want_r)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_overlapped_reduction_two_gloo_ranks(world):
    """(world 8: the north star's node -- eight ranks, a bucket size that is no multiple of any group boundary)"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_a_failed_bucket_launch_inside_the_progress_hook_is_raised_by_finish():
    """notify() runs inside a ctypes callback (from within ezclip_backward_*), where an exception would be printed and
    dropped: the reducer keeps it, stops sending, and finish() raises it -- gradients are never left silently unreduced."""
    names, shapes = _names_shapes()
    a = P.GradArena(names, shapes, "cpu")
    calls = []

    def flaky(t):
        calls.append(t.numel())
        if len(calls) == 2:
            raise RuntimeError("collective failed")
        return None
    r = P.OverlappedGradReducer(a, bucket_bytes=1, all_reduce=flaky)
    for g in progress_sequence(O.CONFIGS["tiny"]):
        r.notify(*g)                     # (never raises: it is a callback)
    assert len(calls) == 2               # nothing was sent after the failure
    with pytest.raises(RuntimeError, match="gradient bucket"):
        r.finish()
    r.reset()
    r.finish()                           # usable again after reset()


@pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")
def test_reference_optimizers_step_without_touching_version_counters():
    """Why CLIPApp re-packs its weight copies after every backward pass instead of watching ``p._version``: the CLI-default
    AdamW and BertAdam of the reference (easynlp/core/optimizers.py:367,451,462) write through ``p.data``."""
    R.install_shims()
    from easynlp.core.optimizers import AdamW, BertAdam
    for cls, kw in ((AdamW, dict(lr=1e-2)), (BertAdam, dict(lr=1e-2, warmup=-1, t_total=-1))):
        p = torch.nn.Parameter(torch.ones(4, 3))
        opt = cls([p], **kw)
        before = p.detach().clone()
        v = p._version
        p.grad = torch.ones_like(p)
        opt.step()
        assert not torch.equal(p.detach(), before)
        assert p._version == v, "%s now bumps the version counter" % cls.__name__

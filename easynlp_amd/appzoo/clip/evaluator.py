"""Recall@k evaluator -- mirror of ``easynlp.appzoo.clip.evaluator.CLIPEvaluator``
(easynlp/appzoo/clip/evaluator.py:27-72; base easynlp/core/evaluator.py:19-34).

Same contract: ``evaluate(model) -> [("mean_recall", float)]`` with text->image
R@1/5/10 over the whole validation set.  The per-row Python loop with a full
``torch.sort`` (evaluator.py:53-61) is replaced by a fused HIP sweep: the f32
MFMA kernel that computes a similarity tile compares it, in registers, with the
paired scores and counts, per query, how many images score above the paired one
(``ezclip_recall_ranks_fused``): hit@k  <=>  rank < k.  No similarity block is
materialised, and the image->text ranks come out of the same sweep on request
(``both_directions=True``; the reference reports text->image only).
"""
from __future__ import annotations

import time

import torch
from torch.utils.data import DataLoader

from ... import lib as L


class Evaluator(object):
    def __init__(self, valid_dataset, **kwargs):
        eval_batch_size = kwargs.get("eval_batch_size", 32)
        self.valid_loader = DataLoader(valid_dataset, batch_size=eval_batch_size, shuffle=False,
                                       collate_fn=valid_dataset.batch_fn)
        self.best_valid_score = float("-inf")

    def evaluate(self, model):
        raise NotImplementedError

    @property
    def eval_metrics(self):
        raise NotImplementedError


def _paired_scores(t: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """paired[i] = <t_i, v_i> by the diagonal tiles of the f32 similarity kernel (``ezclip_recall_paired_scores``)"""
    n, e = t.shape
    paired = torch.empty(n, dtype=torch.float32, device=t.device)
    L.check(L.load().ezclip_recall_paired_scores(L.ptr(t), L.ptr(v), n, e, L.ptr(paired), L.stream_ptr()), "recall_paired_scores")
    return paired


def _ranks_block(t_rows: torch.Tensor, v: torch.Tensor, row0: int, paired: torch.Tensor, out: torch.Tensor, cols) -> None:
    """one sweep of queries row0 .. row0 + rows - 1 over the gallery (``ezclip_recall_ranks_fused``): ``out[:rows]`` = their
    text -> image ranks, ``cols`` (int32 [n] or None) += the block's image -> text counts"""
    rows, e = t_rows.shape
    L.check(L.load().ezclip_recall_ranks_fused(L.ptr(t_rows), L.ptr(v), rows, row0, v.shape[0], e, L.ptr(paired), out.data_ptr(),
                                               cols.data_ptr() if cols is not None else None, L.stream_ptr()), "recall_ranks_fused")


def recall_ranks(text_embeds: torch.Tensor, image_embeds: torch.Tensor, block_rows: int = 4096, group=None,
                 shard: bool = False, both_directions: bool = False, materialise: bool = False):
    """rank[i] = #{j : <t_i, v_j> > <t_i, v_i>} (+ ties with j < i) on the GPU, ``block_rows`` queries at a time, compared inside the
    similarity kernel (``ezclip_recall_paired_scores`` + ``ezclip_recall_ranks_fused``: no scratch at all).
    ``both_directions=True`` returns ``(text->image ranks, image->text ranks)``; the latter count, for image j, the texts i with
    <t_i, v_j> > <t_j, v_j> (+ ties with i < j).  ``materialise=True`` keeps the older two-kernel form (a [block_rows, n] similarity
    block in scratch, ``ezclip_recall_ranks_rows``; text->image only) -- the cross-check of the tests.
    ``shard=True`` under an initialised process group: every rank holds the same embeddings (the reference evaluator runs the
    whole validation set on each rank, core/evaluator.py:20-27), ranks split the query blocks, exchange the text->image ranks with
    one all-gather and sum the image->text counts with one all-reduce."""
    import torch.distributed as dist
    t = text_embeds.detach().float().contiguous()
    v = image_embeds.detach().float().contiguous()
    n, e = t.shape
    if materialise and both_directions:
        raise ValueError("recall_ranks: the materialising form computes text->image ranks only")
    world, me = 1, 0
    if shard and dist.is_available() and dist.is_initialized():
        world, me = dist.get_world_size(group), dist.get_rank(group)
    per = (n + world - 1) // world                         # this rank's queries: [lo, hi)
    lo, hi = min(n, me * per), min(n, (me + 1) * per)
    rank = torch.zeros(per * world if world > 1 else n, dtype=torch.int32, device=t.device)
    rows_max = max(1, min(int(block_rows), hi - lo)) if hi > lo else 1
    cols = torch.zeros(n, dtype=torch.int32, device=t.device) if both_directions else None
    if materialise:
        scratch = torch.empty(rows_max * n, dtype=torch.float32, device=t.device)
    else:
        paired = _paired_scores(t, v)
    for r0 in range(lo, hi, rows_max):
        rows = min(rows_max, hi - r0)
        out = rank[(me * per if world > 1 else 0) + (r0 - lo):]
        if materialise:
            L.check(L.load().ezclip_recall_ranks_rows(L.ptr(t[r0:r0 + rows]), L.ptr(v), rows, r0, n, e, out.data_ptr(), L.ptr(scratch),
                                                      L.stream_ptr()), "recall_ranks_rows")
        else:
            _ranks_block(t[r0:r0 + rows], v, r0, paired, out, cols)
    if world > 1:
        rank = gather_rank_shards(rank, me, per, n, group)
        if cols is not None:
            dist.all_reduce(cols, group=group)
    return (rank, cols) if both_directions else rank


def gather_rank_shards(rank: torch.Tensor, me: int, per: int, n: int, group=None) -> torch.Tensor:
    """Every rank filled rank[me*per : (me+1)*per]; returns the first n entries of the concatenation over ranks.  Input and
    output of the collective are separate tensors (a slice of ``rank`` as the input would alias the output buffer: in-place
    all-gather is only defined for NCCL at the exact offset, not for gloo)."""
    import torch.distributed as dist
    mine = rank[me * per:(me + 1) * per].clone()
    gathered = torch.empty_like(rank)
    dist.all_gather_into_tensor(gathered, mine, group=group)
    return gathered[:n]


def _recall_from_ranks(rank, ks):
    n = rank.numel()
    stats = [int((rank < k).sum().item()) for k in ks]
    rs = [s * 1.0 / n for s in stats]
    return (sum(rs) / len(rs),) + tuple(rs), stats


def recall_at_k(text_embeds, image_embeds, ks=(1, 5, 10), both_directions: bool = False):
    """((mean_recall, r@k...), hit counts) of text->image retrieval; with ``both_directions`` a pair of those: (text->image, image->text)"""
    if not both_directions:
        return _recall_from_ranks(recall_ranks(text_embeds, image_embeds), ks)
    t2i, i2t = recall_ranks(text_embeds, image_embeds, both_directions=True)
    return _recall_from_ranks(t2i, ks), _recall_from_ranks(i2t, ks)


def recall_report(text_embeds, gallery_embeds, spent_seconds: float, both_directions: bool = False):
    """text -> gallery R@1/5/10 + mean, printed the way the reference evaluators do (clip/evaluator.py:62-72), returned as the
    metric list the Trainer compares (``[("mean_recall", fraction)]``: text -> gallery, as in the reference).  ``both_directions`` adds
    one line for gallery -> text (same sweep of the similarity kernel) and an ``("i2t_mean_recall", fraction)`` entry AFTER the first."""
    n = text_embeds.shape[0]
    extra = []
    if both_directions:
        ((mean_recall, r1, r5, r10), hits), ((m2, q1, q5, q10), hits2) = recall_at_k(text_embeds, gallery_embeds, both_directions=True)
    else:
        (mean_recall, r1, r5, r10), hits = recall_at_k(text_embeds, gallery_embeds)
    print(" ".join("r%d_num:%d" % (k, h) for k, h in zip((1, 5, 10), hits)), "query_num:" + str(n))
    print(" ".join("%s(%%):%s" % (name, v * 100) for name, v in (("r1", r1), ("r5", r5), ("r10", r10), ("mean_recall", mean_recall))))
    if both_directions:
        print("image->text", " ".join("r%d_num:%d" % (k, h) for k, h in zip((1, 5, 10), hits2)),
              " ".join("%s(%%):%s" % (name, v * 100) for name, v in (("r1", q1), ("r5", q5), ("r10", q10), ("mean_recall", m2))))
        extra = [("i2t_mean_recall", m2)]
    print("Inference time = {:.2f}s, [{:.4f} ms / sample] ".format(spent_seconds, spent_seconds * 1000 / max(n, 1)))
    return [("mean_recall", mean_recall)] + extra


class CLIPEvaluator(Evaluator):

    def __init__(self, valid_dataset, **kwargs):
        super().__init__(valid_dataset, **kwargs)
        self.metrics = ["accuracy", "f1"]
        self.before = 0.0
        self.both_directions = bool(kwargs.get("both_directions", False))      # (not in the reference: it reports text -> image only)

    def evaluate(self, model):
        model.eval()
        total_spent_time = 0.0
        image_embeds_all, text_embeds_all = [], []
        # batch k + 1 crosses PCIe while batch k is encoded (identity for a model on the CPU)
        from .data import DevicePrefetcher
        p0 = next(iter(model.parameters()), None)
        loader = DevicePrefetcher(self.valid_loader, p0.device) if (p0 is not None and p0.is_cuda) else self.valid_loader
        for _step, batch in enumerate(loader):
            infer_start_time = time.time()
            with torch.no_grad():
                outputs = model(batch, feat=True) if getattr(model, "_engine", None) is not None else model(batch)
            total_spent_time += time.time() - infer_start_time
            image_embeds_all.append(outputs["image_embeds"])
            text_embeds_all.append(outputs["text_embeds"])
        image_embeds_tensor = torch.cat(image_embeds_all, dim=0)
        text_embeds_tensor = torch.cat(text_embeds_all, dim=0)
        return recall_report(text_embeds_tensor, image_embeds_tensor, total_spent_time, both_directions=self.both_directions)

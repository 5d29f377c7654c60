"""The one exchange step of the path: sharding the global
contrastive batch over ranks with ``torch.distributed`` (backend ``nccl`` = RCCL
over xGMI on MI355X; ``gloo`` in the CPU tests).

Forward: ONE all-gather of ``[n, 2E]`` (image | text embeddings).  Every rank then
evaluates its own rows of both similarity directions, so both log-sum-exps are
rank-local -- no second collective.  Backward: each rank holds gradient
contributions to *all* rows; ONE reduce-scatter (sum) returns the rows it owns.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def world_info(group=None) -> Tuple[int, int]:
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def gather_embeddings(img: torch.Tensor, txt: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """[n, E] x 2 per rank -> ([N, E] images, [N, E] texts, row offset of this rank), N = world * n."""
    world, rank = world_info(group)
    n, e = img.shape
    if world == 1:
        return img, txt, 0
    both = torch.cat([img, txt], dim=1).contiguous()
    gathered = torch.empty((world * n, 2 * e), dtype=both.dtype, device=both.device)
    dist.all_gather_into_tensor(gathered, both, group=group)
    return gathered[:, :e].contiguous(), gathered[:, e:].contiguous(), rank * n


def scatter_embedding_grads(d_img_all: torch.Tensor, d_txt_all: torch.Tensor, n_local: int,
                            group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Sum the per-rank contributions [N, E] x 2 over ranks and keep this rank's n_local rows."""
    world, rank = world_info(group)
    if world == 1:
        return d_img_all, d_txt_all
    e = d_img_all.shape[1]
    both = torch.cat([d_img_all, d_txt_all], dim=1).contiguous()
    mine = torch.empty((n_local, 2 * e), dtype=both.dtype, device=both.device)
    try:
        dist.reduce_scatter_tensor(mine, both, op=dist.ReduceOp.SUM, group=group)
    except (RuntimeError, NotImplementedError):
        # backends without reduce-scatter (gloo): all-reduce and keep the local slice
        dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)
        mine = both[rank * n_local:(rank + 1) * n_local].clone()
    return mine[:, :e].contiguous(), mine[:, e:].contiguous()


def sum_gradients(params, group=None, bucket_bytes: int = 256 << 20) -> None:
    """Sum parameter gradients over ranks.  This is what follows ``CLIPApp.contrastive_step(backward=True)`` with a
    process group: its embedding gradients are already those of the GLOBAL mean loss (grad_scale 1 / world inside), so
    the per-rank parameter gradients are partial sums of the true gradient."""
    average_gradients(params, group, bucket_bytes, average=False)


def average_gradients(params, group=None, bucket_bytes: int = 256 << 20, average: bool = True) -> None:
    """DDP-style gradient averaging (``average=False``: plain sum) with large flat buckets: xGMI rings are per-link bound,
    so few big all-reduces (256 MiB) beat many 25 MiB ones."""
    world, _ = world_info(group)
    if world == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    bucket, size = [], 0

    def flush():
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(world)
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    for g in grads:
        bucket.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            flush()
            bucket, size = [], 0
    flush()

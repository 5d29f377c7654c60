"""The one exchange step of the path: sharding the global
contrastive batch over ranks with ``torch.distributed`` (backend ``nccl`` = RCCL
over xGMI on MI355X; ``gloo`` in the CPU tests).

Forward: ONE all-gather of ``[n, 2E]`` (image | text embeddings).  Every rank then
evaluates its own rows of both similarity directions, so both log-sum-exps are
rank-local -- no second collective.  Backward: each rank holds gradient
contributions to *all* rows; ONE reduce-scatter (sum) returns the rows it owns.
"""
from __future__ import annotations

import re
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

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
    if dist.get_backend(group) == "gloo":
        # gloo (the CPU tests) has no reduce-scatter: all-reduce and keep the local slice.  Decided by the backend's NAME, not
        # by catching an exception: an RCCL error must surface as an RCCL error, not as a silent all-reduce
        dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)
        mine = both[rank * n_local:(rank + 1) * n_local].clone()
    else:
        dist.reduce_scatter_tensor(mine, both, op=dist.ReduceOp.SUM, group=group)
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


# ---------------------------------------------------------------------------------------------------------------------
# Gradient arena + overlapped all-reduce (the data-parallel half of a training step; reference: the bucketed
# all-reduce hooks of DistributedDataParallel that easynlp/core/trainer.py:101-108 wraps the model in).
#
# The library finishes the gradients group by group -- head, block L-1 ... block 0, embeddings, image tower first, then
# the text tower -- and says so through ``ezclip_set_backward_progress``.  ``GradArena`` lays every gradient out in ONE
# flat float32 buffer in exactly that order, so "the groups finished so far" is always one contiguous byte range:
# zeroing the gradients is one memset, and a bucket of the all-reduce is one slice (no ``torch.cat`` staging copies).
# ``OverlappedGradReducer`` turns progress notifications into asynchronous all-reduces of such slices while the
# backward kernels of the remaining groups are still running.  Buckets are float32 (the gradients' own dtype: the sum is
# what DDP would produce); over xGMI a ring all-reduce is per-link bound, so buckets are few and large (64 MiB default).

STAGE_HEAD, STAGE_EMBED = 1000000, -1          # include/ezclip.h: EZCLIP_STAGE_HEAD / EZCLIP_STAGE_EMBED
_LAYER_RE = re.compile(r"(?:^|\.)(?:resblocks|layer|layers)\.(\d+)\.")
_HEAD_NAMES = ("visual.proj", "visual.proj_bias", "visual.ln_post.weight", "visual.ln_post.bias", "text_projection",
               "text_projection_bias", "ln_final.weight", "ln_final.bias", "bert.pooler.dense.weight",
               "bert.pooler.dense.bias")


def grad_group(name: str) -> Tuple[int, int]:
    """(tower, stage) of a library parameter name: the progress notification after which its gradient is final.
    tower 0 image / 1 text / 2 neither (logit_scale: written by the contrastive step itself, before either tower)."""
    if name == "logit_scale":
        return (2, STAGE_HEAD)
    tower = 0 if name.startswith("visual.") else 1
    if name in _HEAD_NAMES:
        return (tower, STAGE_HEAD)
    m = _LAYER_RE.search(name)
    if m:
        return (tower, int(m.group(1)))
    return (tower, STAGE_EMBED)


def _completion_key(group: Tuple[int, int]):
    tower, stage = group
    t = {2: 0, 0: 1, 1: 2}[tower]                 # logit_scale, image tower, text tower
    s = 0 if stage == STAGE_HEAD else (10 ** 9 if stage == STAGE_EMBED else 10 ** 6 - stage)
    return (t, s)


class GradArena:
    """One flat float32 buffer for the gradients of ``names`` (shapes ``shapes``), grouped in completion order."""

    def __init__(self, names: Sequence[str], shapes: Dict[str, tuple], device, align: int = 64, keep_views: bool = True):
        groups: Dict[Tuple[int, int], List[str]] = {}
        for n in names:
            groups.setdefault(grad_group(n), []).append(n)
        self.group_order = sorted(groups, key=_completion_key)
        self.offsets: Dict[str, Tuple[int, int]] = {}
        self.group_range: Dict[Tuple[int, int], Tuple[int, int]] = {}
        off = 0
        for g in self.group_order:
            start = off
            for n in groups[g]:
                k = 1
                for d in shapes[n]:
                    k *= int(d)
                self.offsets[n] = (off, k)
                off += k + (-k) % 4                      # every view 16-byte aligned (ezclip_bind_param)
            off += (-off) % align                        # every group 256-byte aligned
            self.group_range[g] = (start, off)
        self.total = off
        self.shapes = {n: tuple(int(d) for d in shapes[n]) for n in names}
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.views = self.make_views(self.flat) if keep_views else {}
        self._base_refs = self._storage_refs()

    _warned_no_use_count = False

    def _storage_refs(self) -> int:
        fn = getattr(torch._C, "_storage_Use_Count", None)
        if fn is None and not GradArena._warned_no_use_count:
            GradArena._warned_no_use_count = True
            import warnings
            warnings.warn("torch._C._storage_Use_Count is missing in this torch build: the autograd path cannot tell whether its "
                          "gradient arena is still referenced and allocates (and zeroes) a fresh one for every backward pass")
        return int(fn(self.flat.untyped_storage()._cdata)) if fn is not None else -1

    def lent(self) -> bool:
        """Does anything besides the arena itself still reference its memory (views handed to autograd that live on as
        ``.grad``, or sit in the engine waiting to be accumulated)?  Unknown (no storage use count in this torch) = yes."""
        n = self._storage_refs()
        return n < 0 or n > self._base_refs

    def make_views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {n: flat[o:o + k].view(self.shapes[n]) for n, (o, k) in self.offsets.items()}

    def fresh(self) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """A second, zeroed buffer with the same layout (a backward pass whose results autograd must ADD to live gradients)."""
        flat = torch.zeros(self.total, dtype=torch.float32, device=self.flat.device)
        return flat, self.make_views(flat)

    def zero(self) -> None:
        self.flat.zero_()

    def owns(self, name: str, t: Optional[torch.Tensor]) -> bool:
        v = self.views.get(name)
        return t is not None and v is not None and t.data_ptr() == v.data_ptr() and t.shape == v.shape


class OverlappedGradReducer:
    """All-reduce (sum) the arena's gradients over ``group`` while the backward pass is still running.

    ``notify(tower, stage)`` is called (from the library's progress hook) when a group's kernels have been enqueued; once
    the finished-but-unsent range reaches ``bucket_bytes`` it is all-reduced asynchronously -- on RCCL that runs on the
    process group's own stream, ordered behind everything enqueued so far on the current stream.  ``finish()`` sends the
    rest, waits for every bucket (the current stream then waits for the collectives) and applies ``scale``."""

    def __init__(self, arena: GradArena, group=None, bucket_bytes: int = 64 << 20, scale: float = 1.0,
                 all_reduce: Optional[Callable] = None, bucket_dtype: Optional[torch.dtype] = None):
        self.arena, self.group, self.bucket_bytes, self.scale = arena, group, int(bucket_bytes), float(scale)
        # bucket_dtype = torch.bfloat16: each bucket is cast to bf16, all-reduced, and cast back -- half the bytes on the xGMI
        # links (755 -> 378 MB per step for ViT-B/16 + BERT-base), sums rounded to bf16 hop by hop: what DDP's
        # ``bf16_compress_hook`` does.  Default None = float32 buckets, bit for bit what DDP's own all-reduce produces.
        self.bucket_dtype = None if bucket_dtype in (None, torch.float32) else bucket_dtype
        self._all_reduce = all_reduce or (lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True))
        self.reset()

    def reset(self) -> None:
        self._done = {}                # group -> callable that orders the CURRENT stream behind the group's kernels (None: nothing to wait for)
        self._sent = 0                 # element offset up to which buckets have been launched
        self._works: List = []
        self._staged: List = []        # (arena slice, compressed bucket) pairs to copy back in finish()
        self.buckets: List[Tuple[int, int]] = []
        self._error: Optional[BaseException] = None

    def _frontier(self) -> int:
        end = self._sent
        for g in self.arena.group_order:
            s, e = self.arena.group_range[g]
            if e <= end:
                continue
            if g not in self._done:
                break
            end = e
        return end

    def _launch(self, upto: int) -> None:
        if upto <= self._sent:
            return
        # a bucket may hold groups produced on another stream than the current one (the towers run on two streams, and a
        # group that finished early is sent together with the later one that completes the contiguous range)
        for g, wait in self._done.items():
            s, e = self.arena.group_range[g]
            if wait is not None and s < upto and e > self._sent:
                wait()
        self.buckets.append((self._sent, upto))
        piece = self.arena.flat[self._sent:upto]
        if self.bucket_dtype is not None:
            staged = piece.to(self.bucket_dtype)          # (on the current stream, behind the waits above)
            self._staged.append((piece, staged))
            piece = staged
        self._works.append(self._all_reduce(piece))
        self._sent = upto

    def notify(self, tower: int, stage: int, wait: Optional[Callable[[], None]] = None) -> None:
        """Group (tower, stage) is final once the work enqueued so far has run.  ``wait``: a callable that orders the current
        stream behind that work -- the library's progress event log hands one per group (CLIPApp.contrastive_step: drained
        after the backward call returned, no Python inside it).  Without it (the legacy ctypes callback from inside
        ezclip_backward_*, tests) an event is recorded on the current stream here.  An exception raised in a ctypes callback
        would be swallowed (printed, not propagated) and leave gradients unreduced: it is kept and re-raised by ``finish()``."""
        if self._error is not None:
            return
        try:
            self._notify(tower, stage, wait)
        except BaseException as e:      # noqa: BLE001  (re-raised in finish())
            self._error = e

    def _notify(self, tower: int, stage: int, wait=None) -> None:
        if wait is None and self.arena.flat.is_cuda:
            ev = torch.cuda.Event()
            ev.record()                # on the current stream: the caller enters the producing tower's stream first
            wait = lambda ev=ev: torch.cuda.current_stream().wait_event(ev)     # noqa: E731
        self._done[(tower, stage)] = wait
        end = self._frontier()
        # bucket_bytes counts bytes ON THE WIRE: with bf16 buckets an element is 2 bytes there, so a bucket holds twice the
        # elements and the number of collectives stays what bucket_bytes asks for
        wire = 4 if self.bucket_dtype is None else torch.empty((), dtype=self.bucket_dtype).element_size()
        if (end - self._sent) * wire >= self.bucket_bytes:
            self._launch(end)

    def finish(self) -> None:
        """Every group is final (the backward calls have returned): send what is left, wait, scale."""
        if self._error is not None:
            err, self._error = self._error, None
            for w in self._works:           # (do not leave collectives of the buckets already sent unwaited)
                if w is not None:
                    w.wait()
            self._works = []
            # the buckets that did go out are complete sums: copy them back so the arena is not left half bf16-staged, and
            # drop the staging list (the arena still mixes reduced and unreduced groups: the step must be abandoned)
            for piece, staged in self._staged:
                piece.copy_(staged)
            self._staged = []
            raise RuntimeError("a gradient bucket could not be launched from the backward progress hook; the arena holds a mix "
                               "of reduced and unreduced gradients -- discard this step") from err
        self._launch(self.arena.total)
        for w in self._works:
            if w is not None:
                w.wait()
        for piece, staged in self._staged:
            piece.copy_(staged)
        self._staged = []
        if self.scale != 1.0:
            self.arena.flat.mul_(self.scale)
        self._works = []

"""The whole data-parallel training step with REAL kernels on two ranks.

No multi-GPU box is reachable from the build side and RCCL refuses two ranks on one device, so two processes share cuda:0 and talk
through gloo (which stages device tensors through the host): each rank runs `CLIPApp.contrastive_step(process_group=True,
backward=True, reduce_gradients=True)` on its half of the batch -- all-gather of the embeddings, the tiled contrastive kernels
on its rows of the global batch (rank 1: row offset n), reduce-scatter of the embedding gradients, both towers' backward with the
library's progress callbacks driving the bucketed all-reduce of the gradient arena -- and the result must be the single-process
step on the whole batch: mean of the rank losses = the loss, reduced gradients = the gradients (reference behaviour:
core/trainer.py:101-108 + the global loss of DESIGN 5)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from oracle import clip_oracle as O
from oracle import ref_harness as R

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, ckpt, dtype, q):
    try:
        import torch.distributed as dist
        from easynlp_amd.appzoo.clip import CLIPApp
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        cfg = O.CONFIGS["small"]
        n = 6
        px, ids = O.make_inputs(cfg, world * n, 24, 5)
        app = CLIPApp(ckpt, user_defined_parameters={"clip_compute_dtype": dtype}).cuda()
        app.eval()                                     # (dropout off: the two runs must see the same function)
        try:
            loss = app.contrastive_step(px[rank * n:(rank + 1) * n].cuda(), ids[rank * n:(rank + 1) * n].cuda(), process_group=True,
                                        backward=True, zero_grad=True, reduce_gradients=True, bucket_bytes=64 << 10)
        except RuntimeError as e:
            if "gloo" in str(e).lower() or "not supported" in str(e).lower() or "not implemented" in str(e).lower():
                q.put((rank, "SKIP: %s" % str(e)[:200]))
                return
            raise
        torch.cuda.synchronize()
        lt = loss.detach().clone()
        dist.all_reduce(lt)
        grads = {k: p.grad.detach().float().cpu().clone() for k, p in app._params.items() if p.grad is not None}
        buckets = list(getattr(app, "last_grad_buckets", []) or [])
        if rank == 0:
            ref = CLIPApp(ckpt, user_defined_parameters={"clip_compute_dtype": dtype}).cuda()
            ref.eval()
            full = ref.contrastive_step(px.cuda(), ids.cuda(), process_group=False, backward=True, zero_grad=True)
            torch.cuda.synchronize()
            f32 = dtype == "fp32"
            assert abs(lt.item() / world - full.item()) < (1e-5 if f32 else 3e-3), (lt.item() / world, full.item())
            want = {k: p.grad.detach().float().cpu() for k, p in ref._params.items() if p.grad is not None}
            assert set(want) == set(grads)
            floor = 1e-3 * max(float(v.norm()) for v in want.values())
            worst = max((float((grads[k] - v).norm()) / (float(v.norm()) + floor), k) for k, v in want.items())
            assert worst[0] < (2e-4 if f32 else 3e-2), worst
            assert len(buckets) >= 2, buckets            # the arena went out in several buckets while the backward was running
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        try:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:
            pass


@pytest.mark.parametrize("world,dtype", [(2, "fp32"), (2, "bf16"), (4, "bf16")])
def test_two_ranks_on_one_gpu_equal_the_single_process_step(tmp_path, world, dtype):
    """(world 4: ranks at row offsets 6, 12, 18 of a 24-pair global batch -- ragged tiles of the contrastive kernels)"""
    cfg = O.CONFIGS["small"]
    R.write_checkpoint_dir(str(tmp_path), cfg, O.make_state_dict(cfg, 9))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), dtype, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    if any(r[1].startswith("SKIP") for r in res):
        pytest.skip("gloo cannot move device tensors in this build: " + [r[1] for r in res if r[1].startswith("SKIP")][0])
    assert all(r[1] == "ok" for r in res), res

"""N > 1 path on CPU: world_size 2, gloo.  The exchange step (easynlp_amd/parallel.py) is
exercised with real collectives; the per-rank compute is stood in for by the CPU oracle's
per-rank share of the global InfoNCE (oracle.global_clip_loss_rank), which is also the
oracle of the HIP kernel ezclip_infonce_fused (tests/test_ops_gpu.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import clip_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, e, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from easynlp_amd import parallel as P
        g = torch.Generator().manual_seed(123)
        T = torch.nn.functional.normalize(torch.randn(world * n, e, generator=g, dtype=torch.float64), dim=-1)
        I = torch.nn.functional.normalize(T + torch.randn(world * n, e, generator=g, dtype=torch.float64), dim=-1)
        ls = torch.tensor(2.0, dtype=torch.float64)
        sl = slice(rank * n, (rank + 1) * n)
        img_l = I[sl].clone().requires_grad_(True)
        txt_l = T[sl].clone().requires_grad_(True)
        img_all, txt_all, off = P.gather_embeddings(img_l.detach(), txt_l.detach(), None)
        assert off == rank * n
        assert torch.equal(img_all, I) and torch.equal(txt_all, T)       # layout of the gathered batch
        # rank-local share of the global loss and its gradient w.r.t. ALL rows (what the HIP kernel returns)
        ia, ta = img_all.clone().requires_grad_(True), txt_all.clone().requires_grad_(True)
        loss_r = O.global_clip_loss_rank(ta, ia, ls, rank, n)
        (loss_r / world).backward()                                      # grad_scale = 1 / world
        d_img_l, d_txt_l = P.scatter_embedding_grads(ia.grad, ta.grad, n, None)
        # oracle: the full-batch reference loss on one process
        Ig, Tg = I.clone().requires_grad_(True), T.clone().requires_grad_(True)
        full = O.clip_loss((Tg @ Ig.t()) * ls.exp())
        full.backward()
        lt = torch.tensor([loss_r.item()], dtype=torch.float64)
        dist.all_reduce(lt)
        assert abs(lt.item() / world - full.item()) < 1e-12
        assert float((d_img_l - Ig.grad[sl]).abs().max()) < 1e-12
        assert float((d_txt_l - Tg.grad[sl]).abs().max()) < 1e-12
        # gradient averaging helper
        p = torch.nn.Parameter(torch.zeros(5))
        p.grad = torch.full((5,), float(rank + 1))
        P.average_gradients([p])
        assert torch.allclose(p.grad, torch.full((5,), (1 + world) / 2.0))
        q.put((rank, "ok"))
    except Exception as ex:  # pragma: no cover
        q.put((rank, repr(ex)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,e", [(3, 8), (16, 32)])
def test_global_contrastive_exchange_world2_gloo(n, e):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, e, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res

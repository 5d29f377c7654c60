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
        p.grad = torch.full((5,), float(rank + 1))
        P.sum_gradients([p])
        assert torch.allclose(p.grad, torch.full((5,), world * (1 + world) / 2.0))
        q.put((rank, "ok"))
    except Exception as ex:  # pragma: no cover
        q.put((rank, repr(ex)))
    finally:
        dist.destroy_process_group()


def _oracle_shard(txt_all, img_all, n, off, ls, grad_scale, need):
    """stand-in for the HIP kernel behind fused_infonce_shard: this rank's rows of the global InfoNCE + gradients"""
    with torch.enable_grad():            # (autograd.Function.forward runs with grad mode off)
        ta, ia = txt_all.clone().requires_grad_(True), img_all.clone().requires_grad_(True)
        lsv = ls.clone().requires_grad_(True)
        loss = O.global_clip_loss_rank(ta, ia, lsv, off // n, n)
        if not need:
            return loss.detach(), None, None, None
        (loss * grad_scale).backward()
    return loss.detach(), ta.grad, ia.grad, lsv.grad


def _scope_worker(rank, world, port, n, e, q):
    """contrastive_scope='global' on the autograd path under DDP semantics: per-rank backward through _GlobalInfoNCEFn,
    parameter gradients AVERAGED over ranks == gradient of the single-process full-batch loss."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from easynlp_amd import parallel as P
        from easynlp_amd.appzoo.clip.model import _GlobalInfoNCEFn
        g = torch.Generator().manual_seed(7)
        X = torch.randn(world * n, e, generator=g, dtype=torch.float64)          # raw "image" / "text" features of all pairs
        Y = X + 0.5 * torch.randn(world * n, e, generator=g, dtype=torch.float64)
        W0 = torch.randn(e, e, generator=g, dtype=torch.float64) / e ** 0.5      # the replicated "encoder" parameters
        V0 = torch.randn(e, e, generator=g, dtype=torch.float64) / e ** 0.5
        ls0 = torch.tensor(1.5, dtype=torch.float64)
        enc = lambda x, w: torch.nn.functional.normalize(x @ w, dim=-1)       # noqa: E731
        sl = slice(rank * n, (rank + 1) * n)
        W, V, ls = (torch.nn.Parameter(t.clone()) for t in (W0, V0, ls0))
        loss = _GlobalInfoNCEFn.apply(_oracle_shard, None, enc(Y[sl], V), enc(X[sl], W), ls)
        loss.backward()
        P.average_gradients([W, V, ls])                                          # what DDP does
        Wf, Vf, lsf = (t.clone().requires_grad_(True) for t in (W0, V0, ls0))
        full = O.clip_loss((enc(Y, Vf) @ enc(X, Wf).t()) * lsf.exp())
        full.backward()
        lt = torch.tensor([loss.item()], dtype=torch.float64)
        dist.all_reduce(lt)
        assert abs(lt.item() / world - full.item()) < 1e-12                      # mean of the rank losses = the global loss
        for a, b in ((W, Wf), (V, Vf), (ls, lsf)):
            assert float((a.grad - b.grad).abs().max()) < 1e-11
        # no gradient requested: forward only, nothing saved, no second collective
        with torch.no_grad():
            l2 = _GlobalInfoNCEFn.apply(_oracle_shard, None, enc(Y[sl], V0), enc(X[sl], W0), ls0)
        assert abs(l2.item() - loss.item()) < 1e-12
        q.put((rank, "ok"))
    except Exception as ex:  # pragma: no cover
        import traceback
        q.put((rank, repr(ex) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,e", [(4, 8), (9, 16)])
def test_global_scope_autograd_path_under_ddp_averaging_world2_gloo(n, e):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scope_worker, args=(r, world, port, n, e, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def _ddp_worker(rank, world, port, ckpt, q):
    """The drop-in CLIPApp (contrastive_scope=global) under REAL DistributedDataParallel, as the reference Trainer wraps it
    (trainer.py:101-108: find_unused_parameters=True) -- tower encodes and the rank-local InfoNCE shard stood in for by the
    oracle.  Every rank's averaged gradients must equal the single-process gradients of the reference loss on the
    concatenated batch."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from easynlp_amd.appzoo.clip import model as CM

        def oracle_encode(self, pixel_values=None, input_ids=None, token_type_ids=None, attention_mask=None, pack_hint=None):
            sd = {n: p for n, p in self.chinese_clip.named_parameters()}
            return (O.encode_image(sd, self.raw_config, pixel_values) if pixel_values is not None else None,
                    O.encode_text(sd, self.raw_config, input_ids) if input_ids is not None else None)
        CM.CLIPApp.encode = oracle_encode
        CM.fused_infonce_shard = lambda eng, *a: _oracle_shard(*a)

        class OracleSimilarity:          # (the rank's own [n, n] logits the global-scope forward hands out for logging, detached)
            apply = staticmethod(lambda txt, img, ls: (txt @ img.t()) * ls.exp())
        CM._SimilarityFn = OracleSimilarity
        cfg = O.CONFIGS["tiny"]
        n = 3
        px, ids = O.make_inputs(cfg, world * n, 16, 21)
        app = CM.CLIPApp(ckpt, user_defined_parameters={"contrastive_scope": "global"})
        app.train()
        ddp = torch.nn.parallel.DistributedDataParallel(app, broadcast_buffers=False, find_unused_parameters=True)
        sl = slice(rank * n, (rank + 1) * n)
        out = ddp({"pixel_values": px[sl].clone(), "input_ids": ids[sl].clone()})
        loss = app.compute_loss(out, [])["loss"]
        loss.backward()
        # single process, whole batch, reference math
        sd = {k: v.detach().clone().requires_grad_(True) for k, v in app.chinese_clip.named_parameters()}
        full = O.clip_loss(O.clip_forward(sd, cfg, px, ids)["logits_per_text"])
        full.backward()
        lt = torch.tensor([loss.item()], dtype=torch.float64)
        dist.all_reduce(lt)
        assert abs(lt.item() / world - full.item()) < 1e-5
        checked = 0
        for name, p in app.chinese_clip.named_parameters():
            ref = sd[name].grad
            if ref is None:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, name      # the unused pooler
                continue
            assert float((p.grad - ref).norm()) <= 2e-4 * float(ref.norm()) + 1e-7, name
            checked += 1
        assert checked >= 30
        q.put((rank, "ok"))
    except Exception as ex:  # pragma: no cover
        import traceback
        q.put((rank, repr(ex) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_dropin_app_global_scope_under_real_ddp_world2_gloo(tmp_path):
    from oracle import ref_harness as R
    cfg = O.CONFIGS["tiny"]
    R.write_checkpoint_dir(str(tmp_path), cfg, O.make_state_dict(cfg, 9))
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


@pytest.mark.parametrize("world,n,e", [(2, 3, 8), (2, 16, 32), (4, 5, 16), (8, 3, 16)])
def test_global_contrastive_exchange_world2_gloo(world, n, e):
    """(world 4: ranks 2 and 3 sit at offsets the two-rank run never forms; world 8: the north star's node)"""
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


_RCCL_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["EZ_ROOT"])
lr = int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(lr)
dev = torch.device("cuda", lr)
dist.init_process_group(backend="nccl", device_id=dev)     # exactly what bench.py does for N > 1
w, r = dist.get_world_size(), dist.get_rank()
x = torch.arange(8, dtype=torch.float32, device=dev).reshape(2, 4) + r
g = torch.empty((w * 2, 4), dtype=torch.float32, device=dev)
dist.all_gather_into_tensor(g, x)
assert torch.equal(g[2 * r:2 * r + 2], x)
m = torch.empty((2, 4), dtype=torch.float32, device=dev)
dist.reduce_scatter_tensor(m, g.clone(), op=dist.ReduceOp.SUM)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
from easynlp_amd import parallel as P
p = torch.nn.Parameter(torch.zeros(5, device=dev)); p.grad = torch.ones(5, device=dev)
P.average_gradients([p])
dist.destroy_process_group()
print("RCCL_OK", w)
"""


@pytest.mark.gpu
def test_rccl_backend_single_rank_launch(tmp_path):
    """The launch line the driver uses for N > 1 (python -m torch.distributed.run ... one rank per GPU, backend nccl =
    RCCL), on the one GPU a test box has: process-group creation with device_id, the two collectives of the exchange
    step, the barrier / MAX all-reduce of bench.py's timing fence."""
    import subprocess
    import sys
    script = tmp_path / "rccl_smoke.py"
    script.write_text(_RCCL_SCRIPT)
    env = dict(os.environ, EZ_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "RCCL_OK 1" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])

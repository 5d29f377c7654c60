"""Operator-level parity (GPU): each HIP kernel, called through the C ABI,
against the CPU oracle's formula on the same seeded inputs.

Tolerances: the f32 path uses exact-f32 MFMA (an fmaf chain), so it differs
from the CPU only by summation order: |err| <= ~1e-5 * scale.  The bf16 path
rounds operands/outputs to bf16 (8 bits of mantissa): rel 2^-8 per rounding.
"""
import math

import pytest
import torch

from easynlp_amd import lib as L
from oracle import clip_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_err(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max())


GEMM_SHAPES = [
    (128, 128, 64), (256, 384, 128), (197, 768, 768), (1, 512, 768), (130, 136, 256),
    (394, 2304, 768), (64, 100, 64), (6, 6, 64), (257, 68, 192), (1000, 3072, 768), (520, 768, 3072),
]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_gemm_nt_plain(M, N, K, dtype):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g)
    if dtype == "bf16":
        a, b = a.bfloat16(), b.bfloat16()
    ref = a.double() @ b.double().t()
    c = L.op_gemm_nt(a.to(DEV), b.to(DEV))
    torch.cuda.synchronize()
    tol = 2e-6 if dtype == "f32" else 4e-3   # bf16: output rounding 2^-9 relative
    assert rel_err(c.float(), ref) < tol
    scale = math.sqrt(K)
    assert max_err(c.float(), ref) < (2e-5 if dtype == "f32" else 0.02) * scale


@pytest.mark.parametrize("act", [L.ACT_NONE, L.ACT_QUICKGELU, L.ACT_GELU_ERF])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(197, 768, 256), (70, 132, 64), (33, 7, 64)])
def test_gemm_nt_epilogues(M, N, K, dtype, act):
    g = torch.Generator().manual_seed(act + M)
    a = torch.randn(M, K, generator=g) * 0.5
    b = torch.randn(N, K, generator=g) * 0.2
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    if dtype == "bf16":
        a, b, res = a.bfloat16(), b.bfloat16(), res.bfloat16()
    z = a.double() @ b.double().t() + bias.double()
    if act == L.ACT_QUICKGELU:
        z = O.quick_gelu(z)
    elif act == L.ACT_GELU_ERF:
        z = O.gelu_erf(z)
    ref = z + res.double()
    c = L.op_gemm_nt(a.to(DEV), b.to(DEV), bias=bias.to(DEV), residual=res.to(DEV), act=act)
    torch.cuda.synchronize()
    assert max_err(c.float(), ref) < (3e-5 if dtype == "f32" else 0.06)
    # bf16 inputs, f32 output (projection / logits path)
    if dtype == "bf16":
        c32 = L.op_gemm_nt(a.to(DEV), b.to(DEV), bias=bias.to(DEV), act=act, out_f32=True)
        assert c32.dtype == torch.float32
        assert max_err(c32, z) < 2e-3


def test_gemm_nt_strided_rows_and_inplace_residual():
    # A rows strided (CLS rows of [B, L, D]), residual aliasing the output (x += ...)
    g = torch.Generator().manual_seed(5)
    B_, Lq, D = 9, 5, 128
    x = torch.randn(B_, Lq, D, generator=g).to(DEV)
    w = torch.randn(64, D, generator=g).to(DEV)
    a = x[:, 0, :]
    assert a.stride(0) == Lq * D
    lib = L.load()
    c = torch.empty(B_, 64, device=DEV)
    L.check(lib.ezclip_op_gemm_nt(a.data_ptr(), a.stride(0), w.data_ptr(), D, c.data_ptr(), 64, None, None, 0,
                                  B_, 64, D, 0, L.DTYPE_F32, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    assert max_err(c, a.cpu().double() @ w.cpu().double().t()) < 1e-4
    y = torch.randn(200, 128, generator=g).to(DEV)
    a2 = torch.randn(200, 64, generator=g).to(DEV)
    w2 = torch.randn(128, 64, generator=g).to(DEV)
    ref = y.cpu().double() + a2.cpu().double() @ w2.cpu().double().t()
    L.op_gemm_nt(a2, w2, residual=y, out=y)
    torch.cuda.synchronize()
    assert max_err(y, ref) < 1e-4


@pytest.mark.parametrize("rows,D", [(5, 128), (197, 768), (64, 1024), (33, 192), (4099, 768), (777, 520), (300, 256), (20001, 1024)])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("eps", [1e-5, 1e-12])
def test_layernorm(rows, D, dtype, eps):
    g = torch.Generator().manual_seed(rows + D)
    x = torch.randn(rows, D, generator=g) * 2 + 0.3
    w = 1 + 0.1 * torch.randn(D, generator=g)
    b = 0.1 * torch.randn(D, generator=g)
    if dtype == "bf16":
        x = x.bfloat16()
    ref = O.layer_norm(x.double(), w.double(), b.double(), eps)
    y, mean, rstd = L.op_layernorm(x.to(DEV), w.to(DEV), b.to(DEV), eps, want_stats=True)
    torch.cuda.synchronize()
    assert max_err(y.float(), ref) < (5e-6 if dtype == "f32" else 0.03)
    assert max_err(mean, x.double().mean(-1)) < 1e-5
    var = x.double().var(-1, unbiased=False)
    assert rel_err(rstd, torch.rsqrt(var + eps)) < 1e-5


def ref_attention(qkv, B_, Lq, H, key_bias=None):
    D = H * 64
    q, k, v = qkv.double().split(D, dim=-1)
    q = q.reshape(B_, Lq, H, 64).transpose(1, 2)
    k = k.reshape(B_, Lq, H, 64).transpose(1, 2)
    v = v.reshape(B_, Lq, H, 64).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * 0.125
    if key_bias is not None:
        s = s + key_bias.double().reshape(B_, 1, 1, Lq)
    p = torch.softmax(s, -1)
    ctx = (p @ v).transpose(1, 2).reshape(B_ * Lq, D)
    return ctx, torch.logsumexp(s, -1)


@pytest.mark.parametrize("B_,Lq,H", [(2, 197, 12), (3, 64, 2), (2, 26, 3), (1, 1, 2), (2, 33, 2), (1, 257, 4), (2, 288, 1)])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("masked", [False, True])
def test_attention(B_, Lq, H, dtype, masked):
    g = torch.Generator().manual_seed(B_ * 1000 + Lq + H)
    qkv = torch.randn(B_ * Lq, 3 * H * 64, generator=g) * 1.5
    kb = None
    if masked:
        lens = torch.randint(1, Lq + 1, (B_,), generator=g)
        lens[0] = Lq
        kb = torch.zeros(B_, Lq)
        for i in range(B_):
            kb[i, lens[i]:] = -10000.0
        if B_ > 1 and dtype == "bf16":  # (f32 cannot hold a score next to -10000: the reference itself is only good to 1e-3 there)
            kb[1, :] = -10000.0         # every key masked: the bias is part of the log-sum-exp too (BERT on an all-pad sentence)
        kb = kb.reshape(-1)
    if dtype == "bf16":
        qkv = qkv.bfloat16()
    ref, ref_lse = ref_attention(qkv, B_, Lq, H, kb)
    ctx, lse = L.op_attention(qkv.to(DEV), B_, Lq, H, key_bias=None if kb is None else kb.to(DEV), want_lse=True)
    torch.cuda.synchronize()
    assert max_err(ctx.float(), ref) < (2e-5 if dtype == "f32" else 0.03)
    assert max_err(lse, ref_lse) < (1e-4 if dtype == "f32" else 0.05)


@pytest.mark.parametrize("Lq,masked,causal", [(300, False, False), (400, True, False), (512, False, False), (512, True, False),
                                              (512, False, True), (330, True, True)])
def test_attention_f32_long_sequences_walk_the_keys_in_blocks(Lq, masked, causal):
    """f32 (and bf16 beyond 576 tokens): K and V^T of a head no longer fit the 160 KiB of LDS above 288 tokens -- BERT goes
    to 512 positions (modeling_bert.py max_position_embeddings).  attn_fwd_chunked_kernel stages the keys in blocks with
    an online softmax across blocks; two query blocks per wave at 512."""
    B_, H = 2, 2
    g = torch.Generator().manual_seed(Lq)
    D = H * 64
    qkv = torch.randn(B_ * Lq, 3 * D, generator=g) * 1.5
    kb = None
    if masked:
        lens = torch.tensor([Lq, 290])
        kb = torch.zeros(B_, Lq)
        for i in range(B_):
            kb[i, lens[i]:] = -10000.0
        kb = kb.reshape(-1)
    q, k, v = [t.double().reshape(B_, Lq, H, 64).transpose(1, 2) for t in qkv.split(D, dim=-1)]
    s_ = (q @ k.transpose(-1, -2)) * 0.125
    if kb is not None:
        s_ = s_ + kb.double().reshape(B_, 1, 1, Lq)
    if causal:
        s_ = s_.masked_fill(torch.triu(torch.ones(Lq, Lq, dtype=torch.bool), 1), float("-inf"))
    ref = (torch.softmax(s_, -1) @ v).transpose(1, 2).reshape(B_ * Lq, D)
    ref_lse = torch.logsumexp(s_, -1)
    ctx, lse = L.op_attention(qkv.to(DEV), B_, Lq, H, key_bias=None if kb is None else kb.to(DEV), want_lse=True, causal=causal)
    torch.cuda.synchronize()
    assert max_err(ctx, ref) < 2e-5
    assert max_err(lse, ref_lse) < 1e-4


def test_attention_bf16_long_sequence_512():
    B_, Lq, H = 1, 512, 2
    g = torch.Generator().manual_seed(11)
    qkv = (torch.randn(B_ * Lq, 3 * H * 64, generator=g)).bfloat16()
    ref, _ = ref_attention(qkv, B_, Lq, H)
    ctx = L.op_attention(qkv.to(DEV), B_, Lq, H)
    torch.cuda.synchronize()
    assert max_err(ctx.float(), ref) < 0.03


@pytest.mark.parametrize("n,e", [(6, 64), (8, 512), (100, 128), (256, 512), (1000, 512)])
def test_similarity_and_infonce(n, e):
    lib = L.load()
    g = torch.Generator().manual_seed(n)
    t = torch.nn.functional.normalize(torch.randn(n, e, generator=g), dim=-1)
    i = torch.nn.functional.normalize(t + 0.8 * torch.randn(n, e, generator=g), dim=-1)
    ls = torch.tensor(math.log(1 / 0.07))
    S = L.similarity(t.to(DEV), i.to(DEV), ls.to(DEV))
    ref = (t.double() @ i.double().t()) * ls.double().exp()
    assert max_err(S, ref) < 2e-4
    loss = torch.empty((), device=DEV)
    scratch = torch.empty(4 * n, device=DEV)
    L.check(lib.ezclip_infonce_from_logits(S.data_ptr(), n, loss.data_ptr(), scratch.data_ptr(), L.stream_ptr()))
    ref_loss = O.clip_loss(ref)
    assert abs(loss.item() - ref_loss.item()) < 1e-4 * max(1.0, abs(ref_loss.item()))
    # d loss / d logits
    Sr = ref.clone().requires_grad_(True)
    O.clip_loss(Sr).backward()
    d = torch.empty_like(S)
    gout = torch.tensor(1.0, device=DEV)
    L.check(lib.ezclip_infonce_from_logits_bwd(S.data_ptr(), n, gout.data_ptr(), d.data_ptr(), scratch.data_ptr(),
                                               L.stream_ptr()))
    assert max_err(d, Sr.grad) < 1e-5
    # recall ranks == the reference evaluator's sort loop
    from easynlp_amd.appzoo.clip.evaluator import recall_at_k
    (mean, r1, r5, r10), _ = recall_at_k(t.to(DEV), i.to(DEV))
    want = O.recall_at_k(t, i)
    assert abs(r1 - want[1]) < 1e-12 and abs(r5 - want[2]) < 1e-12 and abs(r10 - want[3]) < 1e-12


def test_recall_ranks_in_row_blocks_equal_the_sort_loop():
    """ezclip_recall_ranks_rows: a block of queries at a time ([rows, n] scratch instead of n x n) gives the ranks of the
    reference evaluator's descending stable sort (evaluator.py:53-61), ties included (duplicated gallery rows)."""
    from easynlp_amd.appzoo.clip.evaluator import recall_ranks
    g = torch.Generator().manual_seed(11)
    n, e = 517, 64
    t = torch.nn.functional.normalize(torch.randn(n, e, generator=g), dim=-1)
    v = torch.nn.functional.normalize(t + 0.8 * torch.randn(n, e, generator=g), dim=-1)
    v[5] = v[300]
    v[301] = v[300]                                   # exact ties in every row
    sim = t.double() @ v.double().t()
    want = torch.empty(n, dtype=torch.int64)
    for i in range(n):
        order = torch.sort(sim[i].float(), descending=True, stable=True).indices
        want[i] = int((order == i).nonzero()[0, 0])
    one = recall_ranks(t.to(DEV), v.to(DEV), block_rows=n)
    lib = L.load()
    full = torch.empty(n, dtype=torch.int32, device=DEV)
    scratch = torch.empty(n * n, device=DEV)
    td, vd = t.to(DEV), v.to(DEV)
    L.check(lib.ezclip_recall_ranks(td.data_ptr(), vd.data_ptr(), n, e, full.data_ptr(), scratch.data_ptr(), L.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(one, full)
    # (the GPU similarity is exact-f32 MFMA, the sort above ran on the f32 rounding of an f64 product: identical except
    # where two DIFFERENT scores collide after rounding -- none at this size; the constructed exact ties must agree)
    assert torch.equal(one.cpu().long(), want)
    for block in (1, 7, 64, 200, 516):
        assert torch.equal(recall_ranks(td, vd, block_rows=block), full), block
        assert torch.equal(recall_ranks(td, vd, block_rows=block, materialise=True), full), block


@pytest.mark.parametrize("n,e", [(517, 64), (128, 512), (1000, 512), (2500, 128)])
def test_fused_recall_ranks_both_directions(n, e):
    """ezclip_recall_paired_scores + ezclip_recall_ranks_fused: the similarity tile is compared in the registers of
    the kernel that computes it.  Text -> image ranks equal the materialising form's bit for bit (same kernel, same scores) and the
    oracle's stable sort; the image -> text ranks of the same sweep equal the oracle's column sort; exact ties (duplicated gallery
    rows AND duplicated queries) fall by index in both directions; any split into query blocks gives the same counts."""
    from easynlp_amd.appzoo.clip.evaluator import recall_ranks, recall_at_k
    g = torch.Generator().manual_seed(n + e)
    t = torch.nn.functional.normalize(torch.randn(n, e, generator=g), dim=-1)
    v = torch.nn.functional.normalize(t + 0.8 * torch.randn(n, e, generator=g), dim=-1)
    v[5] = v[100]; v[101] = v[100]; t[7] = t[90]; t[91] = t[90]
    td, vd = t.to(DEV), v.to(DEV)
    t2i, i2t = recall_ranks(td, vd, both_directions=True)
    assert torch.equal(t2i, recall_ranks(td, vd, materialise=True))
    assert torch.equal(t2i, recall_ranks(td, vd))
    # oracle on the GPU's own f32 similarity (ezclip_similarity: the scores the ranks are defined on); the f64 -> f32 CPU product
    # differs from it in the last bit here and there, which moves ranks only where two different scores are that close
    ls = torch.zeros((), device=DEV)
    sim = L.similarity(td, vd, ls).cpu()
    idx = torch.arange(n)
    d = sim.diagonal()
    want_r = ((sim > d[:, None]) | ((sim == d[:, None]) & (idx[None, :] < idx[:, None]))).sum(1)
    want_c = ((sim > d[None, :]) | ((sim == d[None, :]) & (idx[:, None] < idx[None, :]))).sum(0)
    assert torch.equal(t2i.cpu().long(), want_r)
    assert torch.equal(i2t.cpu().long(), want_c)
    if n <= 600:
        o_r, o_c = O.recall_ranks(t.double(), v.double())
        assert (t2i.cpu().long() != o_r).sum() <= 2 and (i2t.cpu().long() != o_c).sum() <= 2
        for i in (5, 100, 101, 7, 90, 91):                     # the constructed ties: exact in f32 and f64 alike
            assert int(t2i[i]) == int(o_r[i]) and int(i2t[i]) == int(o_c[i]), i
    for block in (1, 100, 129, n - 1):
        a, b = recall_ranks(td, vd, block_rows=block, both_directions=True)
        assert torch.equal(a, t2i) and torch.equal(b, i2t), block
    (fwd, _), (bwd, _) = recall_at_k(td, vd, both_directions=True)
    assert abs(fwd[1] - float((want_r < 1).sum()) / n) < 1e-12 and abs(bwd[3] - float((want_c < 10).sum()) / n) < 1e-12


@pytest.mark.parametrize("n,N,off,e", [(4, 12, 4, 64), (8, 8, 0, 128), (32, 128, 64, 512), (100, 300, 200, 64),
                                       (1024, 8192, 3072, 512)])
def test_infonce_fused_shard(n, N, off, e):
    """Fused loss + gradients of one rank's shard == autograd of the oracle's global loss share.  The last case is the
    north star's exchange step: rank 3 of 8, 1024 pairs per GPU, 8192 in the global batch."""
    lib = L.load()
    g = torch.Generator().manual_seed(N + off)
    t = torch.nn.functional.normalize(torch.randn(N, e, generator=g), dim=-1)
    i = torch.nn.functional.normalize(t + torch.randn(N, e, generator=g), dim=-1)
    ls = torch.tensor(2.0)
    td, idd, lsd = (x.double().clone().requires_grad_(True) for x in (t, i, ls))
    ref = O.global_clip_loss_rank(td, idd, lsd, off // n, n) if off % n == 0 else None
    if ref is None:
        pytest.skip("offset must be a multiple of n in this test")
    ref.backward()
    wsb = lib.ezclip_infonce_workspace_bytes(n, N, e)
    ws = L.alloc_bytes(wsb, DEV)
    loss = torch.empty((), device=DEV)
    dT = torch.empty(N, e, device=DEV)
    dI = torch.empty(N, e, device=DEV)
    dls = torch.empty((), device=DEV)
    tg, ig, lsg = t.to(DEV), i.to(DEV), ls.to(DEV)   # keep the device copies alive across the call
    L.check(lib.ezclip_infonce_fused(tg.data_ptr(), ig.data_ptr(), n, N, off, e, lsg.data_ptr(),
                                     1.0, loss.data_ptr(), dT.data_ptr(), dI.data_ptr(), dls.data_ptr(),
                                     ws.data_ptr(), ws.numel(), L.stream_ptr()))
    torch.cuda.synchronize()
    assert abs(loss.item() - ref.item()) < 1e-4 * max(1, abs(ref.item()))
    assert max_err(dT, td.grad) < 1e-5
    assert max_err(dI, idd.grad) < 1e-5
    assert abs(dls.item() - lsd.grad.item()) < 1e-4 * max(1, abs(lsd.grad.item()))


@pytest.mark.parametrize("split", [1, 0])
@pytest.mark.parametrize("n,N,off,e", [(8, 8, 0, 128), (5, 20, 10, 128), (100, 300, 200, 256), (32, 128, 64, 512), (64, 512, 128, 768),
                                       (200, 200, 0, 512), (1024, 1024, 0, 512), (1024, 8192, 3072, 512), (96, 768, 672, 1024)])
def test_infonce_tiled(n, N, off, e, split):
    """csrc/nce.hip: loss and gradients of one rank's shard WITHOUT the [n, N] logit blocks, against autograd of the oracle's
    global loss share in float64.  split = 1 (operands as bf16 hi + lo, three MFMA products): the f32 pipeline's bounds -- loss
    1e-5, gradients 1e-5 absolute / 2e-4 of the largest entry; split = 0 (embeddings rounded to bf16 once): bf16 bounds.  Ragged
    tiles (n, N, off not multiples of 64 / 128), every instantiated width, single-GPU (n = N) and the north star's rank 3 of 8."""
    lib = L.load()
    g = torch.Generator().manual_seed(N + off + e)
    t = torch.nn.functional.normalize(torch.randn(N, e, generator=g), dim=-1)
    i = torch.nn.functional.normalize(t + torch.randn(N, e, generator=g), dim=-1)
    ls = torch.tensor(2.0)
    td, idd, lsd = (x.double().clone().requires_grad_(True) for x in (t, i, ls))
    ref = O.global_clip_loss_rank(td, idd, lsd, off // n, n)
    ref.backward()
    wsb = lib.ezclip_infonce_tiled_workspace_bytes(n, N, e)
    assert 0 < wsb < 16 * (n + N) * e * 4 + (1 << 22), wsb          # O((n + N) e): no n x N term
    ws = L.alloc_bytes(wsb, DEV)
    tg, ig, lsg = t.to(DEV), i.to(DEV), ls.to(DEV)
    out = {}
    for grads in (False, True):
        loss = torch.full((), float("nan"), device=DEV)
        dT = torch.full((N, e), float("nan"), device=DEV) if grads else None
        dI = torch.full((N, e), float("nan"), device=DEV) if grads else None
        dls = torch.full((), float("nan"), device=DEV) if grads else None
        L.check(lib.ezclip_infonce_tiled(tg.data_ptr(), ig.data_ptr(), n, N, off, e, lsg.data_ptr(), 1.0, split, loss.data_ptr(),
                                         L.ptr(dT), L.ptr(dI), L.ptr(dls), ws.data_ptr(), ws.numel(), L.stream_ptr()))
        torch.cuda.synchronize()
        out[grads] = (loss, dT, dI, dls)
    loss, dT, dI, dls = out[True]
    assert out[False][0].item() == loss.item()            # forward-only call: the same kernels, the same bits
    tol_l = 1e-5 if split else 2e-3
    assert abs(loss.item() - ref.item()) < tol_l * max(1, abs(ref.item())), (loss.item(), ref.item())
    for got, want in ((dT, td.grad), (dI, idd.grad)):
        scale = float(want.abs().max())
        tol = min(1e-5, 2e-4 * scale) if split else 2e-2 * scale
        assert max_err(got, want) < tol, (max_err(got, want), scale)
    assert abs(dls.item() - lsd.grad.item()) < (1e-4 if split else 5e-3) * max(1, abs(lsd.grad.item()))
    # bit-reproducible (fixed-order chunk sums, no float atomics)
    dT2 = torch.empty_like(dT); dI2 = torch.empty_like(dI); loss2 = torch.empty_like(loss); dls2 = torch.empty_like(dls)
    L.check(lib.ezclip_infonce_tiled(tg.data_ptr(), ig.data_ptr(), n, N, off, e, lsg.data_ptr(), 1.0, split, loss2.data_ptr(),
                                     dT2.data_ptr(), dI2.data_ptr(), dls2.data_ptr(), ws.data_ptr(), ws.numel(), L.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(dT, dT2) and torch.equal(dI, dI2) and loss.item() == loss2.item() and dls.item() == dls2.item()
    # the materialising exact-f32 path of rounds 1-2 (ezclip_infonce_fused) against the tiled one, grad_scale 0.5
    if split:
        w2 = L.alloc_bytes(lib.ezclip_infonce_workspace_bytes(n, N, e), DEV)
        lo, a, b, c = torch.empty((), device=DEV), torch.empty(N, e, device=DEV), torch.empty(N, e, device=DEV), torch.empty((), device=DEV)
        L.check(lib.ezclip_infonce_fused(tg.data_ptr(), ig.data_ptr(), n, N, off, e, lsg.data_ptr(), 0.5, lo.data_ptr(),
                                         a.data_ptr(), b.data_ptr(), c.data_ptr(), w2.data_ptr(), w2.numel(), L.stream_ptr()))
        torch.cuda.synchronize()
        assert abs(lo.item() - loss.item()) < 2e-5 * max(1, abs(loss.item()))
        assert max_err(a, 0.5 * dT.double()) < 1e-5 and max_err(b, 0.5 * dI.double()) < 1e-5
        assert abs(c.item() - 0.5 * dls.item()) < 1e-4 * max(1, abs(dls.item()))


# ----------------------------------------------------------------------------- backward ops

@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (1000, 768, 768), (197 * 3, 2304, 768), (130, 136, 264), (8, 64, 512),
                                   (5000, 128, 512)])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_gemm_tn(M, N, K, dtype):
    lib = L.load()
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, N, generator=g)
    b = torch.randn(M, K, generator=g)
    if dtype == "bf16":
        a, b = a.bfloat16(), b.bfloat16()
    c0 = torch.randn(N, K, generator=g)
    ref = c0.double() + a.double().t() @ b.double()
    ad, bd, cd = a.to(DEV), b.to(DEV), c0.to(DEV)
    dt = L.DTYPE_BF16 if dtype == "bf16" else L.DTYPE_F32
    L.check(lib.ezclip_op_gemm_tn(ad.data_ptr(), N, bd.data_ptr(), K, cd.data_ptr(), K, M, N, K, 1, dt, L.stream_ptr()))
    torch.cuda.synchronize()
    assert max_err(cd, ref) < (3e-5 if dtype == "f32" else 2e-3) * math.sqrt(M) * 3
    assert rel_err(cd, ref) < (2e-6 if dtype == "f32" else 1e-3)
    # overwrite mode
    c1 = torch.full((N, K), 7.0, device=DEV)
    L.check(lib.ezclip_op_gemm_tn(ad.data_ptr(), N, bd.data_ptr(), K, c1.data_ptr(), K, M, N, K, 0, dt, L.stream_ptr()))
    assert rel_err(c1, a.double().t() @ b.double()) < (2e-6 if dtype == "f32" else 1e-3)


@pytest.mark.parametrize("rows,D", [(7, 128), (394, 768), (4096, 768)])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_layernorm_bwd(rows, D, dtype):
    lib = L.load()
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, D, generator=g) * 1.7 + 0.2
    dy = torch.randn(rows, D, generator=g)
    w = 1 + 0.1 * torch.randn(D, generator=g)
    b = 0.1 * torch.randn(D, generator=g)
    if dtype == "bf16":
        x, dy = x.bfloat16(), dy.bfloat16()
    xd = x.double().requires_grad_(True)
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    O.layer_norm(xd, wd, bd, 1e-5).backward(dy.double())
    xg, dyg = x.to(DEV), dy.to(DEV)
    y, mean, rstd = L.op_layernorm(xg, w.to(DEV), b.to(DEV), 1e-5, want_stats=True)
    dx = torch.empty_like(xg)
    dg = torch.zeros(D, device=DEV)
    db = torch.zeros(D, device=DEV)
    dt = L.DTYPE_BF16 if dtype == "bf16" else L.DTYPE_F32
    wg = w.to(DEV)
    L.check(lib.ezclip_op_layernorm_bwd(xg.data_ptr(), dyg.data_ptr(), wg.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                        dx.data_ptr(), dg.data_ptr(), db.data_ptr(), rows, D, dt, L.stream_ptr()))
    torch.cuda.synchronize()
    assert max_err(dx.float(), xd.grad) < (2e-5 if dtype == "f32" else 0.03)
    assert rel_err(dg, wd.grad) < (1e-5 if dtype == "f32" else 1e-3)
    assert rel_err(db, bd.grad) < (1e-5 if dtype == "f32" else 1e-3)


@pytest.mark.parametrize("B_,Lq,H", [(2, 197, 3), (3, 64, 2), (2, 26, 1), (1, 1, 1), (1, 257, 2), (3, 264, 2), (2, 272, 1), (2, 273, 1),
                                     (1, 400, 1)])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("masked", [False, True])
def test_attention_bwd(B_, Lq, H, dtype, masked):
    lib = L.load()
    g = torch.Generator().manual_seed(B_ * 100 + Lq + H)
    D = H * 64
    qkv = torch.randn(B_ * Lq, 3 * D, generator=g)
    dctx = torch.randn(B_ * Lq, D, generator=g)
    kb = None
    if masked:
        lens = torch.randint(1, Lq + 1, (B_,), generator=g)
        lens[0] = Lq
        kb = torch.zeros(B_, Lq)
        for i in range(B_):
            kb[i, lens[i]:] = -10000.0
        if B_ > 1 and dtype == "bf16":  # (f32 cannot hold a score next to -10000: the reference itself is only good to 1e-3 there)
            kb[1, :] = -10000.0         # every key masked: the bias is part of the log-sum-exp too (BERT on an all-pad sentence)
        kb = kb.reshape(-1)
    if dtype == "bf16":
        qkv, dctx = qkv.bfloat16(), dctx.bfloat16()
    qd = qkv.double().requires_grad_(True)
    ref, _ = ref_attention(qd, B_, Lq, H, kb)
    ref.backward(dctx.double())
    qg, dg_, kbg = qkv.to(DEV), dctx.to(DEV), (None if kb is None else kb.to(DEV))
    ctx, lse = L.op_attention(qg, B_, Lq, H, key_bias=kbg, want_lse=True)
    dqkv = torch.zeros_like(qg)
    esz = qg.element_size()
    dt = L.DTYPE_BF16 if dtype == "bf16" else L.DTYPE_F32
    base, dbase = qg.data_ptr(), dqkv.data_ptr()
    L.check(lib.ezclip_op_attention_bwd(base, base + D * esz, base + 2 * D * esz, 3 * D, ctx.data_ptr(), dg_.data_ptr(), D,
                                        L.ptr(kbg), lse.data_ptr(), dbase, dbase + D * esz, dbase + 2 * D * esz,
                                        B_, Lq, H, dt, None, L.stream_ptr()))
    torch.cuda.synchronize()
    scale = float(qd.grad.abs().max())
    assert max_err(dqkv.float(), qd.grad) < (3e-5 if dtype == "f32" else 0.04) * max(1.0, scale)
    assert rel_err(dqkv.float(), qd.grad) < (1e-5 if dtype == "f32" else 0.02)


@pytest.mark.parametrize("B_,Lq,H", [(5, 197, 3), (7, 64, 2), (3, 26, 1), (2, 256, 2), (2, 257, 1), (3, 270, 2)])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("masked", [False, True])
def test_attention_bwd_projection_bias_gradients(B_, Lq, H, dtype, masked):
    """The q / k / v bias gradients the backward hands out with dq / dk / dv (the fused short-sequence kernel forms them as
    scale K^T c, scale Q^T r and dO^T 1 -- DESIGN.md 4.1b) against the column sums of the reference's dqkv, accumulated
    into what the buffers held."""
    lib = L.load()
    g = torch.Generator().manual_seed(B_ * 100 + Lq + H)
    D = H * 64
    qkv = torch.randn(B_ * Lq, 3 * D, generator=g)
    dctx = torch.randn(B_ * Lq, D, generator=g)
    kb = None
    if masked:
        lens = torch.randint(1, Lq + 1, (B_,), generator=g)
        lens[0] = Lq
        kb = torch.zeros(B_, Lq)
        for i in range(B_):
            kb[i, lens[i]:] = -10000.0
        if B_ > 1 and dtype == "bf16":  # (f32 cannot hold a score next to -10000: the reference itself is only good to 1e-3 there)
            kb[1, :] = -10000.0         # every key masked: the bias is part of the log-sum-exp too (BERT on an all-pad sentence)
        kb = kb.reshape(-1)
    if dtype == "bf16":
        qkv, dctx = qkv.bfloat16(), dctx.bfloat16()
    qd = qkv.double().requires_grad_(True)
    ref, _ = ref_attention(qd, B_, Lq, H, kb)
    ref.backward(dctx.double())
    ref_db = qd.grad.sum(0)
    qg, dg_, kbg = qkv.to(DEV), dctx.to(DEV), (None if kb is None else kb.to(DEV))
    ctx, lse = L.op_attention(qg, B_, Lq, H, key_bias=kbg, want_lse=True)
    dqkv = torch.zeros_like(qg)
    start = torch.randn(3 * D, generator=g)
    db = start.to(DEV)
    scratch = torch.empty(B_ * 3 * D, dtype=torch.float32, device=DEV)
    esz = qg.element_size()
    dt = L.DTYPE_BF16 if dtype == "bf16" else L.DTYPE_F32
    base, dbase, bb = qg.data_ptr(), dqkv.data_ptr(), db.data_ptr()
    L.check(lib.ezclip_op_attention_bwd_bias(base, base + D * esz, base + 2 * D * esz, 3 * D, ctx.data_ptr(), dg_.data_ptr(), D,
                                             L.ptr(kbg), lse.data_ptr(), dbase, dbase + D * esz, dbase + 2 * D * esz,
                                             bb, bb + 4 * D, bb + 8 * D, scratch.data_ptr(), B_, Lq, H, dt, None, L.stream_ptr()))
    torch.cuda.synchronize()
    got = db.cpu().double() - start.double()
    scale = float(ref_db.abs().max())
    # q and v parts: sums of B*L terms of size ~1; the k part is zero in exact arithmetic (rows of dS sum to zero), so only
    # an absolute bound means anything there
    tol = (2e-4 if dtype == "f32" else 0.02) * max(1.0, scale)
    assert float((got - ref_db).abs().max()) < tol, (float((got - ref_db).abs().max()), scale)
    assert max_err(dqkv.float(), qd.grad) < (3e-5 if dtype == "f32" else 0.04) * max(1.0, float(qd.grad.abs().max()))


@pytest.mark.parametrize("variant", [0, 1, 3])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (1000, 768, 768), (513, 512, 128), (2049, 256, 3072), (300, 100, 64)])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_gemm_nt_tile_variants(M, N, K, dtype, variant):
    """Both tile shapes (128x128 / 256x256) on ragged M/N, with the full epilogue."""
    lib = L.load()
    g = torch.Generator().manual_seed(M + N + K + variant)
    a = torch.randn(M, K, generator=g) * 0.5
    b = torch.randn(N, K, generator=g) * 0.2
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    if dtype == "bf16":
        a, b, res = a.bfloat16(), b.bfloat16(), res.bfloat16()
    ref = O.quick_gelu(a.double() @ b.double().t() + bias.double()) + res.double()
    L.check(lib.ezclip_debug_set(0, variant))
    try:
        c = L.op_gemm_nt(a.to(DEV), b.to(DEV), bias=bias.to(DEV), residual=res.to(DEV), act=L.ACT_QUICKGELU)
        torch.cuda.synchronize()
    finally:
        L.check(lib.ezclip_debug_set(0, -1))
    assert max_err(c.float(), ref) < (5e-5 if dtype == "f32" else 0.06) * max(1.0, math.sqrt(K) / 8)


# ----------------------------------------------------------------------------- 256x256 8-phase kernels (gemm8p.hip)

@pytest.mark.parametrize("M,N,K", [(2049, 256, 3072), (1000, 768, 768), (512, 512, 256), (300, 256, 512), (4096, 1024, 384)])
@pytest.mark.parametrize("epi", ["plain", "bias", "bias_res_gelu", "res_qgelu_strided"])
def test_gemm_nt_8phase(M, N, K, epi):
    """The persistent 8-phase bf16 kernel (forced with variant 2) against the fp64 product and, bit for bit,
    against the 128x128 kernel (same accumulation order, same epilogue math): ragged M (bounds-checked DMA and
    stores), the minimum K (4 K-tiles), more tiles than CUs is covered by the model tests / gemm_bench."""
    lib = L.load()
    g = torch.Generator().manual_seed(M + N + K)
    wide = torch.randn(M, 2 * K, generator=g) * 0.5
    a = (wide[:, K:] if epi.endswith("strided") else wide[:, :K].contiguous()).bfloat16()
    b = (torch.randn(N, K, generator=g) * 0.2).bfloat16()
    bias = torch.randn(N, generator=g) if epi != "plain" else None
    res = torch.randn(M, N, generator=g).bfloat16() if "res" in epi else None
    act = L.ACT_GELU_ERF if "gelu" in epi and "qgelu" not in epi else (L.ACT_QUICKGELU if "qgelu" in epi else L.ACT_NONE)
    ref = a.double() @ b.double().t()
    if bias is not None:
        ref = ref + bias.double()
    if act == L.ACT_GELU_ERF:
        ref = O.gelu_erf(ref)
    elif act == L.ACT_QUICKGELU:
        ref = O.quick_gelu(ref)
    if res is not None:
        ref = ref + res.double()
    ad = a.to(DEV)
    if epi.endswith("strided"):
        ad = wide.bfloat16().to(DEV)[:, K:]
        assert ad.stride(0) == 2 * K
    outs = {}
    for variant in (0, 2):
        L.check(lib.ezclip_debug_set(0, variant))
        try:
            bd_ = b.to(DEV)
            biasd = None if bias is None else bias.to(DEV)
            resd = None if res is None else res.to(DEV)
            c = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
            L.check(lib.ezclip_op_gemm_nt(ad.data_ptr(), ad.stride(0), bd_.data_ptr(), K, c.data_ptr(), N,
                                          None if biasd is None else biasd.data_ptr(),
                                          None if resd is None else resd.data_ptr(), N if resd is not None else 0,
                                          M, N, K, act, L.DTYPE_BF16, 0, L.stream_ptr()))
            torch.cuda.synchronize()
            outs[variant] = c
        finally:
            L.check(lib.ezclip_debug_set(0, -1))
    assert max_err(outs[2].float(), ref) < 0.06 * max(1.0, math.sqrt(K) / 8)
    assert torch.equal(outs[0], outs[2]), "8-phase kernel differs from the 128x128 kernel"


@pytest.mark.parametrize("M,N,K", [(4096, 768, 768), (2100, 256, 512), (9000, 512, 256)])
@pytest.mark.parametrize("strided", [False, True])
def test_gemm_tn_8phase(M, N, K, strided):
    """Weight-gradient kernel with LDS transpose reads + split partials: fp64 reference, strided dY (a column
    slice of a packed qkv gradient), ragged M (rows past M read as zeros), accumulate / overwrite, and
    bit-reproducibility (fixed-order reduction instead of atomics)."""
    lib = L.load()
    g = torch.Generator().manual_seed(M + N + K)
    wide = torch.randn(M, 3 * N, generator=g).bfloat16()
    a = wide[:, N:2 * N] if strided else wide[:, :N].contiguous()
    b = torch.randn(M, K, generator=g).bfloat16()
    c0 = torch.randn(N, K, generator=g)
    prod = a.double().t() @ b.double()
    ad = wide.to(DEV)[:, N:2 * N] if strided else a.to(DEV)
    bd = b.to(DEV)
    outs = []
    for rep in range(2):
        cd = c0.to(DEV)
        L.check(lib.ezclip_op_gemm_tn(ad.data_ptr(), ad.stride(0), bd.data_ptr(), K, cd.data_ptr(), K, M, N, K, 1,
                                      L.DTYPE_BF16, L.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(cd)
    assert rel_err(outs[0], c0.double() + prod) < 1e-5
    assert torch.equal(outs[0], outs[1]), "weight gradient is not bit-reproducible"
    c1 = torch.full((N, K), 7.0, device=DEV)
    L.check(lib.ezclip_op_gemm_tn(ad.data_ptr(), ad.stride(0), bd.data_ptr(), K, c1.data_ptr(), K, M, N, K, 0,
                                  L.DTYPE_BF16, L.stream_ptr()))
    torch.cuda.synchronize()
    assert rel_err(c1, prod) < 1e-5


@pytest.mark.parametrize("Lq", [26, 77, 197])
@pytest.mark.parametrize("dtype,variant", [("f32", -1), ("bf16", -1), ("bf16", 0)])
def test_attention_causal_forward_and_backward(Lq, dtype, variant):
    """Causal mask of the CLIP text transformer (OPEN_CLIP.build_attention_mask, modeling_openclip.py:343-349: -inf above
    the diagonal) in every attention kernel: one-pass short kernels (bf16, variant -1), general two-pass kernels
    (f32, and bf16 with variant 0)."""
    B_, H = 2, 2
    g = torch.Generator().manual_seed(Lq)
    D = H * 64
    qkv = torch.randn(B_ * Lq, 3 * D, generator=g)
    dctx = torch.randn(B_ * Lq, D, generator=g)
    if dtype == "bf16":
        qkv, dctx = qkv.bfloat16(), dctx.bfloat16()
    qd = qkv.double().requires_grad_(True)
    q, k, v = [t.reshape(B_, Lq, H, 64).transpose(1, 2) for t in qd.split(D, dim=-1)]
    s = (q @ k.transpose(-1, -2)) * 0.125
    s = s + torch.full((Lq, Lq), float("-inf"), dtype=torch.float64).triu_(1)
    ref = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B_ * Lq, D)
    ref.backward(dctx.double())
    lib = L.load()
    L.check(lib.ezclip_debug_set(1, variant))
    try:
        ctx, lse = L.op_attention(qkv.to(DEV), B_, Lq, H, want_lse=True, causal=True)
        dqkv = L.op_attention_bwd(qkv.to(DEV), ctx, dctx.to(DEV), lse, B_, Lq, H, causal=True)
        # (the option is an argument of the call: the next call without it is a plain one)
        assert max_err(L.op_attention(qkv.to(DEV), B_, Lq, H).float(), ctx.float()) > 1e-3
    finally:
        L.check(lib.ezclip_debug_set(1, -1))
    assert max_err(ctx.float(), ref.detach()) < (2e-5 if dtype == "f32" else 3e-2)
    assert rel_err(dqkv.float(), qd.grad) < (1e-5 if dtype == "f32" else 2e-2)
    # the first token attends only to itself: ctx[0] = v[0]
    v0 = qkv[0, 2 * D:2 * D + 64].float()
    assert float((ctx[0, :64].float().cpu() - v0).abs().max()) < (1e-6 if dtype == "f32" else 1e-2)


@pytest.mark.parametrize("Lq", [8, 33, 64, 197, 300])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("compact_dq", [False, True])
def test_attention_cls_query_forward_and_backward(Lq, dtype, compact_dq):
    """One query per sample (the CLS row of a tower's last block): ctx, and dq / dk / dv for all keys, against torch fp64."""
    B_, H = 3, 2
    g = torch.Generator().manual_seed(Lq + 7)
    D = H * 64
    qkv = torch.randn(B_ * Lq, 3 * D, generator=g)
    dctx = torch.randn(B_, D, generator=g)
    kb = torch.zeros(B_, Lq)
    kb[1, Lq - 3:] = -10000.0
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    qkv, dctx = qkv.to(tdt), dctx.to(tdt)
    qd = qkv.double().requires_grad_(True)
    q, k, v = [t.reshape(B_, Lq, H, 64).transpose(1, 2) for t in qd.split(D, dim=-1)]
    s = (q[:, :, :1] @ k.transpose(-1, -2)) * 0.125 + kb.double().reshape(B_, 1, 1, Lq)
    ref = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B_, D)
    ref.backward(dctx.double())
    lib = L.load()
    dt = L.DTYPE_F32 if dtype == "f32" else L.DTYPE_BF16
    qg, dg_, kbg = qkv.to(DEV), dctx.to(DEV), kb.reshape(-1).to(DEV)
    esz = qg.element_size()
    base = qg.data_ptr()
    ctx = torch.empty((B_, D), dtype=tdt, device=DEV)
    L.check(lib.ezclip_op_attention_cls(base, Lq * 3 * D, base + D * esz, base + 2 * D * esz, 3 * D, L.ptr(kbg), L.ptr(ctx), D,
                                        B_, Lq, H, dt, L.stream_ptr()))
    dqkv = torch.full_like(qg, float("nan"))
    dq_c = torch.zeros((B_, D), dtype=tdt, device=DEV)
    db = dqkv.data_ptr()
    L.check(lib.ezclip_op_attention_cls_bwd(base, Lq * 3 * D, base + D * esz, base + 2 * D * esz, 3 * D, L.ptr(kbg), L.ptr(ctx),
                                            L.ptr(dg_), D, None if compact_dq else db, db + D * esz, db + 2 * D * esz,
                                            L.ptr(dq_c) if compact_dq else None, D, B_, Lq, H, dt, L.stream_ptr()))
    torch.cuda.synchronize()
    tol = 2e-5 if dtype == "f32" else 3e-2
    assert max_err(ctx.float(), ref.detach()) < tol
    gq, gk, gv = qd.grad.split(D, dim=-1)
    got = dqkv.float().cpu().double()
    assert rel_err(got[:, D:2 * D], gk) < (1e-5 if dtype == "f32" else 2e-2)
    assert rel_err(got[:, 2 * D:], gv) < (1e-5 if dtype == "f32" else 2e-2)
    if compact_dq:
        assert rel_err(dq_c.float().cpu().double(), gq.reshape(B_, Lq, D)[:, 0]) < (1e-5 if dtype == "f32" else 2e-2)
        assert torch.isnan(got[:, :D]).all()          # untouched
    else:
        assert rel_err(got[:, :D], gq) < (1e-5 if dtype == "f32" else 2e-2)      # (zeros outside row 0 on both sides)


# ----------------------------------------------------------------------------- round 4: polynomial erf-GELU outside its clamp

@pytest.mark.parametrize("variant", [0, 2])
def test_gelu_polynomial_tails(variant):
    """F.gelu of the bf16 pipeline (ezclip_common.h: Phi(x) ~ 0.5 + xc R(t), xc = clamp(x, +-4.2)) for |x| up to 50 through the
    epilogue of both bf16 GEMM kernels (x = A I^T exactly: identity weights).  The round-3 fit left Phi(-4.2) +- 7.5e-6 at the clamp,
    so gelu(x) = x Phi(x) was off by |x| * 2e-5 for EVERY x below it (-7.2e-5 at -12, -3.6e-4 at -50); the round-4 fit is
    exactly 0 / 1 there (tools/fit_gelu_poly.py): |error| <= 5.7e-5 absolute everywhere, x (0 +- 1e-7) below the clamp."""
    lib = L.load()
    M, K = 512, 256
    xs = torch.cat([torch.linspace(-50, -4.2, 120), torch.linspace(-4.2, 4.2, 272), torch.linspace(4.2, 50, 120)])
    a = xs.bfloat16().float()[:, None].repeat(1, K)                   # row i holds x_i in every column; B = I picks it up unchanged
    b = torch.eye(K)
    ad, bd_ = a.bfloat16().to(DEV), b.bfloat16().to(DEV)
    c = torch.empty((M, K), dtype=torch.bfloat16, device=DEV)
    L.check(lib.ezclip_debug_set(0, variant))
    try:
        L.check(lib.ezclip_op_gemm_nt(ad.data_ptr(), K, bd_.data_ptr(), K, c.data_ptr(), K, None, None, 0, M, K, K, L.ACT_GELU_ERF,
                                      L.DTYPE_BF16, 0, L.stream_ptr()))
        torch.cuda.synchronize()
    finally:
        L.check(lib.ezclip_debug_set(0, -1))
    x = a.bfloat16().double()[:, 0]
    got = c.double().cpu()
    assert float((got - got[:, :1]).abs().max()) == 0.0               # every column of a row saw the same x
    exact = O.gelu_erf(x)
    err = (got[:, 0] - exact).abs()
    bound = 5.7e-5 + 2.0 ** -8 * exact.abs()                          # the fit + bf16 rounding of the stored result
    worst = int(torch.argmax(err / bound))
    assert bool((err <= bound).all()), (float(x[worst]), float(got[worst, 0]), float(exact[worst]))
    far = x < -6.0
    assert float(got[far, 0].abs().max()) < 1e-5                      # round 3: -7e-5 ... -3.6e-4 here


@pytest.mark.parametrize("rows,cols", [(1, 1), (7, 7), (64, 64), (300, 300), (37, 90)])
def test_contrastive_loss_one_direction(rows, cols):
    """CLIPApp.contrastive_loss(logits) = F.cross_entropy(logits, arange(len(logits))) (appzoo/clip/model.py:154-155), value and
    gradient against torch's float64 evaluation; the transposed view of a square block (what clip_loss feeds it in the reference)
    and the sum of both directions = 2 x clip_loss."""
    from easynlp_amd.appzoo.clip import CLIPApp
    app = CLIPApp.__new__(CLIPApp)                                    # the method reads no state of the instance
    g = torch.Generator().manual_seed(rows * 131 + cols)
    s = (torch.randn(rows, cols, generator=g) * 4).to(DEV).requires_grad_(True)
    loss = CLIPApp.contrastive_loss(app, s)
    loss.backward()
    sref = s.detach().double().cpu().requires_grad_(True)
    want = torch.nn.functional.cross_entropy(sref, torch.arange(rows))
    want.backward()
    assert abs(loss.item() - want.item()) < 2e-6 * max(1.0, abs(want.item()))
    assert float((s.grad.double().cpu() - sref.grad).abs().max()) < 1e-6
    if rows == cols:
        st = s.detach().clone().requires_grad_(True)
        both = CLIPApp.contrastive_loss(app, st) + CLIPApp.contrastive_loss(app, st.T)
        both.backward()
        s2 = s.detach().clone().requires_grad_(True)
        fused = CLIPApp.clip_loss(app, s2)
        fused.backward()
        assert abs(both.item() / 2 - fused.item()) < 2e-6 * max(1.0, abs(fused.item()))
        assert float((st.grad / 2 - s2.grad).abs().max()) < 1e-6
    else:
        with pytest.raises(L.EzclipError):
            CLIPApp.contrastive_loss(app, s.detach().T.contiguous())   # more rows than columns: arange(rows) has no column
    with pytest.raises(L.EzclipError):
        CLIPApp.contrastive_loss(app, s.detach().cpu())


# ----------------------------------------------------------------------------- round 4: attention backward, every score tile once

@pytest.mark.parametrize("B_,Lq,H", [(5, 197, 3), (7, 64, 2), (3, 26, 1), (2, 256, 2), (4, 33, 2), (3, 224, 1)])
@pytest.mark.parametrize("masked", [False, True])
def test_attention_bwd_score_tile_once_against_two_pass_and_fp64(B_, Lq, H, masked):
    """The fused backward of round 4 (a wave owns a key block, dS goes through a wave-private LDS tile for the dQ product, dQ is
    summed over the waves in an LDS float image in a fixed, barrier-ordered rotation) against the two-pass kernel it replaces
    (ezclip_debug_set(11, 0)) and the float64 reference: dq / dk / dv and the three projection-bias gradients; two runs give the
    same bits (the rotation fixes the summation order)."""
    lib = L.load()
    g = torch.Generator().manual_seed(B_ * 1000 + Lq * 3 + H)
    D = H * 64
    qkv = torch.randn(B_ * Lq, 3 * D, generator=g).bfloat16()
    dctx = torch.randn(B_ * Lq, D, generator=g).bfloat16()
    kb = None
    if masked:
        lens = torch.randint(1, Lq + 1, (B_,), generator=g)
        lens[0] = Lq
        kb = torch.zeros(B_, Lq)
        for i in range(B_):
            kb[i, lens[i]:] = -10000.0
        if B_ > 1:
            kb[1, :] = -10000.0
        kb = kb.reshape(-1)
    qd = qkv.double().requires_grad_(True)
    ref, _ = ref_attention(qd, B_, Lq, H, kb)
    ref.backward(dctx.double())
    ref_db = qd.grad.sum(0)
    qg, dg_, kbg = qkv.to(DEV), dctx.to(DEV), (None if kb is None else kb.to(DEV))
    ctx, lse = L.op_attention(qg, B_, Lq, H, key_bias=kbg, want_lse=True)
    esz = 2
    base = qg.data_ptr()
    scratch = torch.empty(B_ * 3 * D, dtype=torch.float32, device=DEV)
    outs = {}
    try:
        for variant in (2, 0, 2):
            L.check(lib.ezclip_debug_set(11, variant))
            dqkv = torch.zeros_like(qg)
            db = torch.zeros(3 * D, device=DEV)
            dbase, bb = dqkv.data_ptr(), db.data_ptr()
            L.check(lib.ezclip_op_attention_bwd_bias(base, base + D * esz, base + 2 * D * esz, 3 * D, ctx.data_ptr(), dg_.data_ptr(), D,
                                                     L.ptr(kbg), lse.data_ptr(), dbase, dbase + D * esz, dbase + 2 * D * esz,
                                                     bb, bb + 4 * D, bb + 8 * D, scratch.data_ptr(), B_, Lq, H, L.DTYPE_BF16, None,
                                                     L.stream_ptr()))
            torch.cuda.synchronize()
            if variant == 2 and 2 in outs:
                assert torch.equal(outs[2][0], dqkv) and torch.equal(outs[2][1], db), "not bit-reproducible"
            outs[variant] = (dqkv, db)
    finally:
        L.check(lib.ezclip_debug_set(11, 1))
    scale = float(qd.grad.abs().max())
    for variant in (2, 0):
        got = outs[variant][0].float().cpu()
        assert max_err(got, qd.grad) < 0.04 * max(1.0, scale), variant
        assert rel_err(got, qd.grad) < 0.02, variant
    # the two kernels agree far inside that bound (same products, another summation order for dQ, then one bf16 rounding)
    assert float((outs[2][0].float() - outs[0][0].float()).abs().max()) < 0.02 * max(1.0, scale)
    # bias gradients: q and v against the column sums of the reference; the k part is zero in exact arithmetic (rows of dS sum to zero)
    # -- the new kernel writes exact zeros, the two-pass kernel and the reference their rounding noise
    bscale = float(ref_db.abs().max())
    got_db = outs[2][1].cpu().double()
    assert float((got_db[:D] - ref_db[:D]).abs().max()) < 0.03 * max(1.0, bscale)
    assert float((got_db[2 * D:] - ref_db[2 * D:]).abs().max()) < 0.03 * max(1.0, bscale)
    assert float(got_db[D:2 * D].abs().max()) == 0.0 and float(ref_db[D:2 * D].abs().max()) < 1e-6 * max(1.0, bscale)

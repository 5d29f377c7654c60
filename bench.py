#!/usr/bin/env python
"""Headline benchmark: image-text pairs/sec of the CLIP retrieval hot path
(ViT-B/16 + BERT-base dual-encoder forward + InfoNCE loss) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-also] [--no-cpu-baseline]

`--gpus N` with N > 1 launches itself: one process per GPU under `python -m torch.distributed.run` on 127.0.0.1 (RCCL);
started under an external launcher (WORLD_SIZE set) it uses that one.

A "step" = one pass of the hot path over one synthetic batch that is already resident in HBM: encode_image +
encode_text (+ RCCL all-gather of both embedding sets when N > 1) + similarity (both directions) + InfoNCE.  Rank 0
prints ONE JSON line.  The headline (`value`) is the default workload; the same line carries every other BASELINE.json
configuration this box can run under `also` (three warm-up + eight timed steps each):

  bf16_b1024_fwd_loss        (default, headline) ViT-B/16 + BERT-base, bf16 MFMA, 1024 pairs/GPU, 64 tokens; the text tower
                             runs on the unmasked tokens only (same embeddings: masked keys carry -10000, only x[:, 0] is read)
  bf16_b1024_fwd_loss_padded_text   the same with every padded position fed through the text tower, as the reference does
  bf16_b1024_train           config 3: + full backward into a flat gradient arena (+ overlapped gradient all-reduce, N > 1)
  bf16_b1024_train_opt       the same + torch.optim.AdamW(fused) step + the library's weight re-pack: a whole training step
  fp32_b256_fwd_sim          config 2: exact-f32 MFMA, 256 pairs, forward + similarity
  bf16_vitl14_b512_*         config 5: ViT-L/14 + BERT-base-architecture text tower, 512 pairs
  bf16_hf_vitl14_b512_train  config 5 as the reference trains it (huggingface_clip flavour: frozen vision tower)
  *_autograd                 the same work through the drop-in boundary the reference Trainer drives:
                             CLIPApp.forward() + compute_loss() + loss.backward() (core/trainer.py:628,646,661)

`roofline` describes the dominant kernel (the MFMA GEMM): algorithmic FLOPs of its launches / their HIP-event
durations, measured live on the launch stream in extra (untimed, single-stream) steps.  `cpu_baseline` times the
imported reference (`kind: reference`, only where /root/reference exists) or the CPU oracle (`kind: port`, a torch-CPU
restatement of the reference algorithm) on the host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

# multi-process GPU work on this platform needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise); the launcher
# normally exports it already
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

VITB16_BERTBASE = dict(
    model_type="chinese_clip", embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768,
    vision_patch_size=16, vocab_size=21128, text_attention_probs_dropout_prob=0.0, text_hidden_act="gelu",
    text_hidden_dropout_prob=0.0, text_hidden_size=768, text_initializer_range=0.02, text_intermediate_size=3072,
    text_max_position_embeddings=512, text_num_attention_heads=12, text_num_hidden_layers=12, text_type_vocab_size=2)

# BASELINE.json config 5: ViT-L/14 + hfl/chinese-roberta-wwm-ext (BERT-base architecture), 512 pairs per GPU
VITL14_ROBERTA = dict(VITB16_BERTBASE, embed_dim=768, vision_layers=24, vision_width=1024, vision_patch_size=14)

# the same towers as a huggingface_clip checkpoint (pai-clip-commercial-large style, BASELINE.json config 5 as the reference
# actually runs it: RobertaModel pooled output, CLIPVisionModel DETACHED -- only the text tower and the projections train)
HF_VITL14_ROBERTA = dict(
    text_config=dict(vocab_size=21128, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                     max_position_embeddings=512, type_vocab_size=2, pad_token_id=0, layer_norm_eps=1e-12, hidden_act="gelu",
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0),
    vision_config=dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=224,
                       patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5),
    projection_dim=768)

# ... and with the LARGE text tower of those checkpoints (CLIPTextConfig's defaults, configuration_clip.py:90-95:
# chinese-roberta-wwm-ext-large -- hidden 1024, 24 layers, 16 heads, FFN 4096; SURVEY 8d "201.09 G/pair")
HF_VITL14_ROBERTA_LARGE = dict(HF_VITL14_ROBERTA, text_config=dict(HF_VITL14_ROBERTA["text_config"], hidden_size=1024,
                                                                    intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16))

# path: "fused" = CLIPApp.contrastive_step (one C call per stage, no autograd bookkeeping);
#       "autograd" = forward() + compute_loss() + backward() through the autograd glue (the reference Trainer's calls)
WORKLOADS = {
    "bf16_b1024_fwd_loss": dict(dtype="bf16", batch=1024, seq=64, backward=False),
    "bf16_b1024_train": dict(dtype="bf16", batch=1024, seq=64, backward=True),
    "bf16_b1024_train_padded_text": dict(dtype="bf16", batch=1024, seq=64, backward=True, pack_text=False),
    "bf16_b1024_train_opt": dict(dtype="bf16", batch=1024, seq=64, backward=True, optimizer=True),
    "bf16_b1024_fwd_loss_padded_text": dict(dtype="bf16", batch=1024, seq=64, backward=False, pack_text=False),
    "bf16_b1024_fwd_loss_autograd": dict(dtype="bf16", batch=1024, seq=64, backward=False, path="autograd"),
    "bf16_b1024_train_autograd": dict(dtype="bf16", batch=1024, seq=64, backward=True, path="autograd"),
    "fp32_b256_fwd_sim": dict(dtype="fp32", batch=256, seq=64, backward=False),
    "bf16_vitl14_b512_fwd_loss": dict(dtype="bf16", batch=512, seq=64, backward=False, model="vitl14"),
    "bf16_vitl14_b512_train": dict(dtype="bf16", batch=512, seq=64, backward=True, model="vitl14"),
    "bf16_hf_vitl14_b512_train": dict(dtype="bf16", batch=512, seq=64, backward=True, model="hf_vitl14"),
    "bf16_hf_vitl14_large_b512_train": dict(dtype="bf16", batch=512, seq=64, backward=True, model="hf_vitl14_large"),
    # ModifiedResNet-50 image tower (CHINESE_CLIP with vision_layers = (3, 4, 6, 3): modeling_chineseclip.py:279-287) + BERT-base, 256 pairs,
    # through forward() / compute_loss() / backward() -- the path such a model takes (train: BatchNorm batch statistics + full backward)
    "bf16_rn50_b256_fwd": dict(dtype="bf16", batch=256, seq=64, backward=False, model="rn50", path="autograd"),
    "bf16_rn50_b256_train": dict(dtype="bf16", batch=256, seq=64, backward=True, model="rn50", path="autograd"),
}
# what the default run adds to the headline line (BASELINE.json configs 3, 2, 5 + the boundary overhead)
ALSO_N1 = ["bf16_b1024_fwd_loss_padded_text", "bf16_b1024_train", "bf16_b1024_train_padded_text", "bf16_b1024_train_opt",
           "bf16_b1024_fwd_loss_autograd",
           "bf16_b1024_train_autograd",
           "fp32_b256_fwd_sim", "bf16_vitl14_b512_fwd_loss", "bf16_vitl14_b512_train", "bf16_hf_vitl14_b512_train",
           "bf16_hf_vitl14_large_b512_train", "bf16_rn50_b256_fwd", "bf16_rn50_b256_train"]
ALSO_MULTI = ["bf16_b1024_train"]          # config 4 "(+bwd)": gradient all-reduce overlapped with the backward pass

# algorithmic GFLOP per pair (multiply-add = 2; padded tiles and softmax/LN excluded)
GFLOP_FWD_PER_PAIR = 46.152
# Forward-only workloads evaluate the last block of each tower for the CLS rows only (nothing else reaches the embeddings):
# out_proj + MLP + attention core of one ViT block for 196 of 197 tokens (2.199 G) and query / attention / output / FFN of one
# BERT layer for 63 of 64 tokens (0.756 G) are not executed.  model_tflops is computed from the EXECUTED figure.
# (round 3: the ViT's last block also projects its queries for the CLS rows only -- 2 * 196 * 768 * 768 = 0.231 G less; unless
#  EZCLIP_CLS_Q_ONLY=0)
GFLOP_FWD_EXECUTED_PER_PAIR = 46.152 - 2.199 - 0.756 - (0.0 if os.environ.get("EZCLIP_CLS_Q_ONLY") == "0" else 0.231)
GFLOP_TRAIN_EXECUTED_PER_PAIR = 138.46 - 3 * (2.199 + 0.756)     # the same rows are skipped in the backward pass
GFLOP_TRAIN_PER_PAIR = 138.46
GFLOP_FWD_PER_PAIR_VITL14 = 173.05      # ViT-L/14 (L = 257) + BERT-base text tower, 64 tokens
GFLOP_TRAIN_PER_PAIR_VITL14 = 519.2
GFLOP_TRAIN_PER_PAIR_HF_VITL14 = 162.03 + 3 * 11.025      # frozen vision tower: forward only; text tower fwd + bwd
GFLOP_TRAIN_PER_PAIR_HF_VITL14_LARGE = 162.03 + 3 * 39.06   # text: 24 layers x 64 tokens x (8 H^2 + 4 H F + 4 L H), H 1024, F 4096
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
RN50_CLIP = dict(VITB16_BERTBASE, vision_layers=[3, 4, 6, 3], vision_width=64, embed_dim=1024)


def rn_gflop_per_image(layers, width, res, out_dim):
    """algorithmic forward GFLOP of one image through ModifiedResNet (multiply-add = 2): every convolution 2 * pixels * Cout * k * k * Cin,
    the attention pool's projections and its one-query attention (modeling_chineseclip.py:27-167); BatchNorm / ReLU / pools excluded"""
    f = 0.0
    h = res // 2
    for cin, cout in ((3, width // 2), (width // 2, width // 2), (width // 2, width)):
        f += 2.0 * h * h * cout * 9 * cin
    h //= 2
    inpl = width
    for li, n in enumerate(layers):
        planes = width << li
        for bi in range(n):
            stride = 2 if (li > 0 and bi == 0) else 1
            f += 2.0 * h * h * planes * inpl                      # conv1 1x1
            f += 2.0 * h * h * planes * 9 * planes                # conv2 3x3
            ho = h // stride
            f += 2.0 * ho * ho * planes * 4 * planes              # conv3 1x1
            if stride > 1 or inpl != planes * 4:
                f += 2.0 * ho * ho * planes * 4 * inpl            # downsample 1x1 (after the anti-aliasing pool)
            inpl, h = planes * 4, ho
    e, lt = width * 32, h * h + 1
    f += 2.0 * (2 * lt + 1) * e * e + 4.0 * lt * e + 2.0 * e * out_dim
    return f / 1e9


GFLOP_FWD_PER_PAIR_RN50 = rn_gflop_per_image([3, 4, 6, 3], 64, 224, 1024) + 11.025       # + BERT-base text tower at 64 tokens

# The 101 launches of the dominant kernel (ProfScope PROF_GEMM) in one forward step at 1024 pairs -- the set the roofline's time
# average runs over (round 4 listed 97 of them).  Blocks 0..10 of each tower run the four big products of tools/gemm_bench; the LAST
# block of each tower is evaluated for the CLS rows only (k | v for every token, everything else for 1024 rows); then the two
# projections into the joint space.  Packed text: M = the unmasked tokens (~36 k of 65 536); the summaries were taken at 65 536.
FWD_GEMM_MIX = {"vit.qkv": 11, "vit.out+res": 11, "vit.fc+qgelu": 11, "vit.proj+res": 11, "bert.qkv": 11, "bert.qkvo+res": 11,
                "bert.ffn1+gelu": 11, "bert.ffn2+res": 11, "patch": 1}
# (scale of a measured shape, count): the last blocks' k | v products = two thirds of the q | k | v product of the same tower
FWD_GEMM_MIX_SCALED = {"vit.kv(last)": ("vit.qkv", 2.0 / 3.0, 1), "bert.kv(last)": ("bert.qkv", 2.0 / 3.0, 1)}
# (M, N, K, count) of the small launches: CLS-row products of the two last blocks (q, out, fc / ffn1, proj / ffn2) and the two
# projections; their traffic is taken as their algorithmic bytes (1-5 MB each against ~0.9 GB for a big product)
FWD_GEMM_MIX_SMALL = [(1024, 768, 768, 4), (1024, 3072, 768, 2), (1024, 768, 3072, 2), (1024, 512, 768, 2)]
FWD_GEMM_LAUNCHES = sum(FWD_GEMM_MIX.values()) + sum(v[2] for v in FWD_GEMM_MIX_SCALED.values()) + sum(v[3] for v in FWD_GEMM_MIX_SMALL)
assert FWD_GEMM_LAUNCHES == 101


def pmc_traffic(workload):
    """Average HBM-side bytes per launch of the dominant kernel (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc passes of
    tools/pmc_gemm.sh summarised by tools/pmc_summary.py into profiles/*_gemm_traffic.json; MALL hits are counted as
    fetches on gfx950), averaged over the SAME 101 launches as the roofline's time average (FWD_GEMM_MIX*).  Measured offline with
    the same kernels, shapes and batch, one cool launch per shape: bench.py cannot run a profiler around itself.  None when no
    summary is committed or the workload is not the 1024-pair forward."""
    if workload != "bf16_b1024_fwd_loss":
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_gemm_traffic.json")))
    if not files:
        return None
    t = json.load(open(files[-1]))
    if "bert.qkv" not in t and "bert.qkvo+res" in t:       # summaries taken before the q | k | v products were merged
        t = dict(t, **{"bert.qkv": {k: 3 * v for k, v in t["bert.qkvo+res"].items()}})
    tot, alg = 0.0, 0.0
    for name, cnt in FWD_GEMM_MIX.items():
        if name not in t:
            return None
        tot += cnt * (t[name]["fetch_bytes"] + t[name]["write_bytes"])
        alg += cnt * t[name]["algorithmic_bytes"]
    for name, (src, scale, cnt) in FWD_GEMM_MIX_SCALED.items():
        tot += cnt * scale * (t[src]["fetch_bytes"] + t[src]["write_bytes"])
        alg += cnt * scale * t[src]["algorithmic_bytes"]
    for M, N, K, cnt in FWD_GEMM_MIX_SMALL:
        b = 2.0 * (M * K + N * K + M * N)
        tot += cnt * b
        alg += cnt * b
    n = FWD_GEMM_LAUNCHES
    return {"bytes_per_launch": round(tot / n), "algorithmic_bytes_per_launch": round(alg / n), "launches": n,
            "source": os.path.basename(files[-1])}


def synth_batch(batch, seq, vocab, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    px = torch.randn((batch, 3, 224, 224), generator=g, device=device, dtype=torch.float32)
    ids = torch.randint(1, vocab, (batch, seq), generator=g, device=device, dtype=torch.int64)
    lens = torch.randint(8, seq + 1, (batch,), generator=g, device=device)
    ids = ids * (torch.arange(seq, device=device)[None, :] < lens[:, None])
    return px, ids


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# port / reference throughput of cpu_baseline's two legs, re-measured EVERY round with tools/cpu_baseline_compare.py in the build
# container (the only place both exist; the GPU box has no /root/reference).  Round 5, 2026-09-23, 8 vCPU Xeon 2.1 GHz, 25 s per leg:
# reference 15.0 fwd / 3.1-4.3 train pairs/s, port 13.8-13.9 / 4.0-4.2 -> forward 0.92-0.93; the training ratio moved between 0.94 and
# 1.33 from run to run (4-5 iterations of ~2 s each on a shared host) -- quoted as ~1.  Round 3 on the same host: 0.99 / 0.84.
PORT_VS_REFERENCE = {"fwd": 0.93, "train": 0.94, "train_range": [0.94, 1.33],
                     "measured_on": "build container, 8 vCPU Xeon 2.1 GHz, 2026-09-23 (round 5), tools/cpu_baseline_compare.py 25"}


def cpu_baseline(seconds_target=10.0):
    """The reference path on the host cores, fp32, 8-pair batches: forward + similarity + InfoNCE (`value`)
    and forward + loss + backward (`train_value`).  Where the reference checkout exists (the build container) it IS the
    imported reference `CLIPApp` (`kind: reference`); on the GPU box (no /root/reference) it is the CPU oracle, a
    torch-CPU restatement of the same algorithm (`kind: port`).  The port is NOT the reference's speed: `port_vs_reference`
    (PORT_VS_REFERENCE below) is the ratio of the two measured back to back on the build container with
    tools/cpu_baseline_compare.py -- the forward matches, the port's backward (explicit softmax / LayerNorm formulas: more
    autograd nodes) is slower.  A baseline for orientation only; never quote a training speed-up from a `port` number."""
    from oracle import clip_oracle as O
    from oracle import ref_harness as R
    # a small-batch CPU forward stops scaling (and collapses from oversubscription) beyond a few
    # dozen threads: use at most 32 and report the number actually used
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    cfg = O.CONFIGS["vitb16_bertbase"]
    sd = O.make_state_dict(cfg, 1234)
    px, ids = O.make_inputs(cfg, 8, 64, 0)
    kind = "port"
    if R.reference_available():
        import contextlib
        import tempfile
        try:
            with tempfile.TemporaryDirectory() as d, contextlib.redirect_stdout(sys.stderr):   # (the reference prints its config)
                R.write_checkpoint_dir(d, cfg, sd)
                ref = R.reference_clip_app(d)
            ref.eval()
            kind = "reference"

            def fwd():
                out = ref({"pixel_values": px.clone(), "input_ids": ids.clone()})
                return ref.compute_loss(out, [])["loss"]

            def train():
                ref.zero_grad(set_to_none=True)
                fwd().backward()
        except Exception as e:      # a broken checkout must not take the bench line with it
            print("cpu_baseline: reference import failed (%s); timing the port" % e, file=sys.stderr)
            kind = "port"
    if kind == "port":
        def fwd():
            out = O.clip_forward(sd, cfg, px, ids)
            return O.clip_loss(out["logits_per_text"])

        def train():
            O.forward_loss_backward(sd, cfg, px, ids)

    def timed(fn, budget, cap):
        fn()                                    # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            fn()
            n += 1
            el = time.perf_counter() - t0
            if el >= budget or n >= cap:
                return n, el

    with torch.no_grad():
        n, el = timed(fwd, seconds_target, 40)
    nt, elt = timed(train, seconds_target * 0.8, 12)
    return {"value": round(8 * n / el, 3), "unit": "pairs/s", "cores": cores, "kind": kind,
            "train_value": round(8 * nt / elt, 3), **({"port_vs_reference": PORT_VS_REFERENCE} if kind == "port" else {}),
            "sample": "%d x (8 pairs, 224x224 + 64 tokens) fp32 fwd+similarity+InfoNCE and %d x fwd+loss+backward, torch CPU %s, %s"
                      % (n, nt, torch.__version__, _cpu_model())}


TINY_RECALL = dict(model_type="chinese_clip", embed_dim=64, image_resolution=64, vision_layers=2, vision_width=128, vision_patch_size=16,
                   vocab_size=211, text_attention_probs_dropout_prob=0.0, text_hidden_act="gelu", text_hidden_dropout_prob=0.0, text_hidden_size=128,
                   text_initializer_range=0.02, text_intermediate_size=512, text_max_position_embeddings=64, text_num_attention_heads=2,
                   text_num_hidden_layers=2, text_type_vocab_size=2)


def recall_leg(device, pairs=96, steps=60):
    """BASELINE.json's metric ends "; R@1 vs ref" (appzoo/clip/evaluator.py:47-67).  A small dual encoder is trained by the bf16
    HIP path on `pairs` FIXED synthetic pairs until they are separable (the "overfit" fixture of tests/test_model_gpu.py, larger), then its
    text -> image recall is computed by the library's fused similarity + rank kernel.  So that R@1 is not a trivial 1.0, the last eight
    captions are exact DUPLICATES of the first eight: of two queries with one caption and two images exactly one can rank its own image
    first (the reference's stable descending sort breaks an exact tie by index), so a correct pipeline reads R@1 = 88 / 96 = 0.916667 whichever
    of the two images scores higher -- a figure that does not hinge on near-ties (the fp32 oracle's smallest margin between a paired score
    and the best other one is 0.5 in cosine after 40 steps).  Returns (fields, trained weights, inputs): the cpu_baseline leg evaluates the
    SAME weights with the CPU oracle (`recall_at_1_oracle`); the two must agree to 1e-3."""
    from easynlp_amd.appzoo.clip import CLIPApp
    from easynlp_amd.appzoo.clip.evaluator import recall_at_k
    cfg = TINY_RECALL
    app = CLIPApp.from_config(cfg, seed=8, device=device, compute_dtype="bf16")
    g = torch.Generator(device="cpu").manual_seed(5)
    px = torch.randn(pairs, 3, 64, 64, generator=g)
    ids = torch.randint(1, cfg["vocab_size"], (pairs, 12), generator=g)
    ids[:, 8:] *= (torch.rand(pairs, 4, generator=g) < 0.6)
    ids[pairs - 8:] = ids[:8]
    pxd, idd = px.to(device), ids.to(device)
    opt = torch.optim.AdamW(app.parameters(), lr=2e-3, weight_decay=0.0)
    app.train()
    losses = []
    for _ in range(steps):
        opt.zero_grad()
        loss = app.compute_loss(app({"pixel_values": pxd, "input_ids": idd}), [])["loss"]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(app.parameters(), 1.0)
        opt.step()
        losses.append(float(loss.item()))
    app.eval()
    with torch.no_grad():
        out = app({"pixel_values": pxd, "input_ids": idd}, feat=True)
    (mean_recall, r1, r5, r10), _ = recall_at_k(out["text_embeds"], out["image_embeds"])
    weights = {n: p.detach().float().cpu() for n, p in app._params.items()}
    fields = {"recall_at_1": round(r1, 6), "recall_detail": {"r5": round(r5, 6), "r10": round(r10, 6), "mean_recall": round(mean_recall, 6), "pairs": pairs,
              "expected": round((pairs - 8) / pairs, 6),
              "what": "text->image R@k of a 2+2-layer dual encoder after %d AdamW steps of the bf16 HIP path on %d fixed synthetic pairs, 8 captions "
                      "duplicated (loss %.3f -> %.3f); ranks by ezclip_recall_ranks_fused" % (steps, pairs, losses[0], losses[-1])}}
    del app, opt
    return fields, weights, (px, ids)


def recall_oracle(weights, inputs):
    """the reference evaluator's arithmetic (oracle.clip_oracle.recall_at_k: full sort per query, evaluator.py:47-67) on the fp32 CPU oracle's
    embeddings of the SAME trained weights -- part of the cpu_baseline leg (the only place bench.py may use oracle/)"""
    from oracle import clip_oracle as O
    px, ids = inputs
    with torch.no_grad():
        ref = O.clip_forward(weights, TINY_RECALL, px, ids)
    r = O.recall_at_k(ref["text_embeds"], ref["image_embeds"])
    return float(r[1]) if len(r) > 3 else float(r[0])


def relaunch(args):
    """`python bench.py --gpus N` without an external launcher: one process per GPU under torch.distributed.run."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus and not os.environ.get("EZCLIP_BENCH_ONE_GPU"):
        sys.exit("bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, have))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + \
          [a for a in sys.argv[1:] if a != "--launcher"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", EZCLIP_BENCH_LAUNCHED="1")
    sys.exit(subprocess.run(cmd, env=env, cwd=ROOT).returncode)


class Ctx:
    pass


def preflight(world, rank, device, backend, use_dist):
    """`bench.py --gpus N --preflight`: the collectives of the N > 1 path (parallel.py: one all-gather of [n, 2E], one
    reduce-scatter, 64 MiB gradient buckets), each on its own, each timed, each inside its own try -- a first 8-GPU run that fails
    reports the stage it failed in (`failed_at`) instead of a stack from inside a training step.  No model, no kernels of this library."""
    import torch.distributed as dist
    res = {"preflight": True, "rccl_ranks": world if use_dist else 0, "collective_backend": backend if use_dist else None,
           "device": torch.cuda.get_device_name(device), "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
           "stages": {}, "failed_at": None}

    def timed(name, fn, reps=5):
        if res["failed_at"]:
            return
        try:
            fn()                                                  # first call: communicator / channel set-up
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            res["stages"][name] = {"ms": round((time.perf_counter() - t0) / reps * 1e3, 4)}
        except Exception as e:      # noqa: BLE001 -- the point of the preflight is to name the failing stage
            res["failed_at"] = name
            res["error"] = "%s: %s" % (type(e).__name__, str(e)[:500])
            if use_dist and world > 1:
                # The ranks that did not fail are inside (or about to enter) the next collective and would wait there for this one until
                # the RCCL timeout.  The failing rank prints THE line itself and leaves with a non-zero status: the launcher
                # (torch.distributed.run) then tears the other workers down within seconds.
                res["all_ranks_ok"], res["reported_by_rank"] = False, rank
                print(json.dumps(res), flush=True)
                os._exit(3)

    n, e2 = 1024, 1024                                            # [n, 2E] bf16-free: float32 embeddings, 4 MiB per rank
    mine = torch.randn(n, e2, device=device)
    if use_dist and world > 1:
        gathered = torch.empty(world * n, e2, device=device)
        timed("all_gather_4MiB", lambda: dist.all_gather_into_tensor(gathered, mine))
        if not res["failed_at"]:
            ok = bool(torch.equal(gathered[rank * n:(rank + 1) * n], mine))
            res["stages"]["all_gather_4MiB"]["own_rows_intact"] = ok
        back = torch.empty(n, e2, device=device)
        if backend == "gloo":
            timed("reduce_scatter_4MiB(all_reduce on gloo)", lambda: dist.all_reduce(gathered))
        else:
            timed("reduce_scatter_4MiB", lambda: dist.reduce_scatter_tensor(back, gathered, op=dist.ReduceOp.SUM))
        bucket = torch.ones(16 << 20, device=device)              # one 64 MiB float32 gradient bucket
        timed("all_reduce_64MiB", lambda: dist.all_reduce(bucket, op=dist.ReduceOp.SUM))
        if not res["failed_at"]:
            st = res["stages"]["all_reduce_64MiB"]
            st["bus_gbps"] = round(2 * (world - 1) / world * 64 * 2 ** 20 / (st["ms"] * 1e-3) / 1e9, 1)
            torch.cuda.synchronize()
            res["stages"]["all_reduce_64MiB"]["finite"] = bool(torch.isfinite(bucket[:16]).all())
        timed("barrier", lambda: dist.barrier(), reps=3)
    else:
        res["note"] = "world size 1: nothing to exchange (run with --gpus N, N > 1)"
    ok = torch.tensor([0 if res["failed_at"] else 1], device=device)
    if use_dist and world > 1 and not res["failed_at"]:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    res["all_ranks_ok"] = bool(int(ok.item()))
    if rank == 0:
        print(json.dumps(res), flush=True)
    if use_dist:
        try:
            dist.destroy_process_group()
        except Exception:      # noqa: BLE001
            pass


class Telemetry:
    """Shader clock and socket power of GPU 0 while a leg runs, sampled by a host thread twice a second: amdgpu's hwmon files
    (power1_average / power1_input in microwatts, freq1_input in Hz) where present, `rocm-smi --showpower --showclocks` otherwise.
    Evidence for DESIGN.md 6.0 ("the sustained rate is the 1 400 W socket cap"): it never influences the timing."""

    def __init__(self, period=0.5):
        import glob
        import threading
        self.period, self.samples, self._stop = period, [], threading.Event()
        self.source = None
        self._hw = None
        # the hwmon directory of cuda:0 -- matched by PCI address (the box exposes more amdgpu cards than this process may use)
        want = None
        try:
            pr = torch.cuda.get_device_properties(0)
            want = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pass
        cands = []
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            dev = os.path.dirname(os.path.dirname(d))
            try:
                if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
                    continue
            except OSError:
                continue
            pw = [f for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(d, f))]
            if pw and os.path.exists(os.path.join(d, "freq1_input")):
                cands.append((os.path.basename(os.path.realpath(dev)), os.path.join(d, pw[0]), os.path.join(d, "freq1_input")))
        for addr, pwf, fqf in cands:
            if want and addr.lower().startswith(want):
                self._hw = (pwf, fqf)
                self.source = "hwmon:%s@%s" % (os.path.basename(pwf), addr)
        if self._hw is None and len(cands) == 1:
            self._hw = cands[0][1:]
            self.source = "hwmon:%s@%s" % (os.path.basename(cands[0][1]), cands[0][0])
        if self._hw is None:
            import shutil
            if shutil.which("rocm-smi"):
                self.source = "rocm-smi"
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        if self._hw is not None:
            try:
                return (time.perf_counter(), float(open(self._hw[1]).read()) / 1e6, float(open(self._hw[0]).read()) / 1e6)
            except (OSError, ValueError):
                return None
        if self.source == "rocm-smi":
            import re
            import subprocess
            try:
                txt = subprocess.run(["rocm-smi", "-d", "0", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            except Exception:
                return None
            mhz = re.search(r"sclk clock level[^\n]*\((\d+)Mhz\)", txt)
            w = re.search(r"Package Power \(W\):\s*([0-9.]+)", txt)
            if mhz and w:
                return (time.perf_counter(), float(mhz.group(1)), float(w.group(1)))
        return None

    def _run(self):
        while not self._stop.is_set():
            r = self._read()
            if r is not None:
                self.samples.append(r)
            self._stop.wait(self.period)

    def __enter__(self):
        self.t0 = time.perf_counter()
        if self.source:
            self._thread.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self.source:
            self._thread.join(timeout=6)

    def summary(self, skip_seconds=3.0):
        """means over the samples taken after the first `skip_seconds` (the chip needs ~3 s to settle at its cap)"""
        xs = [x for x in self.samples if x[0] - self.t0 >= skip_seconds] or self.samples
        if not xs:
            return {"source": self.source, "samples": 0}
        return {"source": self.source, "samples": len(xs), "shader_clock_mhz_mean": round(sum(x[1] for x in xs) / len(xs), 1),
                "shader_clock_mhz_min": round(min(x[1] for x in xs), 1), "socket_power_w_mean": round(sum(x[2] for x in xs) / len(xs), 1),
                "socket_power_w_max": round(max(x[2] for x in xs), 1)}


# gradient all-reduce buckets of the N > 1 training workloads: float32 (what DDP's own all-reduce moves: the default) or, with
# EZCLIP_GRAD_BUCKET_DTYPE=bf16, bf16-compressed buckets (DDP's bf16_compress_hook: half the bytes per xGMI link)
GRAD_BUCKET_DTYPE = torch.bfloat16 if os.environ.get("EZCLIP_GRAD_BUCKET_DTYPE", "").lower() in ("bf16", "bfloat16") else None
NBATCH = 4      # distinct synthetic batches a workload rotates through (run_workload)


def build_app(wl, device, text_dropout=0.0):
    from easynlp_amd.appzoo.clip import CLIPApp
    model_cfg = VITL14_ROBERTA if wl.get("model") == "vitl14" else VITB16_BERTBASE
    name = ("ViT-L/14 + chinese-roberta-wwm-ext (BERT-base arch)" if wl.get("model") == "vitl14"
            else "ViT-B/16 + BERT-base") + " (chinese_clip), random init"
    if text_dropout > 0:
        model_cfg = dict(model_cfg, text_hidden_dropout_prob=text_dropout, text_attention_probs_dropout_prob=text_dropout)
    if wl.get("model") in ("hf_vitl14", "hf_vitl14_large"):
        hf_cfg = HF_VITL14_ROBERTA_LARGE if wl["model"] == "hf_vitl14_large" else HF_VITL14_ROBERTA
        if text_dropout > 0:        # RoBERTa's train-mode dropouts live in text_config (CLIPTextConfig)
            hf_cfg = dict(hf_cfg, text_config=dict(hf_cfg["text_config"], hidden_dropout_prob=text_dropout,
                                                   attention_probs_dropout_prob=text_dropout))
        app = CLIPApp.from_hf_config(hf_cfg, seed=1234, device=device, compute_dtype=wl["dtype"])
        name = "huggingface_clip: frozen ViT-L/14 + chinese-roberta-wwm-ext%s + pooler, random init" % (
            "-large (hidden 1024, 24 layers, 16 heads)" if wl["model"] == "hf_vitl14_large" else " (BERT-base arch)")
    elif wl.get("model") == "rn50":
        app = CLIPApp.from_config(RN50_CLIP, seed=1234, device=device, compute_dtype=wl["dtype"])
        name = "ModifiedResNet-50 (3, 4, 6, 3) width 64 + BERT-base (chinese_clip), random init"
    else:
        app = CLIPApp.from_config(model_cfg, seed=1234, device=device, compute_dtype=wl["dtype"])
    app.eval()
    if text_dropout > 0 or (wl["backward"] and wl.get("path") == "autograd"):
        app.train()             # what Trainer does (core/trainer.py:281); dropout probabilities are 0 unless --text-dropout
    return app, name


def gflop_per_pair(wl):
    if wl.get("model") == "hf_vitl14":
        return GFLOP_TRAIN_PER_PAIR_HF_VITL14, None
    if wl.get("model") == "hf_vitl14_large":
        return GFLOP_TRAIN_PER_PAIR_HF_VITL14_LARGE, None
    if wl.get("model") == "vitl14":
        return (GFLOP_TRAIN_PER_PAIR_VITL14 if wl["backward"] else GFLOP_FWD_PER_PAIR_VITL14), None
    if wl.get("model") == "rn50":       # (the stem's first convolution has no input gradient: 3x is an upper bound by 0.2 %)
        return (3.0 * GFLOP_FWD_PER_PAIR_RN50 if wl["backward"] else GFLOP_FWD_PER_PAIR_RN50), None
    cls_last = os.environ.get("EZCLIP_CLS_LAST", "1") != "0"
    cls_train = os.environ.get("EZCLIP_CLS_TRAIN", "1") != "0"
    if wl["backward"]:
        return (GFLOP_TRAIN_EXECUTED_PER_PAIR if cls_train else GFLOP_TRAIN_PER_PAIR), GFLOP_TRAIN_PER_PAIR
    return (GFLOP_FWD_EXECUTED_PER_PAIR if cls_last else GFLOP_FWD_PER_PAIR), GFLOP_FWD_PER_PAIR


def run_workload(name, c, steps, warmup, batch_override=0, text_dropout=0.0, profile=True, sustained_steps=0, sustained_seconds=20.0):
    """Build the model of workload `name`, run `warmup` untimed and exactly `steps` timed steps (barrier + synchronize on
    both sides, MAX over ranks), then the roofline leg.  Returns the fields of the JSON line for this workload."""
    import torch.distributed as dist
    from easynlp_amd import lib as L
    from easynlp_amd import parallel as P
    wl = dict(WORKLOADS[name])
    if batch_override:
        wl["batch"] = batch_override
    B, S = wl["batch"], wl["seq"]
    world, rank, device = c.world, c.rank, c.device
    # The models of the workloads that ran before this one in the same process are garbage by now, but cyclic garbage: collect it here,
    # outside any timed region, and give its device memory back, rather than leave it to a collection inside this workload's timed steps.
    # (Hygiene; it does NOT explain why the autograd training step reads 3-4 % slower as the 7th workload of the default line than as the
    # 2nd.  Round 5 ruled out: any single predecessor (profiles/r5_autograd_bisect.log), the allocator cache (EZCLIP_BENCH_KEEP_CACHE),
    # the clock (clock_mhz_timed_steps is HIGHER there: the GPU is ~4 % less busy, time_share sums to 0.93 instead of 0.97), Python's
    # cyclic collector (gc.freeze() changes nothing: profiles/r5_autograd_order.log).  A kernel trace of the whole line
    # (profiles/r5_default_line_trace_segments.log) puts the idle time at the step boundaries: 1.3-3.9 ms between the last backward kernel
    # and the next step's first launch -- the host side of the autograd path is late there; why more so late in the process: open.)
    import gc
    gc.collect()
    if not os.environ.get("EZCLIP_BENCH_KEEP_CACHE"):       # (A/B switch: does returning the previous workload's 60 GB to the driver matter?)
        torch.cuda.empty_cache()
    app, model_name = build_app(wl, device, text_dropout)
    if wl.get("pack_text") is False:
        app._engine.pack_text = False
    # NBATCH distinct synthetic batches (different images, tokens and sentence lengths), visited round-robin: every step sees a
    # batch other than the previous one, as a training loop does -- the text tower's packing metadata (one device launch whose
    # scalars the host reads from pinned memory) is rebuilt inside every timed step.  Batch 0 is the batch of
    # rounds 1-2 (seed 1000 + rank): the reported loss is the loss on it.
    batches = [synth_batch(B, S, VITB16_BERTBASE["vocab_size"], device, seed=1000 + rank + 97 * k) for k in range(NBATCH)]
    px, ids = batches[0]
    pg = True if world > 1 else False
    autograd = wl.get("path") == "autograd"
    if autograd and world > 1:
        app.contrastive_scope = "global"
    opt = None
    if wl.get("optimizer"):
        # a real in-place weight update per step: the next forward re-packs the library's bf16 / transposed copies
        opt = torch.optim.AdamW([p for p in app.parameters() if p.requires_grad], lr=1e-6, eps=1e-6, weight_decay=0.01, fused=True)
    hf_all = [{} for _ in batches]
    if wl.get("model") in ("hf_vitl14", "hf_vitl14_large"):
        hf_all = [{"token_type_ids": torch.zeros_like(i_), "attention_mask": i_.ne(0).long()} for _, i_ in batches]
    counter = [0]

    def step(which=None):
        k = counter[0] % NBATCH if which is None else which
        counter[0] += 1
        px, ids = batches[k]
        hf_inputs = hf_all[k]
        if autograd:
            if wl["backward"]:
                for p in app.parameters():          # optimizer.zero_grad() of the Trainer loop (set_to_none, core/trainer.py:337)
                    p.grad = None
                out = app(dict({"pixel_values": px, "input_ids": ids}, **hf_inputs))
                loss = app.compute_loss(out, [])["loss"]
                loss.backward()
                if world > 1:                       # what DDP's hooks do for the reference (not overlapped on this path)
                    P.average_gradients(list(app.parameters()))
                return loss.detach()
            with torch.no_grad():
                out = app(dict({"pixel_values": px, "input_ids": ids}, **hf_inputs))
                return app.compute_loss(out, [])["loss"]
        if wl["backward"]:
            # the gradient all-reduce belongs to a training step (DDP in trainer.py:101-108): buckets of the flat gradient
            # arena go out while the backward pass is still running; the embedding gradients are those of the global mean
            # loss, so ranks hold partial sums and the reduction is a SUM
            loss = app.contrastive_step(px, ids, process_group=pg, backward=True, zero_grad=True, reduce_gradients=world > 1,
                                        bucket_dtype=GRAD_BUCKET_DTYPE)
            if opt is not None:
                opt.step()
                app._engine.mark_weights_dirty()
            return loss
        with torch.no_grad():
            return app.contrastive_step(px, ids, process_group=pg, backward=False)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        loss = step()
    fence()
    rows_seen = []
    host_s = 0.0
    # shader clock / socket power while the timed steps run (a host thread reading hwmon files four times a second: it never touches
    # the timing).  The chip sits at its power cap and warms up over the ~2 minutes of the default line: the SAME step reads 2-4 %
    # slower as the 7th workload than as the 2nd (profiles/r5_autograd_order.log) -- the clock next to every number says why.
    tel_w = Telemetry(period=0.25) if rank == 0 else None
    if tel_w is not None:
        tel_w.__enter__()
    # HIP events on the launch stream around the same K steps: a step ends with the main stream having joined the text
    # tower's side stream, so the closing event is behind all of the step's work.  Reported beside the host clock, never instead of it.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        h0 = time.perf_counter()
        loss = step()
        host_s += time.perf_counter() - h0                     # host time inside step(): the packed-text metadata poll makes the host
        rows_seen.append(app._engine.last_text_rows)          # wait for the device, so this tracks ms_per_step for packed workloads
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    event_ms = ev0.elapsed_time(ev1) / steps
    tel_timed = None
    if tel_w is not None:
        tel_w.__exit__(None, None, None)
        tel_timed = tel_w.summary(skip_seconds=0.0)
    per_rank_ms = elapsed / steps * 1e3
    if world > 1:
        t = torch.tensor([elapsed, -elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, fastest = float(t[0].item()), -float(t[1].item())
    else:
        fastest = elapsed
    step_ms = elapsed / steps * 1e3
    loss_val = float(step(0).item())       # untimed: the loss on batch 0 (the batch the parity tests evaluate the oracle on)
    fence()

    # ---- roofline leg: per-launch HIP events around the dominant kernel (untimed extra steps, towers back to back on
    # one stream so that no two timed kernels share the CUs) ----
    lib = L.load()
    roof, extra = None, {}
    if profile:
        # as many steps as were timed (at most 10), entered warm: two un-recorded steps first and no idle gap before the recorded ones,
        # so that the leg's kernels run at the clock of the timed region (the chip sits at its 1 400 W cap: DESIGN 6.0).  The events'
        # sum per step agrees with the rocprofv3 kernel trace of the same run (profiles/r3ac_*_kernel_stats.md).
        nprof = max(2, min(int(steps), 10))
        two = app.two_streams
        app.two_streams = False
        step()
        step()
        if rank == 0:
            L.check(lib.ezclip_profile_begin())
        for _ in range(nprof):      # every rank steps (the collectives need all of them); only rank 0 records events
            step()
        fence()
        app.two_streams = two
        if rank == 0:
            res = {}
            for cls, kname in ((0, "gemm"), (1, "attention"), (2, "layernorm")):
                ms, work, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
                L.check(lib.ezclip_profile_end(cls, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(n)))
                res[kname] = (ms.value, work.value, n.value)
            ms, flops, n = res["gemm"]
            peak = PEAK_TFLOPS[wl["dtype"]]
            ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            traffic = pmc_traffic(name)
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": (traffic or {}).get("bytes_per_launch"),
                    "traffic_detail": traffic,
                    "kernel": ("ezclip::gemm_nt_8p_kernel / gemm_tn_8p_kernel (bf16, 256x256x64 8-phase, v_mfma_f32_16x16x32_bf16)"
                               if wl["dtype"] == "bf16" else "ezclip::gemm_nt_kernel<fp32> (128x128, exact-f32 MFMA)"),
                    "launches_per_step": n // nprof, "avg_launch_us": round(ms * 1e3 / max(n, 1), 2),
                    "algorithmic_gflop_per_launch": round(flops / max(n, 1) / 1e9, 3)}
            # shares of the single-stream step (the timed steps overlap the towers on two streams)
            extra["time_share"] = {k: round(v[0] / nprof / step_ms, 4) for k, v in res.items()}
            a_ms, a_fl, a_n = res["attention"]
            extra["attention_tflops"] = round(a_fl / (a_ms * 1e-3) / 1e12, 2) if a_ms > 0 else None
            l_ms, l_by, l_n = res["layernorm"]
            extra["layernorm_gbps"] = round(l_by / (l_ms * 1e-3) / 1e9, 1) if l_ms > 0 else None
    # ---- sustained leg (round 4; VERDICT r3 "put the sustained regime in the line"): the same step for `sustained_steps` steps or
    # `sustained_seconds`, whichever ends first -- the chip settles at its socket power cap within a few seconds and the default
    # 20-step figure above still has thermal headroom.  Chunks of 25 steps with a synchronisation between them
    # (one idle gap of microseconds per second of work); clock and power sampled by a host thread.
    sustained = None
    if sustained_steps > 0 and world == 1:
        chunk, done, marks = 25, 0, []
        fence()
        evs = [torch.cuda.Event(enable_timing=True)]
        with Telemetry() as tel:
            t0s = time.perf_counter()
            evs[0].record()
            while done < sustained_steps and time.perf_counter() - t0s < sustained_seconds:
                for _ in range(chunk):
                    step()
                evs.append(torch.cuda.Event(enable_timing=True))
                evs[-1].record()
                torch.cuda.synchronize()
                done += chunk
                marks.append((done, time.perf_counter() - t0s))
        total_s = marks[-1][1]
        ev_total_ms = evs[0].elapsed_time(evs[-1])
        half = [m for m in marks if m[0] * 2 >= done]                # the second half: past the settling seconds
        first_of_half = marks[len(marks) - len(half) - 1] if len(half) < len(marks) else (0, 0.0)
        ms_steady = (half[-1][1] - first_of_half[1]) / max(1, half[-1][0] - first_of_half[0]) * 1e3
        sustained = {"steps": done, "seconds": round(total_s, 2), "ms_per_step": round(total_s / done * 1e3, 3),
                     "ms_per_step_hip_events": round(ev_total_ms / done, 3),
                     "ms_per_step_second_half": round(ms_steady, 3), "value_second_half": round(B / ms_steady * 1e3, 1),
                     "telemetry": tel.summary()}
        clk = sustained["telemetry"].get("shader_clock_mhz_mean")
        if clk and wl["dtype"] == "bf16":
            # the dense bf16 MFMA peak at the clock the chip was allowed: 2.5 PFLOP/s x clock / 2.4 GHz
            sustained["peak_tflops_at_mean_clock"] = round(PEAK_TFLOPS["bf16"] * clk / 2400.0, 1)
    buckets = getattr(app, "last_grad_buckets", None)
    text_rows = None
    if rows_seen and all(r is not None for r in rows_seen):       # mean over the timed steps (the batches differ in length)
        text_rows = (int(round(sum(r[0] for r in rows_seen) / len(rows_seen))), rows_seen[0][1])
    two_streams = bool(app.two_streams)
    del app, opt
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    gflop, gflop_all = gflop_per_pair(wl)
    if text_rows and text_rows[0] != text_rows[1] and wl.get("model") in (None, "rn50"):
        # packed text tower: its GEMM / LayerNorm work scales with the rows that went through it (11.025 G per pair forward at 64
        # tokens, of which 0.756 G are the skipped CLS-only part of the last layer; three times that with the backward pass)
        gflop -= (3 if wl["backward"] else 1) * (11.025 - 0.756) * (1.0 - text_rows[0] / text_rows[1])
    value = world * B * steps / elapsed
    out = {
        "workload": name, "value": round(value, 2), "ms_per_step": round(step_ms, 3), "dtype": wl["dtype"],
        "towers": model_name, "pairs_per_gpu": B, "seq_len": S, "path": wl.get("path", "fused"),
        "stages": "encode_image+encode_text" + ("+allgather" if world > 1 else "") + "+similarity(2 dirs)+InfoNCE"
                  + ("+backward" if wl["backward"] else "") + ("+grad_allreduce(overlapped)" if wl["backward"] and world > 1 and not autograd else "")
                  + ("+AdamW+repack" if wl.get("optimizer") else ""),
        "two_streams": two_streams, "loss": round(loss_val, 5), "host_ms_per_step": round(host_s / steps * 1e3, 3),
        "ms_per_step_hip_events": round(event_ms, 3),
        "clock_mhz_timed_steps": (tel_timed or {}).get("shader_clock_mhz_mean"), "power_w_timed_steps": (tel_timed or {}).get("socket_power_w_mean"),
        "batches_rotated": NBATCH, "pack_meta_in_timed_region": True,
        "ms_per_step_ranks": {"max": round(step_ms, 3), "min": round(fastest / steps * 1e3, 3), "this_rank": round(per_rank_ms, 3)},
        "text_tower_rows": {"through_the_tower": text_rows[0], "tokens_in_the_batch": text_rows[1]} if text_rows else None,
        "gflop_per_pair": {"algorithmic_all_tokens": gflop_all, "executed": round(gflop, 3)},
        "model_tflops_per_gpu": round(value / world * gflop / 1e3, 2),
        "model_mfma_frac": round(value / world * gflop / 1e3 / PEAK_TFLOPS[wl["dtype"]], 4),
        "roofline": roof,
    }
    if buckets:
        out["grad_allreduce_buckets_mib"] = [round((e - s) * (2 if GRAD_BUCKET_DTYPE is not None else 4) / 2 ** 20, 1) for s, e in buckets]
        out["grad_allreduce_bucket_dtype"] = "bf16" if GRAD_BUCKET_DTYPE is not None else "f32"
    if sustained:
        tf = sustained["value_second_half"] * gflop / 1e3                        # executed model TFLOP/s in the steady state
        sustained["model_mfma_frac_second_half"] = round(tf / PEAK_TFLOPS[wl["dtype"]], 4)
        if sustained.get("peak_tflops_at_mean_clock"):
            sustained["model_frac_of_clock_adjusted_peak"] = round(tf / sustained["peak_tflops_at_mean_clock"], 4)
        out["sustained"] = sustained
    out.update(extra)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="bf16_b1024_fwd_loss", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override pairs per GPU (debugging only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-recall", action="store_true", help="skip the recall_at_1 leg (a few seconds: a tiny dual encoder trained on fixed pairs)")
    ap.add_argument("--no-also", action="store_true", help="only the headline workload (no `also` object)")
    ap.add_argument("--also", default="", help="comma-separated workloads for the `also` object (default: every BASELINE config)")
    ap.add_argument("--also-steps", type=int, default=8)
    ap.add_argument("--also-warmup", type=int, default=3,
                    help="untimed steps of every `also` workload")
    ap.add_argument("--launcher", action="store_true", help="self-launch under torch.distributed.run even for --gpus 1")
    ap.add_argument("--sustained-steps", type=int, default=300,
                    help="steps of the `sustained` leg of the headline and padded-text workloads (0: off; also capped at 20 s each)")
    ap.add_argument("--preflight", action="store_true",
                    help="N > 1 only: initialise RCCL, time one all-gather (4 MiB), one reduce-scatter and one 64 MiB all-reduce, print ONE "
                         "JSON line with the per-collective times and exit -- so that a first multi-GPU run that fails says WHERE")
    ap.add_argument("--text-dropout", type=float, default=0.0,
                    help="BERT hidden / attention dropout probability and train() mode (reference default 0.1; BASELINE runs 0)")
    args = ap.parse_args()

    if int(os.environ.get("RANK", "0")) == 0 and not os.environ.get("EZCLIP_BENCH_LAUNCHED") and not os.environ.get("EZCLIP_NO_CANARY"):
        # is the box healthy?  A child process that imports only torch (libezclip_hip.so is not in it); a failure here is the
        # box's, and the message says so before any kernel of this library has run
        import __graft_entry__ as G
        ok, text = G.run_canary()
        print(text.strip(), file=sys.stderr, flush=True)
        if not ok:
            sys.exit(3)

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.launcher):
        relaunch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == max(1, args.gpus), "WORLD_SIZE %d != --gpus %d" % (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    # Dry run of the N > 1 code path on a one-GPU box (tests/test_zz_bench_contract_gpu.py): EZCLIP_BENCH_ONE_GPU=1 puts every rank
    # on cuda:0 and EZCLIP_BENCH_BACKEND=gloo moves the collectives through the host (RCCL refuses two ranks on one device).  The
    # numbers of such a run mean nothing; the driver's runs use neither variable.
    one_gpu = bool(os.environ.get("EZCLIP_BENCH_ONE_GPU"))
    backend = os.environ.get("EZCLIP_BENCH_BACKEND", "nccl")
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or "WORLD_SIZE" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend=backend)

    if args.preflight:
        preflight(world, rank, device, backend, use_dist)
        return

    from easynlp_amd import lib as L
    if os.environ.get("EZCLIP_NO_LNFOLD"):      # A/B switch: separate LayerNorm kernels in the inference path too
        L.check(L.load().ezclip_debug_set(2, 0))
    if os.environ.get("EZCLIP_CLS_LAST"):       # A/B switch: 0 = evaluate the last block of each tower for every token
        L.check(L.load().ezclip_debug_set(3, int(os.environ["EZCLIP_CLS_LAST"])))
    if os.environ.get("EZCLIP_CLS_TRAIN"):      # A/B switch: 0 = the training path evaluates the last blocks for every token
        L.check(L.load().ezclip_debug_set(4, int(os.environ["EZCLIP_CLS_TRAIN"])))
    if os.environ.get("EZCLIP_LNFOLD_MODE"):    # A/B switch: 2 = folded LayerNorm with a separate statistics pass
        L.check(L.load().ezclip_debug_set(2, int(os.environ["EZCLIP_LNFOLD_MODE"])))

    if os.environ.get("EZCLIP_RASTER_GM"):      # A/B switch: tile order of the persistent GEMM (0 column-fastest, g super-rows of g)
        L.check(L.load().ezclip_debug_set(6, int(os.environ["EZCLIP_RASTER_GM"])))
    if os.environ.get("EZCLIP_CLS_Q_ONLY"):     # A/B switch: 0 = the CLS-only last ViT block projects its queries for every token
        L.check(L.load().ezclip_debug_set(8, int(os.environ["EZCLIP_CLS_Q_ONLY"])))
    if os.environ.get("EZCLIP_ATTN_FWD_OPTS"):  # A/B switch: bits of set_attention_short_tail (1 short tail, 2 MFMA row sums, 4 NO full-line stores, 8 NO persistent grid)
        L.check(L.load().ezclip_debug_set(9, int(os.environ["EZCLIP_ATTN_FWD_OPTS"])))
    if os.environ.get("EZCLIP_ATTN_BWD_ONCE"):  # A/B switch: 0 = the two-pass fused attention backward only, 2 = score-tile-once wherever eligible
        L.check(L.load().ezclip_debug_set(11, int(os.environ["EZCLIP_ATTN_BWD_ONCE"])))
    if os.environ.get("EZCLIP_GEMM_DEPHASE"):   # A/B switch: staggered first tiles of the persistent GEMM (steps + 100 * period code; 0 off)
        L.check(L.load().ezclip_debug_set(12, int(os.environ["EZCLIP_GEMM_DEPHASE"])))
    if os.environ.get("EZCLIP_FUSE_QKV"):       # A/B switch: 0 = BERT q / k / v as three products
        L.check(L.load().ezclip_debug_set(7, int(os.environ["EZCLIP_FUSE_QKV"])))

    c = Ctx()
    c.world, c.rank, c.device = world, rank, device
    sus = args.sustained_steps if world == 1 else 0
    head = run_workload(args.workload, c, args.steps, args.warmup, args.batch, args.text_dropout, sustained_steps=sus)
    also = {}
    if not args.no_also:
        names = [n for n in args.also.split(",") if n] or (ALSO_N1 if world == 1 else ALSO_MULTI)
        for n in names:
            if n == args.workload:
                continue
            try:
                r = run_workload(n, c, args.also_steps, args.also_warmup, args.batch, args.text_dropout,
                                 sustained_steps=sus if n == "bf16_b1024_fwd_loss_padded_text" else 0)
            except Exception as e:          # one configuration failing (e.g. out of memory) must not lose the headline
                torch.cuda.empty_cache()
                r = {"workload": n, "error": "%s: %s" % (type(e).__name__, str(e)[:300])}
                if world > 1:
                    raise
            if rank == 0:
                keep = ("value", "ms_per_step", "ms_per_step_hip_events", "host_ms_per_step", "towers", "clock_mhz_timed_steps", "power_w_timed_steps", "dtype", "path", "stages", "pairs_per_gpu", "loss", "text_tower_rows", "model_tflops_per_gpu",
                        "model_mfma_frac", "time_share", "grad_allreduce_buckets_mib", "sustained", "error")
                also[n] = {k: r[k] for k in keep if k in r}
                if r.get("roofline"):
                    also[n]["roofline_frac"] = r["roofline"]["frac"]
                    also[n]["roofline_achieved_tflops"] = r["roofline"]["achieved"]

    if rank == 0:
        wl = WORKLOADS[args.workload]
        out = {
            "metric": "image-text pairs/sec (fwd+loss) " + ("ViT-L/14+roberta-wwm-ext" if wl.get("model") in ("vitl14", "hf_vitl14", "hf_vitl14_large")
                                                            else "ViT-B/16+BERT-base"),
            "value": head["value"], "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": head["dtype"], "data": "synthetic",
            "config": {"workload": args.workload, "towers": head["towers"],
                       "pairs_per_gpu": head["pairs_per_gpu"], "global_batch": world * head["pairs_per_gpu"], "image": "224x224",
                       "seq_len": head["seq_len"], "stages": head["stages"], "path": head["path"],
                       "contrastive_scope": "global" if world > 1 else "local", "parallelism": "dp%d" % world,
                       "two_streams": head["two_streams"], "text_dropout": args.text_dropout},
            "rccl_ranks": world if use_dist else 0, "collective_backend": backend if use_dist else None,
        }
        for k in ("loss", "host_ms_per_step", "ms_per_step_hip_events", "clock_mhz_timed_steps", "power_w_timed_steps", "batches_rotated", "pack_meta_in_timed_region", "ms_per_step_ranks", "text_tower_rows", "gflop_per_pair",
                  "model_tflops_per_gpu", "model_mfma_frac", "roofline", "time_share", "attention_tflops", "layernorm_gbps",
                  "grad_allreduce_buckets_mib", "sustained"):
            if k in head:
                out[k] = head[k]
        if also:
            out["also"] = also
        # The reference-shaped forward, at top level beside `value`: `value` lets the text tower skip padded positions
        # (same embeddings); `value_padded_text` feeds every padded position through it, as the reference does.  Both are measured.
        pt = also.get("bf16_b1024_fwd_loss_padded_text") if args.workload == "bf16_b1024_fwd_loss" else None
        if pt and "value" in pt:
            out["value_padded_text"] = pt["value"]
            out["ms_per_step_padded_text"] = pt["ms_per_step"]
            out["model_mfma_frac_padded_text"] = pt.get("model_mfma_frac")
            out["value_note"] = ("value: text tower over the unmasked tokens only (packed rows, same embeddings) and the last block of each tower on its "
                                 "CLS row only -- work the result does not depend on is skipped; value_padded_text: THE REFERENCE-SHAPED FIGURE, every "
                                 "padded position goes through the text tower as in the reference; model_mfma_frac* count EXECUTED flops")
            out["reference_shaped"] = {"value": pt["value"], "ms_per_step": pt["ms_per_step"], "model_mfma_frac": pt.get("model_mfma_frac"),
                                       "workload": "bf16_b1024_fwd_loss_padded_text"}
        if world == 1 and not args.no_recall:
            try:
                rf, rw, rin = recall_leg(device)
                out.update(rf)
                out["recall_at_1_oracle"] = None
                if not args.no_cpu_baseline:
                    out["recall_at_1_oracle"] = round(recall_oracle(rw, rin), 6)
                    out["recall_at_1_abs_diff"] = round(abs(out["recall_at_1"] - out["recall_at_1_oracle"]), 6)
            except Exception as e:          # noqa: BLE001 -- the recall figure must not take the throughput line with it
                out["recall_at_1"], out["recall_error"] = None, "%s: %s" % (type(e).__name__, str(e)[:300])
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

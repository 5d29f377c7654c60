#!/usr/bin/env python
"""Headline benchmark: image-text pairs/sec of the CLIP retrieval hot path
(ViT-B/16 + BERT-base dual-encoder forward + InfoNCE loss) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path over one synthetic batch that is already
resident in HBM: encode_image + encode_text (+ RCCL all-gather of both embedding
sets when N > 1) + similarity (both directions) + InfoNCE.  Rank 0 prints ONE
JSON line.  Workloads (BASELINE.json configs):

  bf16_b1024_fwd_loss   (default) ViT-B/16 + BERT-base, bf16 MFMA, 1024 pairs/GPU, 64 tokens
  bf16_b1024_train      same + full backward (only when the backward kernels are built)
  fp32_b256_fwd_sim     config 2: exact-f32 MFMA, 256 pairs, forward + similarity

`roofline` describes the dominant kernel (the MFMA GEMM): algorithmic FLOPs of its
launches / their HIP-event durations, measured live on the launch stream in extra
(untimed) steps.  `cpu_baseline` times the CPU oracle (a torch-CPU port of the
reference algorithm) on the host cores on a bounded sample of the same workload.
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

WORKLOADS = {
    "bf16_b1024_fwd_loss": dict(dtype="bf16", batch=1024, seq=64, backward=False),
    "bf16_b1024_train": dict(dtype="bf16", batch=1024, seq=64, backward=True),
    "fp32_b256_fwd_sim": dict(dtype="fp32", batch=256, seq=64, backward=False),
    "bf16_vitl14_b512_fwd_loss": dict(dtype="bf16", batch=512, seq=64, backward=False, model="vitl14"),
    "bf16_vitl14_b512_train": dict(dtype="bf16", batch=512, seq=64, backward=True, model="vitl14"),
    "bf16_hf_vitl14_b512_train": dict(dtype="bf16", batch=512, seq=64, backward=True, model="hf_vitl14"),
}

# algorithmic GFLOP per pair (multiply-add = 2; padded tiles and softmax/LN excluded)
GFLOP_FWD_PER_PAIR = 46.152
# Forward-only workloads evaluate the last block of each tower for the CLS rows only (nothing else reaches the embeddings):
# out_proj + MLP + attention core of one ViT block for 196 of 197 tokens (2.199 G) and query / attention / output / FFN of one
# BERT layer for 63 of 64 tokens (0.756 G) are not executed.  model_tflops is computed from the EXECUTED figure.
GFLOP_FWD_EXECUTED_PER_PAIR = 46.152 - 2.199 - 0.756
GFLOP_TRAIN_EXECUTED_PER_PAIR = 138.46 - 3 * (2.199 + 0.756)     # the same rows are skipped in the backward pass
GFLOP_TRAIN_PER_PAIR = 138.46
GFLOP_FWD_PER_PAIR_VITL14 = 173.05      # ViT-L/14 (L = 257) + BERT-base text tower, 64 tokens
GFLOP_TRAIN_PER_PAIR_VITL14 = 519.2
GFLOP_TRAIN_PER_PAIR_HF_VITL14 = 162.03 + 3 * 11.025      # frozen vision tower: forward only; text tower fwd + bwd
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md

# launches of the dominant kernel in one forward step at 1024 pairs (shape names of tools/gemm_bench):
FWD_GEMM_MIX = {"vit.qkv": 12, "vit.out+res": 12, "vit.fc+qgelu": 12, "vit.proj+res": 12, "bert.qkvo+res": 48,
                "bert.ffn1+gelu": 12, "bert.ffn2+res": 12, "patch": 1}


def pmc_traffic(workload):
    """Average HBM-side bytes per launch of the dominant kernel (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc passes of
    tools/pmc_gemm.sh summarised by tools/pmc_summary.py into profiles/*_gemm_traffic.json; MALL hits are counted as
    fetches on gfx950).  Measured offline with the same kernels, shapes and batch: bench.py cannot run a profiler
    around itself.  None when no summary is committed or the workload is not the 1024-pair forward."""
    if workload != "bf16_b1024_fwd_loss":
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_gemm_traffic.json")))
    if not files:
        return None
    t = json.load(open(files[-1]))
    tot, n = 0.0, 0
    for name, cnt in FWD_GEMM_MIX.items():
        if name not in t:
            return None
        tot += cnt * (t[name]["fetch_bytes"] + t[name]["write_bytes"])
        n += cnt
    return {"bytes_per_launch": round(tot / n), "algorithmic_bytes_per_launch":
            round(sum(c * t[k]["algorithmic_bytes"] for k, c in FWD_GEMM_MIX.items()) / n),
            "source": os.path.basename(files[-1])}


def synth_batch(batch, seq, vocab, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    px = torch.randn((batch, 3, 224, 224), generator=g, device=device, dtype=torch.float32)
    ids = torch.randint(1, vocab, (batch, seq), generator=g, device=device, dtype=torch.int64)
    lens = torch.randint(8, seq + 1, (batch,), generator=g, device=device)
    ids = ids * (torch.arange(seq, device=device)[None, :] < lens[:, None])
    return px, ids


def cpu_baseline(seconds_target=12.0):
    """The CPU oracle (torch-CPU restatement of the reference path, oracle/clip_oracle.py)
    on the host cores: fp32 forward + similarity + InfoNCE on 8-pair batches."""
    from oracle import clip_oracle as O
    # a small-batch CPU forward stops scaling (and collapses from oversubscription) beyond a few
    # dozen threads: use at most 32 and report the number actually used
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    cfg = O.CONFIGS["vitb16_bertbase"]
    sd = O.make_state_dict(cfg, 1234)
    px, ids = O.make_inputs(cfg, 8, 64, 0)
    with torch.no_grad():
        out = O.clip_forward(sd, cfg, px, ids)       # warm-up
        O.clip_loss(out["logits_per_text"])
        n, t0 = 0, time.perf_counter()
        while True:
            out = O.clip_forward(sd, cfg, px, ids)
            O.clip_loss(out["logits_per_text"])
            n += 1
            el = time.perf_counter() - t0
            if el >= seconds_target or n >= 40:
                break
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(8 * n / el, 3), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "%d x (8 pairs, 224x224 + 64 tokens) fp32 fwd+similarity+InfoNCE, torch CPU %s, %s"
                      % (n, torch.__version__, model)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="bf16_b1024_fwd_loss", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override pairs per GPU (debugging only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--text-dropout", type=float, default=0.0,
                    help="BERT hidden / attention dropout probability and train() mode (reference default 0.1; BASELINE runs 0)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        sys.exit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                 "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    assert world == max(1, args.gpus), "WORLD_SIZE %d != --gpus %d" % (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)   # "nccl" is RCCL on ROCm

    from easynlp_amd import lib as L
    from easynlp_amd.appzoo.clip import CLIPApp

    if os.environ.get("EZCLIP_NO_LNFOLD"):      # A/B switch: separate LayerNorm kernels in the inference path too
        L.check(L.load().ezclip_debug_set(2, 0))
    if os.environ.get("EZCLIP_CLS_LAST"):       # A/B switch: 0 = evaluate the last block of each tower for every token
        L.check(L.load().ezclip_debug_set(3, int(os.environ["EZCLIP_CLS_LAST"])))
    if os.environ.get("EZCLIP_CLS_TRAIN"):      # A/B switch: 0 = the training path evaluates the last blocks for every token
        L.check(L.load().ezclip_debug_set(4, int(os.environ["EZCLIP_CLS_TRAIN"])))
    if os.environ.get("EZCLIP_LNFOLD_MODE"):    # A/B switch: 2 = folded LayerNorm with a separate statistics pass
        L.check(L.load().ezclip_debug_set(2, int(os.environ["EZCLIP_LNFOLD_MODE"])))
    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["batch"] = args.batch
    B, S = wl["batch"], wl["seq"]
    model_cfg = VITL14_ROBERTA if wl.get("model") == "vitl14" else VITB16_BERTBASE
    model_name = ("ViT-L/14 + chinese-roberta-wwm-ext (BERT-base arch)" if wl.get("model") == "vitl14"
                  else "ViT-B/16 + BERT-base") + " (chinese_clip), random init"
    if args.text_dropout > 0:
        model_cfg = dict(model_cfg, text_hidden_dropout_prob=args.text_dropout,
                         text_attention_probs_dropout_prob=args.text_dropout)
    if wl.get("model") == "hf_vitl14":
        app = CLIPApp.from_hf_config(HF_VITL14_ROBERTA, seed=1234, device=device, compute_dtype=wl["dtype"])
        model_name = "huggingface_clip: frozen ViT-L/14 + chinese-roberta-wwm-ext (BERT-base arch) + pooler, random init"
    else:
        app = CLIPApp.from_config(model_cfg, seed=1234, device=device, compute_dtype=wl["dtype"])
    app.eval()
    if args.text_dropout > 0:
        app.train()
    px, ids = synth_batch(B, S, VITB16_BERTBASE["vocab_size"], device, seed=1000 + rank)
    pg = True if world > 1 else False

    def step():
        if wl["backward"]:
            for p in app.parameters():
                if p.grad is not None:
                    p.grad.zero_()
            loss = app.contrastive_step(px, ids, process_group=pg, backward=True)
            if world > 1:       # the gradient all-reduce belongs to a training step (DDP in trainer.py:101-108); the embedding
                from easynlp_amd import parallel as P    # gradients are those of the global mean loss, so ranks hold partial sums
                P.sum_gradients(list(app.parameters()))
            return loss
        with torch.no_grad():
            return app.contrastive_step(px, ids, process_group=pg, backward=False)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss.item())

    # ---- roofline leg: per-launch HIP events around the dominant kernel (untimed extra steps) ----
    lib = L.load()
    roof = None
    extra = {}
    nprof = 2
    if rank == 0:
        L.check(lib.ezclip_profile_begin())
    for _ in range(nprof):      # every rank steps (the collectives need all of them); only rank 0 records events
        step()
    fence()
    if rank == 0:
        res = {}
        for cls, name in ((0, "gemm"), (1, "attention"), (2, "layernorm")):
            ms, work, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
            L.check(lib.ezclip_profile_end(cls, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(n)))
            res[name] = (ms.value, work.value, n.value)
        ms, flops, n = res["gemm"]
        peak = PEAK_TFLOPS[wl["dtype"]]
        ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": (pmc_traffic(args.workload) or {}).get("bytes_per_launch"),
                "traffic_detail": pmc_traffic(args.workload),
                "kernel": ("ezclip::gemm_nt_8p_kernel / gemm_tn_8p_kernel (bf16, 256x256x64 8-phase)"
                           if wl["dtype"] == "bf16" else "ezclip::gemm_nt_kernel<fp32> (128x128, exact-f32 MFMA)"),
                "launches_per_step": n // nprof, "avg_launch_us": round(ms * 1e3 / max(n, 1), 2),
                "algorithmic_gflop_per_launch": round(flops / max(n, 1) / 1e9, 3)}
        step_ms = elapsed / args.steps * 1e3
        extra["time_share"] = {k: round(v[0] / nprof / step_ms, 4) for k, v in res.items()}
        a_ms, a_fl, a_n = res["attention"]
        extra["attention_tflops"] = round(a_fl / (a_ms * 1e-3) / 1e12, 2) if a_ms > 0 else None
        l_ms, l_by, l_n = res["layernorm"]
        extra["layernorm_gbps"] = round(l_by / (l_ms * 1e-3) / 1e9, 1) if l_ms > 0 else None

    if rank == 0:
        pairs = world * B * args.steps
        value = pairs / elapsed
        if wl.get("model") == "hf_vitl14":
            gflop = GFLOP_TRAIN_PER_PAIR_HF_VITL14
        elif wl.get("model") == "vitl14":
            gflop = GFLOP_TRAIN_PER_PAIR_VITL14 if wl["backward"] else GFLOP_FWD_PER_PAIR_VITL14
        else:
            cls_last = os.environ.get("EZCLIP_CLS_LAST", "1") != "0"
            cls_train = os.environ.get("EZCLIP_CLS_TRAIN", "1") != "0"
            if wl["backward"]:
                gflop = GFLOP_TRAIN_EXECUTED_PER_PAIR if cls_train else GFLOP_TRAIN_PER_PAIR
            else:
                gflop = GFLOP_FWD_EXECUTED_PER_PAIR if cls_last else GFLOP_FWD_PER_PAIR
        out = {
            "metric": "image-text pairs/sec (fwd+loss) " + ("ViT-L/14+roberta-wwm-ext" if wl.get("model") in ("vitl14", "hf_vitl14")
                                                            else "ViT-B/16+BERT-base"),
            "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": wl["dtype"], "data": "synthetic",
            "config": {"workload": args.workload, "towers": model_name,
                       "pairs_per_gpu": B, "global_batch": world * B, "image": "224x224", "seq_len": S,
                       "stages": "encode_image+encode_text" + ("+allgather" if world > 1 else "")
                                 + "+similarity(2 dirs)+InfoNCE" + ("+backward" if wl["backward"] else ""),
                       "contrastive_scope": "global" if world > 1 else "local", "parallelism": "dp%d" % world,
                       "text_dropout": args.text_dropout},
            "loss": round(loss_val, 5),
            "gflop_per_pair": {"algorithmic_all_tokens": (GFLOP_TRAIN_PER_PAIR if wl["backward"] else GFLOP_FWD_PER_PAIR)
                               if wl.get("model") is None else None, "executed": round(gflop, 3)},
            "model_tflops_per_gpu": round(value / world * gflop / 1e3, 2),
            "model_mfma_frac": round(value / world * gflop / 1e3 / PEAK_TFLOPS[wl["dtype"]], 4),
            "roofline": roof,
        }
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""bench.py's one JSON line (the driver's contract) on a small batch: required keys, types, the roofline object, the
`also` object that carries the other BASELINE configurations -- and the LOSS the line reports against the CPU oracle run
on the same synthetic batch and the same random-init weights (a bench that computed garbage fast would pass the key
checks; ln(batch) alone would survive large errors, the oracle's value does not)."""
import json
import math
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "64",
                        "--no-cpu-baseline"] + list(extra), capture_output=True, text=True, cwd=ROOT, timeout=1200,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _oracle_loss(batch):
    """The oracle on bench.py's synthetic batch and weights (same seeds, same device generator)."""
    sys.path.insert(0, ROOT)
    import bench as B
    from easynlp_amd.appzoo.clip import CLIPApp
    from oracle import clip_oracle as O
    app = CLIPApp.from_config(B.VITB16_BERTBASE, seed=1234, device="cuda", compute_dtype="fp32")
    sd = {k: v.detach().cpu() for k, v in app._params.items()}
    px, ids = B.synth_batch(batch, 64, B.VITB16_BERTBASE["vocab_size"], torch.device("cuda"), seed=1000)
    with torch.no_grad():
        out = O.clip_forward(sd, O.CONFIGS["vitb16_bertbase"], px.cpu(), ids.cpu())
        return float(O.clip_loss(out["logits_per_text"]))


def test_bench_prints_one_contract_line():
    out = _run("--also", "bf16_b1024_train,bf16_b1024_fwd_loss_autograd,bf16_b1024_train_autograd,bf16_b1024_train_opt,"
               "bf16_b1024_fwd_loss_padded_text,bf16_rn50_b256_fwd,bf16_rn50_b256_train", "--also-steps", "1", "--sustained-steps", "50")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "also", "loss"):
        assert k in out, k
    assert out["unit"] == "pairs/s" and out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["dtype"] == "bf16" and out["data"] == "synthetic" and out["value"] > 0 and out["ms_per_step"] > 0
    assert out["config"]["workload"] == "bf16_b1024_fwd_loss" and "model" not in out["config"] and out["config"]["pairs_per_gpu"] == 64
    roof = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 2500.0
    assert 0 < roof["frac"] < 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    # the loss of every workload on this batch: the oracle's value (bf16 bound of the model tests: 5e-3 at full size;
    # a training step with the optimizer reports the loss BEFORE its single update, so it is the same number)
    ref = _oracle_loss(64)
    assert abs(ref - math.log(64)) < 0.2                      # random init: near ln N, but not equal to it
    assert abs(out["loss"] - ref) < 5e-3, (out["loss"], ref)
    for name, a in out["also"].items():
        assert "error" not in a, (name, a)
        assert a["value"] > 0 and a["ms_per_step"] > 0 and a["ms_per_step_hip_events"] > 0
        if name.startswith("bf16_rn50"):      # another model (ModifiedResNet-50 image tower): its own random-init loss, near ln N
            assert abs(a["loss"] - math.log(64)) < 0.5 and a["path"] == "autograd" and "ModifiedResNet-50" in a["towers"], (name, a)
            assert a["model_tflops_per_gpu"] > 0 and "clock_mhz_timed_steps" in a and "power_w_timed_steps" in a
            continue
        # (the AdamW workload has already moved the weights by the time its timed step reports a loss)
        assert abs(a["loss"] - ref) < (5e-2 if name.endswith("_opt") else 5e-3), (name, a["loss"], ref)
    # round 6: the line carries what the metric names ("...; R@1 vs ref"), the ModifiedResNet workloads, HIP-event timing beside the host clock
    assert 0.3 <= out["recall_at_1"] <= 1.0 and "recall_at_1_oracle" in out and out["recall_detail"]["pairs"] == 96
    assert {"bf16_rn50_b256_fwd", "bf16_rn50_b256_train"} <= set(out["also"])
    assert out["ms_per_step_hip_events"] > 0 and abs(out["ms_per_step_hip_events"] - out["ms_per_step"]) < 0.25 * out["ms_per_step"] + 2.0
    assert out["reference_shaped"]["value"] == out["value_padded_text"] and out["sustained"]["ms_per_step_hip_events"] > 0
    assert out["also"]["bf16_b1024_train_autograd"]["path"] == "autograd"
    # the reference-shaped forward (every padded position through the text tower) at top level, beside `value`
    assert out["value_padded_text"] == out["also"]["bf16_b1024_fwd_loss_padded_text"]["value"] and "value_note" in out
    # the `sustained` block (round 4): the same step for N more steps, with the clock / power the chip ran at -- in the headline and
    # in the padded-text workload
    for blk in (out["sustained"], out["also"]["bf16_b1024_fwd_loss_padded_text"]["sustained"]):
        assert blk["steps"] == 50 and blk["ms_per_step"] > 0 and blk["ms_per_step_second_half"] > 0 and blk["value_second_half"] > 0
        assert 0 < blk["model_mfma_frac_second_half"] < 1
        tel = blk["telemetry"]
        assert tel["source"] and (tel["samples"] == 0 or (100 < tel["shader_clock_mhz_mean"] < 3000 and 50 < tel["socket_power_w_mean"] < 2000))


def test_recall_leg_agrees_with_the_oracle_evaluator():
    """the bench line's `recall_at_1` (fused similarity + rank kernel on the bf16 HIP path's embeddings) against `recall_at_1_oracle` (the
    reference evaluator's full-sort arithmetic on the fp32 CPU oracle's embeddings of the SAME trained weights): within 1e-3 (north star)"""
    sys.path.insert(0, ROOT)
    import bench as B
    fields, weights, inputs = B.recall_leg(torch.device("cuda", 0))
    want = B.recall_oracle(weights, inputs)
    assert abs(fields["recall_at_1"] - want) <= 1e-3, (fields, want)
    assert abs(want - 88.0 / 96.0) < 1e-6, want          # 8 duplicated captions: exactly one query of each such pair ranks its own image first


def test_bench_self_launches_under_torch_distributed_run():
    """`python bench.py --gpus N` must work unattended (no external launcher): with --launcher the same code path runs for
    N = 1 -- re-exec under torch.distributed.run on 127.0.0.1, RCCL process group, one JSON line from rank 0."""
    out = _run("--launcher", "--no-also")
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["value"] > 0
    assert out["config"]["parallelism"] == "dp1"


def test_bench_two_rank_code_path_dry_run_on_one_gpu():
    """`bench.py --gpus 2` end to end on this one-GPU box: both ranks on cuda:0, collectives through gloo (EZCLIP_BENCH_ONE_GPU /
    EZCLIP_BENCH_BACKEND, bench.py) -- self-launch under torch.distributed.run, per-rank batches, the all-gather + tiled contrastive
    step of the global batch, barrier, MAX over ranks, the one JSON line with the
    per-rank timings.  (The training step with the overlapped bucketed gradient all-reduce on several ranks is
    tests/test_zz_two_ranks_one_gpu.py: through gloo a 755 MB arena per step would take minutes here.)  The driver's multi-GPU run is
    the first time this path meets RCCL; it should not also be the first time it runs."""
    env = {"EZCLIP_BENCH_ONE_GPU": "1", "EZCLIP_BENCH_BACKEND": "gloo", "EZCLIP_NO_CANARY": "1"}
    out = _run("--gpus", "2", "--no-also", env=env)
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["collective_backend"] == "gloo"
    assert out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 128 and out["config"]["contrastive_scope"] == "global"
    assert out["value"] > 0 and out["scaling"] == "weak" and "allgather" in out["config"]["stages"]
    r = out["ms_per_step_ranks"]
    assert r["min"] <= r["max"] + 1e-9 and r["max"] == out["ms_per_step"]
    assert abs(out["loss"] - math.log(128)) < 0.3                 # this rank's rows against the 128 columns of the global batch


def test_bench_preflight_names_the_collectives():
    """`bench.py --gpus 2 --preflight` (round 5): process-group init, one all-gather, one reduce-scatter (an all-reduce on gloo), one
    64 MiB all-reduce, each timed on its own, ONE JSON line -- here two ranks on this one GPU through gloo; the driver's 8-GPU box is
    the first time it meets RCCL."""
    env = {"EZCLIP_BENCH_ONE_GPU": "1", "EZCLIP_BENCH_BACKEND": "gloo", "EZCLIP_NO_CANARY": "1"}
    out = _run("--gpus", "2", "--preflight", env=env)
    assert out["preflight"] and out["rccl_ranks"] == 2 and out["failed_at"] is None and out["all_ranks_ok"], out
    st = out["stages"]
    assert st["all_gather_4MiB"]["ms"] > 0 and st["all_gather_4MiB"]["own_rows_intact"]
    assert st["all_reduce_64MiB"]["ms"] > 0 and st["all_reduce_64MiB"]["finite"] and st["all_reduce_64MiB"]["bus_gbps"] > 0
    assert any(k.startswith("reduce_scatter_4MiB") for k in st) and st["barrier"]["ms"] >= 0

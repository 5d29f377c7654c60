#!/usr/bin/env python
"""Turn the rocprofv3 --pmc CSVs written by tools/pmc_gemm.sh into a per-launch table (markdown) and a small JSON
with the HBM-side traffic of the dominant kernel.   usage: pmc_summary.py <tag> [outdir=profiles]
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane streaming reads on gfx950 (the counter
tallies 128-byte requests at 64 B); WRITE_SIZE is used as is (it reproduces the algorithmic output bytes exactly)."""
import collections
import csv
import json
import os
import sys

SHAPES = ["vit.qkv", "vit.out+res", "vit.fc+qgelu", "vit.proj+res", "bert.qkvo+res", "bert.ffn1+gelu", "bert.ffn2+res",
          "patch", "train.fc+c2", "ragged.M", "bwd.dgrad*act'", "bwd.gelu' ragged", "small.M=788", "bert.qkv",
          "wgrad.out", "wgrad.qkv", "wgrad.fc", "wgrad.proj", "wgrad.bert.ffn1", "wgrad.ragged"]
DIMS = {"vit.qkv": (201728, 2304, 768), "vit.out+res": (201728, 768, 768), "vit.fc+qgelu": (201728, 3072, 768),
        "vit.proj+res": (201728, 768, 3072), "bert.qkvo+res": (65536, 768, 768), "bert.ffn1+gelu": (65536, 3072, 768),
        "bert.ffn2+res": (65536, 768, 3072), "patch": (200704, 768, 768), "train.fc+c2": (201728, 3072, 768),
        "ragged.M": (201628, 768, 768), "bwd.dgrad*act'": (201728, 3072, 768), "bwd.gelu' ragged": (788, 3072, 768),
        "small.M=788": (788, 768, 3072), "bert.qkv": (65536, 2304, 768), "wgrad.out": (201728, 768, 768), "wgrad.qkv": (201728, 2304, 768),
        "wgrad.fc": (201728, 3072, 768), "wgrad.proj": (201728, 768, 3072), "wgrad.bert.ffn1": (65536, 3072, 768),
        "wgrad.ragged": (201651, 768, 768)}


def load(path):
    d = collections.OrderedDict()
    if not os.path.exists(path):
        return d
    for r in csv.DictReader(open(path)):
        if "gemm_nt_8p" not in r["Kernel_Name"] and "gemm_tn_8p" not in r["Kernel_Name"]:
            continue
        k = int(r["Dispatch_Id"])
        e = d.setdefault(k, {"kernel": "tn" if "gemm_tn_8p" in r["Kernel_Name"] else "nt",
                             "dur_us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    return d


def per_shape(d):
    """gemm_bench launches every shape 1 (check) + iters (timed) times with variant 2, in SHAPES order."""
    out, groups, last = collections.OrderedDict(), [], None
    rows = list(d.values())
    # consecutive launches of one shape have the same duration class; split by the known launch count (2 each)
    for i in range(0, len(rows) - 1, 2):
        groups.append(rows[i:i + 2])
    for name, g in zip(SHAPES, groups):
        out[name] = g[-1]
    return out


def main():
    tag = sys.argv[1]
    outdir = sys.argv[2] if len(sys.argv) > 2 else "profiles"
    sq = per_shape(load("gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES_%s.csv" % tag))
    lds = per_shape(load("gpurun_out/pmc_SQ_LDS_BANK_CONFLICT_%s.csv" % tag))
    fe = per_shape(load("gpurun_out/pmc_FETCH_SIZE_%s.csv" % tag))
    wr = per_shape(load("gpurun_out/pmc_WRITE_SIZE_%s.csv" % tag))
    lines = ["# rocprofv3 PMC summary of the 8-phase GEMM kernels (%s), one launch per row" % tag, "",
             "Shapes of `tools/gemm_bench 1024` (batch 1024: ViT M = 201 728 tokens, BERT M = 65 536).  Clock = "
             "GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles); the wait "
             "columns are fractions of SQ_WAVE_CYCLES; fetch = 2 x FETCH_SIZE (gfx950 correction, MALL hits included), "
             "write = WRITE_SIZE; algorithmic = A + B (+ residual) once + C once.", "",
             "| shape | M,N,K | dur us | TFLOP/s | clock GHz | MFMA util | wait_any | wait_inst | active | LDS conflict cyc | fetch MB | write MB | algorithmic MB |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    traffic = {}
    for name in SHAPES:
        M, N, K = DIMS[name]
        s, l, f, w = sq.get(name), lds.get(name), fe.get(name), wr.get(name)
        if s is None and f is None:
            continue
        dur = (s or f)["dur_us"]
        tf = 2.0 * M * N * K / dur / 1e6
        cells = [name, "%d,%d,%d" % (M, N, K), "%.0f" % dur, "%.0f" % tf]
        if s and "GRBM_GUI_ACTIVE" in s:
            cyc = s["GRBM_GUI_ACTIVE"] / 8
            wc = s["SQ_WAVE_CYCLES"]
            cells += ["%.2f" % (cyc / s["dur_us"] / 1e3), "%.3f" % (s["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)),
                      "%.2f" % (s["SQ_WAIT_ANY"] / wc), "%.2f" % (s["SQ_WAIT_INST_ANY"] / wc), "%.2f" % (s["SQ_ACTIVE_INST_ANY"] / wc)]
        else:
            cells += ["-"] * 5
        cells.append("%.0f" % l["SQ_LDS_BANK_CONFLICT"] if l and "SQ_LDS_BANK_CONFLICT" in l else "-")
        fmb = f["FETCH_SIZE"] * 2 / 1024 if f and "FETCH_SIZE" in f else None
        wmb = w["WRITE_SIZE"] / 1024 if w and "WRITE_SIZE" in w else None
        if name.startswith("wgrad"):
            alg = (M * N + M * K) * 2 / 2 ** 20 + N * K * 4 / 2 ** 20
        else:
            alg = (M * K + N * K + M * N * (1 + ("res" in name) + ("act'" in name or "gelu'" in name) + ("c2" in name))) * 2 / 2 ** 20
        cells += ["%.0f" % fmb if fmb is not None else "-", "%.0f" % wmb if wmb is not None else "-", "%.0f" % alg]
        lines.append("| " + " | ".join(cells) + " |")
        if fmb is not None and wmb is not None:
            traffic[name] = {"fetch_bytes": fmb * 2 ** 20, "write_bytes": wmb * 2 ** 20, "algorithmic_bytes": alg * 2 ** 20}
    os.makedirs(outdir, exist_ok=True)
    open(os.path.join(outdir, "%s_gemm_pmc.md" % tag), "w").write("\n".join(lines) + "\n")
    json.dump(traffic, open(os.path.join(outdir, "%s_gemm_traffic.json" % tag), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()

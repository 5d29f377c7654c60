#!/usr/bin/env python
"""Side-by-side table of the rocprofv3 --pmc passes of tools/pmc_vendor_vs_ours.sh: the hand-written 8-phase GEMM (`ours`) and
hipBLASLt's kernel behind torch.nn.functional.linear (`vendor`, calibration only) on the same shapes and operand fills.
    usage: pmc_vendor_vs_ours.py <tag> [out.md]
Counters are summed over the dispatch (rocprofv3 reports per-dispatch totals across XCDs / SEs); durations are the kernel-trace
timestamps of the counter run (profiled launches run a few % slower than unprofiled ones -- compare columns, not with gemm_bench).
FETCH_SIZE is doubled (gfx950: the counter tallies 128-byte requests at 64 B, MI355X_MICROARCH.md), Infinity-Cache hits included."""
import collections
import csv
import glob
import os
import sys

SHAPES = [("vit.qkv", 201728, 2304, 768), ("vit.out(+res)", 201728, 768, 768), ("vit.fc(+qgelu)", 201728, 3072, 768),
          ("vit.proj(+res)", 201728, 768, 3072), ("bert.ffn1(+gelu)", 65536, 3072, 768)]
OURS_ORDER = ["vit.qkv", "vit.out(+res)", "vit.fc(+qgelu)", "vit.proj(+res)", "bert.qkvo+res", "bert.ffn1(+gelu)"]   # gemm_bench NT_SHAPES=6


def load(path, want):
    d = collections.OrderedDict()
    if not os.path.exists(path):
        return d
    for r in csv.DictReader(open(path)):
        if not want(r["Kernel_Name"]):
            continue
        k = int(r["Dispatch_Id"])
        e = d.setdefault(k, {"kernel": r["Kernel_Name"], "dur_us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                             "grid": r.get("Grid_Size"), "wg": r.get("Workgroup_Size"), "lds": r.get("LDS_Block_Size"),
                             "vgpr": r.get("VGPR_Count"), "agpr": r.get("Accum_VGPR_Count"), "sgpr": r.get("SGPR_Count")})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return d


def ours_rows(tag, p):
    d = list(load("gpurun_out/pmcvo_ours_p%d_%s.csv" % (p, tag), lambda n: "gemm_nt_8p" in n).values())
    out = {}
    for i, name in enumerate(OURS_ORDER):            # two launches per shape (check + timed): take the second
        if 2 * i + 1 < len(d):
            out[name] = d[2 * i + 1]
    return out


def vendor_rows(tag, p):
    d = list(load("gpurun_out/pmcvo_vendor_p%d_%s.csv" % (p, tag), lambda n: "Cijk_" in n).values())
    out = {}
    per = len(d) // len(SHAPES) if d else 0          # 3 warm-up + ITERS launches per shape
    for i, (name, *_r) in enumerate(SHAPES):
        if per and (i + 1) * per - 1 < len(d):
            out[name] = d[(i + 1) * per - 1]
    return out


def main():
    tag = sys.argv[1]
    npass = len(glob.glob("gpurun_out/pmcvo_ours_p*_%s.csv" % tag))
    ours = [ours_rows(tag, p) for p in range(6)]
    vend = [vendor_rows(tag, p) for p in range(6)]
    lines = ["# Vendor GEMM vs the hand-written 8-phase kernel under the same PMC passes (%s)" % tag, "",
             "`tools/pmc_vendor_vs_ours.sh %s` on one MI355X: `tools/bin/gemm_bench 1024 1 2` (ours, fused epilogues) and "
             "`tools/vendor_calibration.py` (hipBLASLt through `torch.nn.functional.linear`, bias epilogue only; calibration, never "
             "product), one launch per shape and counter pass, uniform random operands.  %d counter files per side." % (tag, npass), ""]
    v0 = vend[0]
    if v0:
        lines += ["Vendor kernels selected (Kernel_Name, grid / workgroup, LDS, VGPR / AGPR):", ""]
        for name, *_ in SHAPES:
            if name in v0:
                e = v0[name]
                lines.append("* `%s`: `%s` grid %s wg %s lds %s vgpr %s agpr %s" % (name, e["kernel"][:220], e["grid"], e["wg"], e["lds"], e["vgpr"], e["agpr"]))
        lines.append("")
    o0 = ours[0]
    if o0:
        e = next(iter(o0.values()))
        lines += ["Ours: `%s` grid %s wg %s lds %s vgpr %s agpr %s" % (e["kernel"][:120], e["grid"], e["wg"], e["lds"], e["vgpr"], e["agpr"]), ""]

    def g(rows, p, name, key):
        try:
            return rows[p][name].get(key)
        except (KeyError, IndexError):
            return None

    hdr = ["shape", "side", "us", "TF", "GHz", "MFMA busy", "wave cyc (M quad)", "parked", "issue-stalled", "issuing", "INSTS_MFMA (M)", "INSTS_LDS (M)",
           "INSTS_VALU (M)", "VMEM_RD (M)", "VMEM_WR (M)", "SALU (M)", "SMEM (M)", "LDS bank conf / idx active", "fetch MB", "write MB", "L2 hit"]
    lines += ["| " + " | ".join(hdr) + " |", "|" + "---|" * len(hdr)]
    for name, M, N, K in SHAPES:
        for side, rows in (("ours", ours), ("vendor", vend)):
            us = g(rows, 0, name, "dur_us")
            if us is None:
                continue
            gui = g(rows, 0, name, "GRBM_GUI_ACTIVE")
            wc = g(rows, 0, name, "SQ_WAVE_CYCLES") or 0
            busy = g(rows, 0, name, "SQ_VALU_MFMA_BUSY_CYCLES")
            ghz = gui / 8.0 / (us * 1e3) if gui else None
            cyc = gui / 8.0 if gui else None

            def m(p, key):
                v = g(rows, p, name, key)
                return "%.2f" % (v / 1e6) if v is not None else "-"

            def frac(key):
                v = g(rows, 0, name, key)
                return "%.2f" % (v / wc) if v is not None and wc else "-"

            fe, wr = g(rows, 3, name, "FETCH_SIZE"), g(rows, 4, name, "WRITE_SIZE")
            hit, miss = g(rows, 5, name, "TCC_HIT_sum"), g(rows, 5, name, "TCC_MISS_sum")
            bc, ia = g(rows, 1, name, "SQ_LDS_BANK_CONFLICT"), g(rows, 1, name, "SQ_LDS_IDX_ACTIVE")
            lines.append("| " + " | ".join([
                name, side, "%.0f" % us, "%.0f" % (2.0 * M * N * K / us / 1e6), "%.2f" % ghz if ghz else "-",
                "%.2f" % (busy / (1024.0 * cyc)) if busy and cyc else "-", "%.1f" % (wc / 1e6), frac("SQ_WAIT_ANY"), frac("SQ_WAIT_INST_ANY"),
                frac("SQ_ACTIVE_INST_ANY"), m(1, "SQ_INSTS_MFMA"), m(1, "SQ_INSTS_LDS"), m(1, "SQ_INSTS_VALU"), m(1, "SQ_INSTS_VMEM_RD"),
                m(1, "SQ_INSTS_VMEM_WR"), m(1, "SQ_INSTS_SALU"), m(2, "SQ_INSTS_SMEM"),
                "%.3f" % (bc / ia) if bc is not None and ia else "-",
                "%.0f" % (2 * fe * 1024 / 1e6) if fe is not None else "-", "%.0f" % (wr * 1024 / 1e6) if wr is not None else "-",
                "%.2f" % (hit / (hit + miss)) if hit is not None and miss is not None and hit + miss > 0 else "-"]) + " |")
    lines += ["", "Algorithmic bytes (A + B + C [+ residual]) for orientation: " +
              ", ".join("%s %.0f MB" % (n, (M * K + N * K + M * N) * 2 / 1e6) for n, M, N, K in SHAPES) +
              " (+ M x N x 2 for the residual the hand-written out / proj products also read).", ""]
    out = sys.argv[2] if len(sys.argv) > 2 else "profiles/%s_vendor_vs_ours_pmc.md" % tag
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()

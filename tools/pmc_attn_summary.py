#!/usr/bin/env python
"""Per-kernel means of the counters tools/pmc_attn.sh / pmc_attn_train.sh / pmc_rn_wgrad.sh collected.
usage: [PMC_FILTER=<regex of kernel names>] pmc_attn_summary.py <tag>      (default filter: the attention kernels)"""
import collections
import csv
import glob
import os
import re
import sys

tag = sys.argv[1]
filt = os.environ.get("PMC_FILTER")
acc = collections.OrderedDict()
for path in sorted(glob.glob("gpurun_out/pmca_*_%s.csv" % tag)):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if (filt and not re.search(filt, k)) or (not filt and "attn_" not in k):
            continue
        mt = re.search(r"((?:%s)\w*(?:<[^>]*>)?)" % filt, k) if filt else re.search(r"(attn_\w+(?:<[^>]*>)?)", k)
        k = mt.group(1) if mt else k
        key = (k, r["Grid_Size"], r["Workgroup_Size"] + (" lds " + r["LDS_Block_Size"] if filt and r.get("LDS_Block_Size") else ""))
        per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        per[key]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for key, d in per.items():
        e = acc.setdefault(key, collections.OrderedDict())
        for c, v in d.items():
            v = v[1:] if len(v) > 1 else v          # drop the first (cold) launch
            if c == "dur_us" and c in e:
                continue
            e[c] = sum(v) / len(v)
print("# %s, PMC means per launch (%s)\n" % ("kernels matching " + filt if filt else "attention kernels", tag))
for key, e in acc.items():
    print("## %s  grid %s x wg %s" % key)
    wc = e.get("SQ_WAVE_CYCLES")
    for c, v in e.items():
        extra = ""
        if wc and c.startswith("SQ_") and c not in ("SQ_WAVE_CYCLES",) and ("WAIT" in c or "ACTIVE" in c):
            extra = "  (%.3f of SQ_WAVE_CYCLES)" % (v / wc)
        print("- %s: %.4g%s" % (c, v, extra))
    print()

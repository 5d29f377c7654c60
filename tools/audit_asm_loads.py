#!/usr/bin/env python
"""Audit hand-issued VMEM in a hipcc .s file.
(2) After a buffer_store_dwordx3/x4 the data registers must survive 2 wait states before a VALU overwrites them
    (hipcc exempts MUBUF stores with an SGPR soffset from this padding; gfx950 does not).
(1) Loads: between an inline-asm `buffer_load_dwordx4 v[a:b] ... offen`
(VGPR destination, i.e. not an LDS-DMA) and the next `s_waitcnt vmcnt`, no instruction may touch v[a:b]
(hipcc treats the destination as written at the asm statement and may copy / reuse it before the data lands:
cdna_hip_programming.md 5.7).  usage: audit_asm_loads.py file.s   -> exit 1 if a violation is found"""
import re
import sys

def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out

def main():
    bad = 0
    kern = None
    ops = []   # outstanding VMEM ops in issue order: (line, set of destination VGPRs)
    for n, line in enumerate(open(sys.argv[1]), 1):
        s = line.strip()
        if s.endswith(":") and s.startswith("_Z"):
            kern, ops = s[:70], []
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", s)
        if m:
            k = int(m.group(1))
            ops = ops[len(ops) - k:] if k > 0 and k < len(ops) else ([] if k == 0 else ops)
            continue
        # (2) store-data hazard
        if 'store_guard' not in globals():
            globals()['store_guard'] = []
        sg = globals()['store_guard']
        mnop = re.match(r"s_nop\s+(\d+)", s)
        step = int(mnop.group(1)) + 1 if mnop else 1
        if s.startswith("v_") and sg:
            dst = regs(s.split(",")[0])
            for (ln, d, left) in sg:
                if left > 0 and dst & d:
                    print("%s line %d overwrites data of the store at line %d too early: %s" % (kern, n, ln, s)); bad += 1
        sg[:] = [(ln, d, left - step) for (ln, d, left) in sg if left - step > 0]
        if re.match(r"buffer_store_dwordx[34]\b", s):
            sg.append((n, regs(s.split(",")[0]), 2))
        pend = {}
        for ln, d in ops:
            for r in d:
                pend[r] = ln
        touched = regs(s)
        if s.startswith(("buffer_load", "global_load", "buffer_store", "global_store", "scratch_")):
            dst = regs(s.split(",")[0]) if (s.startswith(("buffer_load", "global_load")) and " lds" not in s) else set()
            src = touched - dst
            for r in src:
                if r in pend:
                    print("%s line %d reads v%d (asm load at line %d not yet waited for): %s" % (kern, n, r, pend[r], s)); bad += 1
            ops.append((n, dst))
            continue
        for r in touched:
            if r in pend:
                print("%s line %d touches v%d (asm load at line %d not yet waited for): %s" % (kern, n, r, pend[r], s))
                bad += 1
    print("audit:", "FAILED (%d)" % bad if bad else "ok")
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python
"""A rocprofv3 --kernel-trace database of a whole bench.py line: split the dispatch timeline at idle gaps > 200 ms (model builds between
workloads), and for every segment print span, busy share, and the largest gaps with the kernels around them.
usage: trace_segments.py results.db [top=6]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scol = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
    name_col = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else scol[-1])
    rows = cur.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, disp, sym)).fetchall()
    short = lambda n: re.sub(r"\(anonymous namespace\)::|ezclip::|void ", "", n)[:44]
    segs, cur_seg, last_end = [], [], None
    for n, s, e in rows:
        if last_end is not None and s - last_end > 200e6:
            segs.append(cur_seg)
            cur_seg = []
        cur_seg.append((n, s, e))
        last_end = max(last_end or e, e)
    segs.append(cur_seg)
    for i, seg in enumerate(segs):
        if len(seg) < 500:
            continue
        busy, cs, ce, gaps = 0, None, None, []
        prev = None
        for n, s, e in seg:
            if cs is None:
                cs, ce = s, e
            elif s <= ce:
                ce = max(ce, e)
            else:
                busy += ce - cs
                gaps.append((s - ce, short(prev), short(n), (s - seg[0][1]) / 1e6))
                cs, ce = s, e
            prev = n
        busy += ce - cs
        span = ce - seg[0][1]
        names = {}
        for n, s, e in seg:
            names[short(n)] = names.get(short(n), 0) + 1
        kind = "autograd/train" if any("attn_bwd" in k for k in names) else "forward"
        gaps.sort(reverse=True)
        big = sum(g for g, *_ in gaps if g > 50e3)
        print("segment %d: %6d dispatches, span %8.1f ms, busy %.2f %%, idle %.1f ms (%.1f ms of it in gaps > 50 us) [%s]" % (
            i, len(seg), span / 1e6, 100.0 * busy / span, (span - busy) / 1e6, big / 1e6, kind))
        for g, a, b, at in gaps[:top]:
            print("      gap %8.1f us at +%8.1f ms   after %-44s before %s" % (g / 1e3, at, a, b))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""GPU busy time of a rocprofv3 --kernel-trace database: union of all kernel intervals over the span of the last `frac` of the trace
(the steady part), largest gaps between consecutive kernels, and the per-kernel totals of that window.
usage: trace_busy.py results.db [frac=0.5] [top=25]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scol = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
    name_col = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else scol[-1])
    rows = cur.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, disp, sym)).fetchall()
    t_lo, t_hi = rows[0][1], max(r[2] for r in rows)
    w0 = t_hi - (t_hi - t_lo) * frac
    win = [(n, s, e) for n, s, e in rows if s >= w0]
    busy, cur_s, cur_e, gaps = 0, None, None, []
    for n, s, e in win:
        if cur_s is None:
            cur_s, cur_e = s, e
        elif s <= cur_e:
            cur_e = max(cur_e, e)
        else:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, n))
            cur_s, cur_e = s, e
    busy += cur_e - cur_s
    span = max(e for _, _, e in win) - win[0][1]
    print("window %.1f ms, busy %.1f ms (%.2f %%), %d dispatches, idle %.2f ms in %d gaps" % (span / 1e6, busy / 1e6, 100.0 * busy / span, len(win), (span - busy) / 1e6, len(gaps)))
    gaps.sort(reverse=True)
    print("largest gaps (us, next kernel):", [(round(g / 1e3, 1), re.sub(r"\(anonymous namespace\)::|ezclip::", "", n)[:50]) for g, n in gaps[:8]])
    agg = {}
    for n, s, e in win:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        m = re.match(r"(?:void )?([\w:]+(?:<[^()]*>)?)", n)
        n = m.group(1) if m else n
        a = agg.setdefault(n, [0, 0])
        a[0] += 1; a[1] += e - s
    tot = sum(a[1] for a in agg.values())
    print("sum of kernel durations %.1f ms (streams overlap: %.2f x busy)" % (tot / 1e6, tot / busy))
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-110s %6d %9.3f ms %6.2f %%" % (n[:110], a[0], a[1] / 1e6, 100.0 * a[1] / tot))


if __name__ == "__main__":
    main()

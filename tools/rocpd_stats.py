#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace) as a per-kernel table
(the equivalent of --stats' kernel_stats.csv).  usage: rocpd_stats.py results.db [out.md] [group-regex]
With a group regex (e.g. "gemm_(nt|tn)(_8p)?_kernel": the launches bench.py's roofline object counts) a second table splits that group's
launches by position in the process timeline -- the first launches of a process run cold (first touch of the activation buffers, code
pages), and bench.py's `roofline.avg_launch_us` is measured on untimed steps AFTER the timed region: it has to be compared with the
steady part of the trace, not with the whole-process average."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % disp)]
    scol = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
    name_col = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else scol[-1])
    q = "select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id" % (name_col, disp, sym)
    rows = cur.execute(q).fetchall()
    agg = {}
    for name, st, en in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        m = re.match(r"(?:void )?([\w:]+(?:<[^()]*>)?)", name)
        name = m.group(1) if m else name
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        d = en - st
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.2f |" % (
            name[:110], a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / total))
    out = "\n".join(lines) + "\n\ntotal kernel time: %.3f ms over %d dispatches\n" % (total / 1e6, len(rows))
    if len(sys.argv) > 3:
        pat = re.compile(sys.argv[3])
        g = sorted((st, en - st) for name, st, en in rows if pat.search(name))
        if g:
            n = len(g)
            out += "\nlaunches matching `%s`: %d, whole-process average %.2f us (total %.3f ms); by position in the timeline:\n\n" % (
                sys.argv[3], n, sum(d for _, d in g) / n / 1e3, sum(d for _, d in g) / 1e6)
            out += "| launches | average us |\n|---|---|\n"
            for a, b in ((0.0, 0.125), (0.125, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0)):
                seg = g[int(a * n):int(b * n)]
                if seg:
                    out += "| %d .. %d | %.2f |\n" % (int(a * n), int(b * n) - 1, sum(d for _, d in seg) / len(seg) / 1e3)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()

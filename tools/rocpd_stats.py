#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into the per-kernel table that `--stats` prints:
calls, total/avg/min/max duration and share.  Usage: python tools/rocpd_stats.py <results.db> [out.md] [steps]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[:110]


def main():
    db, out = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else None
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    marks = [r[0] for r in con.execute(f"select start from kernels where {namecol} like '%g6d_marker_kernel%' order by start")]
    where = f"where start > {marks[0]} and start < {marks[-1]}" if len(marks) >= 2 else ""
    rows = con.execute(f"select {namecol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels {where} group by {namecol} order by 3 desc").fetchall()
    if len(marks) >= 2:
        print(f"region between g6d markers: {(marks[-1] - marks[0]) / 1e6:.3f} ms wall")
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, t, a, mn, mx in rows:
        lines.append(f"| {short(n)} | {c} | {t / 1e6:.3f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * t / total:.1f} |")
    lines.append(f"\ntotal kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches"
                 + (f" ({total / 1e6 / steps:.3f} ms per step over {steps:g} steps incl. warm-up)" if steps else ""))
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main()

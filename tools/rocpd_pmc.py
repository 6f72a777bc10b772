#!/usr/bin/env python
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database, restricted to the dispatches between the
g6d markers.  Usage: python tools/rocpd_pmc.py <results.db> <COUNTER> [name-filter]"""
import re
import sqlite3
import sys


def main():
    db, counter = sys.argv[1], sys.argv[2]
    filt = sys.argv[3] if len(sys.argv) > 3 else ""
    con = sqlite3.connect(db)
    marks = [r[0] for r in con.execute("select start from counters_collection where kernel_name like '%g6d_marker_kernel%' group by dispatch_id order by start")]
    where = f"and start > {marks[0]} and start < {marks[-1]}" if len(marks) >= 2 else ""
    # per dispatch: the counter value (summed over its rows) and the dispatch duration; the duration-weighted mean is the one to
    # read for a percentage metric such as MfmaUtil (a plain mean gives a 20 us launch the weight of a 400 us one)
    per = con.execute(f"select kernel_name, dispatch_id, sum(value), max(end) - min(start) from counters_collection "
                      f"where counter_name = ? {where} group by kernel_name, dispatch_id", (counter,)).fetchall()
    agg = {}
    for n, _, v, dur in per:
        a = agg.setdefault(n, [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += v; a[2] += v * max(dur, 0); a[3] += max(dur, 0)
    print(f"| kernel | dispatches | {counter} total | per dispatch | duration-weighted mean |\n|---|---|---|---|---|")
    for n, (c, v, vw, w) in sorted(agg.items(), key=lambda t: -t[1][1]):
        n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*", "", n)[:90]
        if filt in n:
            print(f"| {n} | {c} | {v:.4g} | {v / c:.4g} | {(vw / w if w > 0 else 0):.4g} |")


if __name__ == "__main__":
    main()

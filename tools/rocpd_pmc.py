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
    rows = con.execute(f"select kernel_name, count(distinct dispatch_id), sum(value) from counters_collection "
                       f"where counter_name = ? {where} group by kernel_name order by 3 desc", (counter,)).fetchall()
    print(f"| kernel | dispatches | {counter} total | per dispatch |\n|---|---|---|---|")
    for n, c, v in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*", "", n)[:90]
        if filt in n:
            print(f"| {n} | {c} | {v:.4g} | {v / c:.4g} |")


if __name__ == "__main__":
    main()

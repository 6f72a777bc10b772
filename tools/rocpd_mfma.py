#!/usr/bin/env python
"""MFMA-busy share per kernel from a rocprofv3 PMC pass with `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`
(rocprofv3's own derived metric: MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM) * 100,
SIMD_NUM = 256 CUs x 4), restricted to the dispatches between the g6d markers.
Usage: python tools/rocpd_mfma.py <results.db> [min-share-percent]"""
import re
import sqlite3
import sys

SIMD_NUM = 1024


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    marks = [r[0] for r in con.execute("select start from counters_collection where kernel_name like '%g6d_marker_kernel%' "
                                       "group by dispatch_id order by start")]
    where = f"and start > {marks[0]} and start < {marks[-1]}" if len(marks) >= 2 else ""
    busy, act = {}, {}
    for name, did, v in con.execute(f"select kernel_name, dispatch_id, sum(value) from counters_collection where "
                                    f"counter_name = 'SQ_VALU_MFMA_BUSY_CYCLES' {where} group by dispatch_id"):
        busy.setdefault(name, []).append(v)
    for name, did, v in con.execute(f"select kernel_name, dispatch_id, max(value) from counters_collection where "
                                    f"counter_name = 'GRBM_GUI_ACTIVE' {where} group by dispatch_id"):
        act.setdefault(name, []).append(v)
    rows = []
    for name in busy:
        b, a = sum(busy[name]), sum(act.get(name, [0]))
        if a > 0:
            rows.append((a, name, len(busy[name]), 100.0 * b / (a * SIMD_NUM)))
    tot = sum(r[0] for r in rows)
    print("| kernel | dispatches | share of GPU-active cycles % | MFMA busy % |\n|---|---|---|---|")
    for a, name, n, util in sorted(rows, reverse=True):
        n_ = re.sub(r"\(anonymous namespace\)::", "", name); n_ = re.sub(r"\(.*", "", n_)[:90]
        if util > 0.05:
            print(f"| {n_} | {n} | {100 * a / tot:.1f} | {util:.1f} |")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""HBM-side traffic of the g6d_conv_igemm kernel family from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, each
collected on its own with --kernel-trace only) of `bench.py --steps K --warmup W --no-cpu-baseline --no-graph`:
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 summed over the family's dispatches between the g6d markers, divided by the
number of g6d_conv_igemm / g6d_corr2d_patch launches (split launches write and re-read their partial tiles inside the kernel).
gfx950: FETCH_SIZE reports half the bytes of a wide coalesced stream (MI355X_MICROARCH.md, HBM section).
Usage: python tools/pmc_conv_traffic.py <fetch.db> <write.db> <steps> <out.json>"""
import json
import sqlite3
import sys

MAIN = ("conv_igemm_kernel", "conv_patch_kernel", "corr_patch_kernel")
FAMILY = MAIN                      # split launches finish inside the kernels: no reduce kernels any more
WMAIN = ("wino_conv3x3_kernel", "wino43_kernel")
WFAMILY = WMAIN


def family_sum(db, counter, FAMILY=FAMILY, MAIN=MAIN):
    con = sqlite3.connect(db)
    marks = [r[0] for r in con.execute("select start from counters_collection where kernel_name like '%g6d_marker_kernel%' "
                                       "group by dispatch_id order by start")]
    where = f"and start > {marks[0]} and start < {marks[-1]}" if len(marks) >= 2 else ""
    total, launches = 0.0, 0
    for name, n, v in con.execute(f"select kernel_name, count(distinct dispatch_id), sum(value) from counters_collection "
                                  f"where counter_name = ? {where} group by kernel_name", (counter,)):
        if any(f in name for f in FAMILY):
            total += v
        if any(f in name for f in MAIN):
            launches += n
    return total, launches


def main():
    fetch_db, write_db, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fetch_kb, launches = family_sum(fetch_db, "FETCH_SIZE")
    write_kb, _ = family_sum(write_db, "WRITE_SIZE")
    res = {
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps "
                  f"{steps} --warmup 2 --no-cpu-baseline --no-graph`, kernels conv_igemm / conv_patch / corr_patch "
                  "inside the timed region; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 "
                  "(gfx950: FETCH_SIZE counts half of a wide coalesced stream, MI355X_MICROARCH.md HBM section)",
        "fetch_kb_total": fetch_kb, "write_kb_total": write_kb, "steps": steps, "launches": launches,
        "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024 / max(launches, 1),
    }
    wf, wl = family_sum(fetch_db, "FETCH_SIZE", WFAMILY, WMAIN)
    ww, _ = family_sum(write_db, "WRITE_SIZE", WFAMILY, WMAIN)
    res["winograd_family"] = {"kernels": "wino_conv3x3_kernel + wino43_kernel (own trunks, the conv layers routed to them, the 15x15 correlation)",
                              "fetch_kb_total": wf, "write_kb_total": ww, "launches": wl,
                              "hbm_bytes_per_launch": (2 * wf + ww) * 1024 / max(wl, 1)}
    sf, sl = family_sum(fetch_db, "FETCH_SIZE", ("conv16", "corr16"), ("conv16", "corr16"))
    sw, _ = family_sum(write_db, "WRITE_SIZE", ("conv16", "corr16"), ("conv16", "corr16"))
    res["split16_family"] = {"kernels": "conv16w_kernel<3, *> + corr16_kernel<3> (trunks, correlations, selector stacks on fp16 hi / lo pairs)", "fetch_kb_total": sf, "write_kb_total": sw,
                             "launches": sl, "hbm_bytes_per_launch": (2 * sf + sw) * 1024 / max(sl, 1)}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

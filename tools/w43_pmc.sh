# PMC passes over tools/w43_probe.py (the F(4x4,3x3) kernel on four representative launches): separate rocprofv3 runs per counter set
R=$PWD; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/p_$name -o pmc -- python $R/tools/w43_probe.py > /dev/null 2>&1; }
dump() { python - "$@" <<'PY'
import sqlite3, sys, re
db = sys.argv[1]
con = sqlite3.connect(db)
rows = con.execute("select kernel_name, counter_name, dispatch_id, sum(value), max(end)-min(start) from counters_collection group by kernel_name, counter_name, dispatch_id").fetchall()
agg = {}
for n, c, d, v, dur in rows:
    if "wino43" not in n: continue
    agg.setdefault((d, c), [0, 0]); agg[(d, c)] = [v, dur]
disp = sorted({d for d, _ in agg})
cs = sorted({c for _, c in agg})
print("| dispatch | us | " + " | ".join(cs) + " |")
for d in disp[-4:]:
    print(f"| {d} | {agg[(d, cs[0])][1] / 1e3:.0f} | " + " | ".join(f"{agg[(d, c)][0]:.4g}" for c in cs) + " |")
PY
}
REPS=1 run f FETCH_SIZE; dump /tmp/p_f/pmc_results.db
REPS=1 run h TCC_HIT_sum TCC_MISS_sum; dump /tmp/p_h/pmc_results.db
REPS=1 run s SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS; dump /tmp/p_s/pmc_results.db
REPS=1 run l SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL; dump /tmp/p_l/pmc_results.db
REPS=1 run m MfmaUtil; dump /tmp/p_m/pmc_results.db

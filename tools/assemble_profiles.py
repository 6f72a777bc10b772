#!/usr/bin/env python
"""Turn the raw outputs of tools/profile_round.sh (gpurun_out/) into the committed summaries under profiles/."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
RND = sys.argv[1] if len(sys.argv) > 1 else "r01"
FAMILY = ("conv_igemm_kernel", "conv_patch_kernel", "corr_patch_kernel")      # split launches finish inside these kernels
WFAMILY = ("wino_conv3x3_kernel", "wino43_kernel")


def rd(f):
    return open(os.path.join(G, f)).read()


def bench_line(log):
    return json.loads([l for l in open(os.path.join(G, log)) if l.startswith('{"metric')][0])


def family_ms(stats_md, family=FAMILY):
    ms, main = 0.0, 0
    for l in stats_md.splitlines():
        c = [x.strip() for x in l.strip("|").split("|")]
        if len(c) >= 3 and any(k in c[0] for k in family):
            ms += float(c[2])
            if "splitk" not in c[0] and "reduce" not in c[0]:
                main += int(c[1])
    return ms, main


def main():
    fin, ser = bench_line("prof_final.log"), bench_line("prof_serial.log")
    steps = ser["steps"]
    open(os.path.join(P, f"{RND}_bench_kernel_stats.md"), "w").write(
        f"# rocprofv3 --kernel-trace --stats of `python bench.py --steps {steps} --warmup 2 --no-cpu-baseline --no-cached --lowp ''` (1x MI355X)\n\n"
        f"Default launch mode ({fin['config']['launch']}): a step = one batch of {fin.get('batch', 1)} queries.  Where the batches in\n"
        "flight overlap, per-kernel durations are inflated by sharing the chip (sum of durations > wall time of the region); the\n"
        "serialised run next to this file is the one to compare with the roofline.\n"
        "Produced by tools/profile_round.sh + tools/rocpd_stats.py from the rocpd database, cut to the timed region by the two\n"
        f"g6d_marker_kernel launches.  bench.py under the profiler: {fin['value']:.1f} images/s ({fin['ms_per_step']:.2f} ms per step of "
        f"{fin.get('batch', 1)} queries = {fin['ms_per_step'] / fin.get('batch', 1):.2f} ms per query).\n\n"
        + rd("prof_final_stats.md"))
    sstats = rd("prof_serial_stats.md")
    ms, main = family_ms(sstats)
    wms, wmain = family_ms(sstats, WFAMILY)
    fams = {ser["roofline"].get("family", "conv"): ser["roofline"]}
    for k in ("conv", "winograd", "split16"):
        if "roofline_" + k in ser:
            fams[k] = ser["roofline_" + k]
    r, w = fams["conv"], fams.get("winograd", {})
    s16 = fams.get("split16", {})
    s16ms, s16main = family_ms(sstats, ("conv16", "corr16"))
    gflop_step = r["gflop_per_launch"] * r["launches_per_step"]
    wg = w.get("gflop_direct_form_per_step", 0.0)
    wexec = w.get("gflop_per_launch", 0.0) * w.get("launches_per_step", 0.0)
    open(os.path.join(P, f"{RND}_bench_kernel_stats_serial.md"), "w").write(
        f"# rocprofv3 --kernel-trace --stats of `python bench.py --steps {steps} --warmup 2 --no-cpu-baseline --no-cached --lowp '' --serial` (1x MI355X)\n\n"
        f"Same workload, one BATCH of {ser.get('batch', 1)} queries at a time, no graph replay (`--serial`), which is how bench.py's roofline pass\n"
        "runs: per-kernel durations are not inflated by overlap and can be compared with the `roofline` objects of the bench JSON.\n"
        f"A step = {ser.get('batch', 1)} queries: divide the per-step figures by {ser.get('batch', 1)} for per-query numbers.\n\n"
        "* conv family (conv_igemm_kernel<...> + conv_patch_kernel<...> + corr_patch_kernel; split launches finish in-kernel) in the timed region:\n"
        f"  {ms:.2f} ms = {ms / steps:.2f} ms per step ({main / steps:.0f} launches per step, {gflop_step:.1f} GFLOP per step -> "
        f"{gflop_step / (ms / steps):.1f} TFLOP/s by kernel durations; bench.py's HIP-event figure in the same profiled run: "
        f"{r['conv_ms_per_step']:.2f} ms per step, {r['achieved']:.1f} TFLOP/s — the event brackets include launch gaps and the profiler's per-dispatch overhead).\n"
        "* Winograd family (wino_conv3x3_kernel<MODE,KD,NWN> = F(2x2,3x3) and wino43_kernel<MODE,KD,NT> = F(4x4,3x3): the own VGG trunks, the conv\n"
        "  layers routed to them and the detector's 15x15 correlation):\n"
        f"  {wms:.2f} ms = {wms / steps:.2f} ms per step ({wmain / steps:.0f} launches per step, {wg:.1f} GFLOP per step in DIRECT form = "
        f"{wexec:.1f} GFLOP executed in the Winograd domain (direct / 2.25 on F(2x2,3x3) launches, direct / 4 on F(4x4,3x3) launches) -> "
        f"{wexec / max(wms / steps, 1e-9):.1f} TFLOP/s executed, "
        f"{wg / max(wms / steps, 1e-9):.1f} TFLOP/s direct-form equivalent; bench.py: {w.get('ms_per_step', 0):.2f} ms per step, "
        f"{w.get('achieved', 0):.1f} TFLOP/s executed; by transform: {json.dumps({k: {kk: round(vv, 2) for kk, vv in v.items()} for k, v in w.get('by_transform', {}).items()})}).\n"
        "* split-precision family (conv16w_kernel<3, *> and corr16_kernel<3>: VGG trunks of detector pyramid and refiner crops, the detector's 15x15 / 7x7\n"
        "  correlations, the selector's product layers and InstanceNorm stacks as direct convolutions on fp16 hi / lo pairs, three v_mfma_f32_32x32x16_f16 per product):\n"
        f"  {s16ms:.2f} ms = {s16ms / steps:.2f} ms per step ({s16main / steps:.0f} launches per step, {s16.get('gflop_direct_form_per_step', 0):.1f} GFLOP per step in DIRECT form, x3 executed -> "
        f"{3 * s16.get('gflop_direct_form_per_step', 0) / max(s16ms / steps, 1e-9):.1f} TFLOP/s executed on the 16-bit matrix cores by kernel durations; bench.py: "
        f"{s16.get('ms_per_step', 0):.2f} ms per step, {s16.get('achieved', 0):.1f} TFLOP/s executed = {s16.get('frac', 0):.3f} of 2500).\n\n" + sstats)
    traffic = json.load(open(os.path.join(G, "pmc_conv_traffic.json")))
    open(os.path.join(P, f"{RND}_pmc_hbm.md"), "w").write(
        "# HBM traffic counters (rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, no other trace domains)\n\n"
        "`python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cached --lowp '' --no-graph`, dispatches inside the timed region (2 steps = 2 batches of the default size, 16 queries since round 4). Unit: KB.\n"
        "gfx950 note (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes of a wide coalesced stream — double it before comparing.\n\n"
        "Checks against algorithmic bytes (per dispatch, tables below):\n"
        "* `selector_levels_kernel` (score maps + product statistics of the three levels for 8 queries per launch: two launches per batch of 16): FETCH x2 vs\n"
        "  algorithmic 168 + 42 + 10.5 = 220.5 MB of reference cache, read ONCE per batch (+ 8 x 3.3 MB of fp64 R1/R2 sums, 8 x 0.7 MB of query rows).\n"
        "* `refiner_volume_kernel`: WRITE vs algorithmic (queries per launch) x 50.3 MB (mean|query 33.5 MB + std 16.8 MB per query).\n"
        f"* conv family (tools/pmc_conv_traffic.py -> {RND}_pmc_conv_traffic.json): {traffic['hbm_bytes_per_launch'] / 1e6:.1f} MB HBM-side per launch; "
        f"Winograd family: {traffic.get('winograd_family', {}).get('hbm_bytes_per_launch', 0) / 1e6:.1f} MB per launch — both far from HBM-bound.\n\n"
        "## FETCH_SIZE\n" + rd("pmc_fetch.md") + "\n## WRITE_SIZE\n" + rd("pmc_write.md"))
    open(os.path.join(P, f"{RND}_pmc_mfma.md"), "w").write(
        "# MFMA-busy counter (rocprofv3 --kernel-trace --pmc MfmaUtil, own pass)\n\n"
        "`python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cached --lowp '' --no-graph --serial` (default batches); `MfmaUtil` is rocprofv3's derived metric\n"
        "`sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM) * 100` per dispatch, i.e. chip-wide: a launch whose grid\n"
        "fills half the CUs cannot exceed 50.  `per dispatch` is the plain mean over the kernel's dispatches inside the timed region\n"
        "(small and large layers of one instantiation mixed), the last column weights every dispatch with its duration.\n"
        "Full grids (tools/wino_split_probe.py SIZES=big, 64 crops): the Winograd kernel runs at 110 TFLOP/s executed = 70 % of peak.\n\n"
        + rd("pmc_mfmautil.md"))
    shutil.copy(os.path.join(G, "pmc_conv_traffic.json"), os.path.join(P, f"{RND}_pmc_conv_traffic.json"))
    for src, dst in (("bench_final.json", "bench.json"), ("layer_table.md", "layer_table.md"), ("trunk_bench.md", "trunk_bench.md"),
                     ("layer_table_b8.md", "layer_table_batch8.md"), ("layer_table_b1.md", "layer_table_batch1.md"),
                     ("layer_table_fp16.md", "layer_table_fp16.md"), ("layer_table_b16.md", "layer_table_batch16.md"),
                     ("conv16_bench.md", "conv16_bench.md"), ("mfma16_peak.md", "mfma16_peak.md"), ("pmc_conv16_lds.md", "pmc_conv16_lds.md"),
                     ("conv16w_scaling.md", "conv16w_scaling.md"),
                     ("batch_sweep_all.txt", "batch_sweep.txt"), ("prof_b1_serial_stats.md", "kernel_stats_single_query.md"),
                     ("bench_gpus2.json", "bench_gpus2.json"), ("bench_gpus2_shard.json", "bench_gpus2_shard_refs.json"),
                     ("bench_chained.json", "bench_chained.json"), ("bench_shard_rccl_world1.json", "bench_shard_refs_rccl_world1.json"),
                     ("bench_force_dist.json", "bench_force_dist_rccl_world1.json")):
        if os.path.exists(os.path.join(G, src)):
            shutil.copy(os.path.join(G, src), os.path.join(P, f"{RND}_{dst}"))
    if os.path.exists(os.path.join(G, "convbench_direct.log")):
        open(os.path.join(P, f"{RND}_conv_microbench.md"), "w").write(
            "# g6d_conv_igemm micro-benchmark (tools/conv_bench.py, HIP events, 10 reps per shape, 1x MI355X)\n\n"
            "TFLOP/s are DIRECT-form FLOPs over time in both tables; a layer that runs on the Winograd kernel executes 1/2.25 of them.\n\n"
            "## direct kernels (conv_igemm / conv_patch / corr_patch)\n\n```\n" + rd("convbench_direct.log") + "```\n\n"
            "## with G6dConv.weight_wino set (eligible layers on the Winograd kernel; production routing rule applies)\n\n```\n"
            + (rd("convbench_wino.log") if os.path.exists(os.path.join(G, "convbench_wino.log")) else "") + "```\n")


if __name__ == "__main__":
    main()

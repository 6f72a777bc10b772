#!/usr/bin/env python
"""bench.py's `sweep` object (reference-view sweep of north_star: 32 / 64 / 128 selector views x 5 rotations through the full pipeline,
BASELINE configs[1] = selector only at 64 x 36) -> profiles/rNN_ref_sweep.md.  Usage: python tools/sweep_table.py <bench.json> <out.md>"""
import json
import sys


def main():
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric')][-1])
    sw = d["sweep"]
    lines = ["# Reference-view sweep (1x MI355X, fp32, `python bench.py`: the `sweep` object of the default line)", "",
             f"Full pipeline = detector 480x640 vs 32 refs (4 scales) + selector 128x128 crop vs N refs x 5 rotations + 3 refiner steps; batches of "
             f"{d['batch']} queries per launch, {d['config']['launch'].split('), ')[-1]} (hipGraph replay), synthetic weights.  `64 x 36` = BASELINE configs[1], the selector alone",
             "(one captured graph, one batch at a time).  Winograd fraction = FLOPs executed in the Winograd domain / HIP-event time of the family's",
             "launches in a serialised eager pass / 157.3 TFLOP/s.  Parity = selector logits against the reference's own module (tests/golden).", "",
             "| selector refs x rotations | workload | throughput | ms per batch | Winograd family TFLOP/s executed (fraction of fp32 MFMA peak) | all MFMA-family launches | logits vs reference (bar 1e-4) |",
             "|---|---|---|---|---|---|---|"]
    for key in ("32x5", "64x5", "128x5", "64x36_selector_only"):
        v = sw[key]
        r = v.get("roofline", {})
        w, a = r.get("winograd", {}), r.get("all_mfma", {})
        par = v.get("parity_vs_reference")
        ptxt = (f"{par['logits_max_abs_diff']:.1e}, arg-max {'equal' if par['argmax_equal'] else 'DIFFERS'} ({par['source'].split()[0]})" if par else
                ("timed rows vs pipeline_rows.npz: " + f"{d['parity_vs_reference']['max_rel_diff_row']:.1e} relative, arg-max equal" if key == "64x5" else "no fixture of this size"))
        ms = v.get("ms_per_step", d["ms_per_step"] if key == "64x5" else None)
        lines.append(f"| {key.replace('_selector_only', '').replace('x', ' x ')} | {'selector only' if 'selector_only' in key else 'full pipeline'} | "
                     f"{v['value']:.1f} {v['unit']} | {ms:.2f} | {w.get('achieved_TFLOPs_executed', 0):.1f} ({w.get('frac_of_fp32_mfma_peak', 0):.2f}) | "
                     f"{(str(round(a['achieved_TFLOPs_executed'], 1)) + ' (' + format(a['frac_of_fp32_mfma_peak'], '.2f') + ')') if a else '-'} | {ptxt} |")
    lines += ["", "Supersedes `profiles/r01_selector_sweep.md` (selector only, round-1 kernels, MIOpen trunk: 64 x 36 at 98.2 queries/s)."]
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()

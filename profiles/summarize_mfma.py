#!/usr/bin/env python3
"""Per-kernel MFMA-busy summary of one `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE`
pass (counter_collection.csv).  Per kernel: launches, average counters per launch and two ratios:
  mfma/sq_busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES              (the ratio SURVEY.md section 8d names)
  mfma_util    = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)   (fraction of all matrix pipes of
                 the chip that were busy while the kernel ran; rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs)
usage: summarize_mfma.py <counter_collection.csv> [...]"""
import collections
import csv
import statistics
import sys

KEYS = ["gemm_f32_split_bf16", "gemm_f32_mfma", "lstm_seq_persist_fwd", "lstm_step_fwd_fast", "lstm_step_bwd_fast", "lstm_step_fwd", "lstm_step_bwd",
        "attn_energy_co", "attn_energy", "attn_context", "attn_bwd_dw", "attn_bwd_ds", "linear_rows", "bn_", "adam", "sumsq", "colsum"]


def main(paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            name = next((k for k in KEYS if k in r["Kernel_Name"]), None)
            if name is None:
                continue
            if name.startswith("lstm_step"):
                name += f" grid={r['Grid_Size']}"
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"{'kernel':40s} {'launches':>8s} {'MFMA_BUSY/launch':>17s} {'SQ_BUSY/launch':>15s} {'GUI_ACTIVE/launch':>18s} {'mfma/sq_busy':>13s} {'mfma_util':>10s}")
    rows = []
    for name, d in agg.items():
        m, b, g = d.get("SQ_VALU_MFMA_BUSY_CYCLES", []), d.get("SQ_BUSY_CYCLES", []), d.get("GRBM_GUI_ACTIVE", [])
        if not m:
            continue
        mm, bb, gg = statistics.mean(m), (statistics.mean(b) if b else 0.0), (statistics.mean(g) if g else 0.0)
        util = mm / (gg / 8 * 256 * 4) if gg else float("nan")
        rows.append((sum(m), name, len(m), mm, bb, gg, mm / bb if bb else float("nan"), util))
    for _, name, n, mm, bb, gg, r1, util in sorted(rows, reverse=True):
        print(f"{name:40s} {n:8d} {mm:17.0f} {bb:15.0f} {gg:18.0f} {r1:13.3f} {util:10.3f}")


if __name__ == "__main__":
    main(sys.argv[1:])

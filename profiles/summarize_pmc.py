#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection.csv) per kernel: average counter value per launch, and the HBM
traffic of one teacher-forced decoder step (forward frame-loop kernels).  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950
FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) - the table prints raw and
corrected (x2) values.  usage: summarize_pmc.py <counter_collection.csv> [...]"""
import collections
import csv
import statistics
import sys

KEYS = ["lstm_step_fwd_fast", "lstm_step_bwd_fast", "attn_energy", "attn_context", "attn_bwd_dw", "attn_bwd_ds", "gemm_f32_mfma"]


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
    print(f"{'kernel':44s} {'counter':10s} {'launches':>8s} {'avg/launch':>14s}")
    per_step = 0.0
    for name, d in sorted(agg.items()):
        for c, v in sorted(d.items()):
            print(f"{name:44s} {c:10s} {len(v):8d} {statistics.mean(v):14.1f}")
            fwd = name.startswith("lstm_step_fwd_fast grid=65536") or name in ("attn_energy", "attn_context")
            if fwd and c in ("FETCH_SIZE", "WRITE_SIZE"):
                # the fwd LSTM kernel is launched twice per frame (attention cell + decoder cell): mean * 2
                mult = 2 if name.startswith("lstm_step_fwd_fast") else 1
                per_step += statistics.mean(v) * 1024 * mult * (2 if c == "FETCH_SIZE" else 1)
    if per_step:
        print(f"\nHBM-side traffic of the forward frame-loop kernels per decoder step (FETCH x2 corrected + WRITE): "
              f"{per_step / 1e6:.1f} MB   (algorithmic bytes/step 89.1 MB at B=32, L=188)")


if __name__ == "__main__":
    main(sys.argv[1:])

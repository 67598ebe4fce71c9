#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection.csv) per kernel: average counter value per launch, and the HBM
traffic of one teacher-forced decoder step (forward frame-loop kernels).  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950
FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) - the table prints raw and
corrected (x2) values.  usage: summarize_pmc.py [--steps=N --T=872] <counter_collection.csv> [...]"""
import collections
import csv
import statistics
import sys

KEYS = ["lstm_seq_persist_fwd", "lstm_step_fwd_fast", "lstm_step_bwd_fast", "attn_energy", "attn_context", "attn_bwd_dw", "attn_bwd_ds",
        "gemm_f32_split_bf16", "gemm_f32_mfma"]
# kernels of the teacher-forced forward frame loop (the BiLSTM uses lstm_step_fwd_fast with a different grid; the decoder-LSTM
# chain is the persistent launch, one per chunk of frames)
FWD_LOOP = ("lstm_step_fwd_fast grid=65536", "attn_energy", "attn_context", "lstm_seq_persist_fwd")


def main(argv):
    paths = [a for a in argv if not a.startswith("--")]
    opts = dict(a[2:].split("=") for a in argv if a.startswith("--"))
    n_steps, T = int(opts.get("steps", 2)), int(opts.get("T", 872))
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
    total = 0.0
    for name, d in sorted(agg.items()):
        for c, v in sorted(d.items()):
            print(f"{name:44s} {c:10s} {len(v):8d} {statistics.mean(v):14.1f}")
            if name in FWD_LOOP and c in ("FETCH_SIZE", "WRITE_SIZE"):
                total += sum(v) * 1024 * (2 if c == "FETCH_SIZE" else 1)
    if total:
        per_step = total / (n_steps * T)
        print(f"\nHBM-side traffic of the forward frame-loop kernels (attention cell, energies, context, persistent decoder-LSTM "
              f"chain) per decoder step, FETCH x2 corrected + WRITE, {n_steps} training steps x T={T}: "
              f"{per_step / 1e6:.1f} MB   (algorithmic bytes/step 89.1 MB at B=32, L=188)")


if __name__ == "__main__":
    main(sys.argv[1:])

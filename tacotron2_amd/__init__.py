"""MI355X-native Tacotron 2 hot path (see README.md / DESIGN.md)."""
import os as _os

# The HIP runtime maps the streams of a process onto a bounded number of hardware queues (GPU_MAX_HW_QUEUES, default 4, read when
# the runtime initialises); streams that share a queue do not run next to each other.  The engine needs its two streams (the
# latency-bound frame chains on one, GEMMs / the persistent decoder-LSTM launches on the other) on DIFFERENT queues - and a live
# RCCL communicator brings streams of its own: with the default of 4, `init_process_group("nccl", device_id=...)` alone (no
# collective issued) turns the 62.9 ms training step into 86.4 ms, every dependent launch of the chains 3-4 us slower; with 2, 8 or
# 16 queues the same process runs 62.9 ms (profiles/r04_rccl_hw_queues.txt).  16 leaves room for the loader thread's copy stream
# and whatever RCCL creates at 8 ranks.  Set before anything can have initialised the runtime: this package is imported before the
# first GPU call of every entry point (bench.py, main.py); a user's own setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

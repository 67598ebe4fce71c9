"""Data parallelism on the HIP path: two ranks sharing cuda:0 over gloo run Trainer.train_step on their
utterance shards; the result must equal the single-process HIP step on the concatenated batch (tests/dp_hip_check.py).
The 8-GPU RCCL run itself is the driver's; this covers everything but the transport."""
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.gpu_processes(3)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp_hip_check.py")] + extra
    return subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)


def test_two_rank_step_with_sync_bn_equals_single_process_step_on_the_whole_batch():
    r = _run([], 29621)
    assert r.returncode == 0 and "DP_CHECK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_two_rank_step_with_per_shard_bn_runs_and_keeps_replicas_identical():
    """Default mode (per-shard BatchNorm statistics, what a Lightning DDP run of the reference would do): gradients differ from
    the whole-batch step by construction, replicas must still end bit-identical."""
    r = _run(["--per-shard-bn"], 29622)
    assert r.returncode == 0 and "DP_CHECK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_two_rank_step_with_a_single_allreduce_call():
    """Trainer(overlap_allreduce=False): ONE all-reduce of the whole flat buffer after the backward (the default reduces it as two
    buckets and starts the larger one while the encoder backward runs; the tests above cover that path)."""
    r = _run(["--one-allreduce"], 29623)
    assert r.returncode == 0 and "DP_CHECK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_two_rank_step_at_the_judged_dims_and_batch_equals_the_single_process_step():
    """BASELINE configs[2]'s code path at configs[1]'s size: vanilla-lj-hifi dims, the bench batch (B = 32, L = 188, T = 872) sharded
    16 + 16 over two ranks, synchronised BatchNorm, ONE all-reduce of the 112.5 MB flat gradient buffer, clip + Adam with the 1/world
    scale - against the single-process HIP step on the whole batch (itself oracle-checked with every gradient,
    tests/test_gpu_judged_shapes.py): loss, every parameter gradient within 3e-4 of its scale, clip norm, BatchNorm statistics,
    bit-identical replicas.  (Two ranks share the one card over gloo; the 8-GPU RCCL run is the driver's.)"""
    r = _run(["--judged", "--one-allreduce"], 29624)
    assert r.returncode == 0 and "DP_CHECK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    print(r.stdout.strip().splitlines()[-1])

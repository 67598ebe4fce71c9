"""The RCCL (torch.distributed backend "nccl") code path of the data-parallel step, on the one GPU a builder has: a process
group of ONE rank (tests/rccl_world1_check.py).  Every other DP test of this tree runs over gloo; here the
async tail-bucket all-reduce under the side stream, the head all-reduce, Work.wait(), the synchronised BatchNorm reduces, the
barrier and destroy_process_group run on RCCL itself, and at world size 1 every reduce must be a bitwise identity."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    return subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_check.py")] + extra, cwd=ROOT,
                          capture_output=True, text=True, timeout=900, env=env)


@pytest.mark.parametrize("extra", [[], ["--one-allreduce"], ["--sync-bn"]], ids=["two_buckets", "one_call", "sync_bn"])
def test_dp_step_over_rccl_world1_is_the_single_process_step(extra):
    r = _run(extra, 29641 + len(extra) + (7 if "--sync-bn" in extra else 0))
    assert r.returncode == 0 and "RCCL_CHECK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    print(r.stdout.strip().splitlines()[-1])


def test_bench_force_dp_runs_the_rccl_path_at_one_gpu():
    """`bench.py --gpus 1 --force-dp`: the driver's multi-GPU command line minus the launcher - process group over RCCL, both
    gradient buckets, the `allreduce` segment in the breakdown - so that the 8-GPU run is not the first time RCCL sees this code."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29655")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dp", "--overlap-allreduce", "--steps", "2",
                        "--warmup", "1", "--batch", "4", "--no-cpu-baseline", "--no-decode", "--no-high"], cwd=ROOT, capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["allreduce"].startswith("2 buckets") and "allreduce" in d["segments_ms"]
    assert d["value"] > 0 and all(np.isfinite(d["loss"]))
    # the stream-concurrency self-check of Trainer ran in a process with a live RCCL communicator and passed: the chain of
    # dependent launches on the main stream was not held up by the wave idling on the side stream
    qc = d["config"]["queue_check"]
    assert qc["ok"] is True and qc["chain_beside_spin_us"] < qc["chain_alone_us"] + 0.5 * qc["spin_us"], qc
    assert qc["GPU_MAX_HW_QUEUES"] == "16" and d["per_rank"][0]["allreduce_ms"] > 0
    assert "predicted_unmeasured" in d["config"] and "expected_scaling" not in d["config"]


_QC = """
import json, os, sys
sys.path.insert(0, %r)
import tacotron2_amd, torch
import torch.distributed as dist
if %r:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29671", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
from tacotron2_amd.params import ParamStore
from tacotron2_amd.engine import Engine
from tests.helpers import SMALL
from oracle import tacotron2_ref as R
eng = Engine(ParamStore(R.default_dims(**SMALL), torch.device("cuda:0")))
first = eng.stream_concurrency_check()
print("QC " + json.dumps(dict(first=first, ensured=eng.ensure_concurrent_streams())))
if dist.is_initialized():
    dist.destroy_process_group()
"""


@pytest.mark.parametrize("queues,nccl,first_ok,ensured_ok", [("16", True, True, True), ("4", True, False, True), ("1", False, False, False)])
def test_stream_concurrency_check_sees_streams_that_share_a_hardware_queue(queues, nccl, first_ok, ensured_ok):
    """The probe behind Trainer.queue_check / bench.py's `config.queue_check` (profiles/r05_queue_check_probe.txt).  With ONE
    hardware queue for the whole process the engine's two streams serialise, the check says so and no new side stream can help.
    With the runtime's default of 4 queues and a live RCCL communicator - the condition that cost 37 % per step in round 4 - the
    first check fails too and Engine.ensure_concurrent_streams repairs it with a side stream on another queue.  With the package's
    default of 16 the first check passes."""
    r = subprocess.run([sys.executable, "-c", _QC % (ROOT, nccl)], cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, GPU_MAX_HW_QUEUES=queues))
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    qc = json.loads([l for l in r.stdout.splitlines() if l.startswith("QC ")][-1][3:])
    assert qc["first"]["ok"] is first_ok and qc["ensured"]["ok"] is ensured_ok, qc
    assert (qc["ensured"]["tries"] == 1) == first_ok
    if not first_ok:         # serialised: the chain ends at least the spin's time later than it does alone
        assert qc["first"]["chain_beside_spin_us"] > qc["first"]["chain_alone_us"] + 0.5 * qc["first"]["spin_us"]
    assert qc["first"]["GPU_MAX_HW_QUEUES"] == queues and 0.5 < qc["first"]["us_per_dependent_launch"] < 50

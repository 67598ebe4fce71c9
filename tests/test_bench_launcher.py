"""bench.py as its own launcher (`python bench.py --gpus N` with no WORLD_SIZE in the environment): CPU-only checks of the
launch plan and of the child supervision.  The reference has nothing here (run/train.py:235-243 is single-device); the contract is
the task's bench line for N GPUs of one node (BASELINE configs[2], [3])."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}


def test_dry_run_launch_prints_one_child_per_rank_and_touches_no_gpu():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "20", "--warmup", "5", "--dry-run-launch"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    plan = d["launch"]
    assert len(plan) == 8
    ports = {p["env"]["MASTER_PORT"] for p in plan}
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536
    for r_, p in enumerate(plan):
        e = p["env"]
        assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_ADDR"]) == (str(r_), str(r_), "8", "127.0.0.1")
        assert e["GPU_MAX_HW_QUEUES"] == "16" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        # the child IS the same command line (an ordinary rank: WORLD_SIZE is set for it), minus the dry-run flag
        assert p["cmd"][1] == BENCH and p["cmd"][2:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    # one card shared by all ranks (rehearsals): every rank on device 0
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--share-gpu", "--backend", "gloo", "--dry-run-launch"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300, env=_env())
    plan = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert [p["env"]["LOCAL_RANK"] for p in plan] == ["0", "0", "0"] and [p["env"]["RANK"] for p in plan] == ["0", "1", "2"]


def test_a_rank_count_that_disagrees_with_an_external_launcher_is_refused_with_a_message():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8"], cwd=ROOT, capture_output=True, text=True, timeout=300,
                       env=dict(_env(), WORLD_SIZE="2", RANK="0"))
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr


def test_launcher_relays_the_worst_return_code_and_leaves_no_child_behind():
    """No GPU in the build container: every rank fails its `torch.cuda.is_available()` assertion after the gloo rendezvous.  The
    launcher must come back promptly with a non-zero code (the driver's run then fails for a stated reason instead of hanging)."""
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--share-gpu", "--backend", "gloo", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600, env=_env())
    import torch
    if torch.cuda.is_available():          # (on a GPU box this is simply a two-rank rehearsal)
        assert r.returncode == 0, r.stderr[-3000:]
        return
    assert r.returncode != 0 and time.time() - t0 < 300
    assert "launcher: started 2 ranks" in r.stderr and "return codes per rank" in r.stderr
    assert "needs a GPU" in r.stderr
    # without --share-gpu the launcher itself refuses: fewer devices than ranks
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], cwd=ROOT, capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 2 and "shows 0 GPU" in r.stderr


def test_a_terminated_launcher_takes_its_ranks_with_it(tmp_path):
    """SIGTERM to the launcher (the driver's time limit) must end the child ranks too.  The children here are stand-ins that sleep:
    the launcher is pointed at them through T2_BENCH_CHILD_CMD (test hook), writes their PIDs to stderr, gets SIGTERM after they
    have started, and must come back within seconds with every child gone."""
    import re
    import signal
    child = tmp_path / "sleeper.py"
    child.write_text("import time, sys\nprint('up', flush=True)\ntime.sleep(600)\n")
    p = subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--share-gpu", "--backend", "gloo"], cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, env=dict(_env(), T2_BENCH_CHILD_CMD=f"{sys.executable} {child}"))
    line = ""
    t0 = time.time()
    while "started 2 ranks" not in line and time.time() - t0 < 120:
        line = p.stderr.readline()
    pids = [int(x) for x in re.search(r"pids \[([0-9, ]+)\]", line).group(1).split(",")]
    assert len(pids) == 2
    time.sleep(1.0)
    p.send_signal(signal.SIGTERM)
    rc = p.wait(timeout=60)
    assert rc != 0
    time.sleep(0.5)
    for pid in pids:
        try:
            os.kill(pid, 0)
            alive = True
        except ProcessLookupError:
            alive = False
        assert not alive, f"child {pid} survived its launcher"

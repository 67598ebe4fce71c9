"""Runs the named oracle jobs of tests/oracle_jobs.py as CPU child processes beside the GPU tests (test infrastructure).

start(names) - called by tests/conftest.py once the selected tests are known - queues one `python -m tests.oracle_jobs NAME OUT`
per name, WORKERS at a time, in the order the tests will ask for them.  oracle(name) returns the job's result, waiting for it if it
is still running; a name that was never started (a test run on its own) is computed in-process.  The children are plain CPU
processes: they never touch the GPU, and they are started from threads of this process with subprocess (fork + exec of a new
interpreter; nothing replaces the test process)."""
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The GPU box admits 6 processes with the device open at once.  A job is a CPU process, but torch's autograd engine enumerates the
# devices when its worker threads start and holds /dev/kfd open for ~0.3 s while it does (seen with HIP_VISIBLE_DEVICES empty too) -
# so a running job may count against that limit at any moment.  4 workers + the test process + one child process of a CLI test = 6;
# tests that start more GPU processes reserve their share first (`gpu_processes` marker -> reserve()).
WORKERS = int(os.environ.get("T2_ORACLE_WORKERS", "4"))
GPU_PROCESS_LIMIT = 6
THREADS = int(os.environ.get("T2_ORACLE_THREADS", "2"))
JOB_THREADS = {"judged_step": 8}        # the one job whose autograd graph is big enough to use them (B = 32, T = 872)

_pool = None
_dir = None
_futures = {}
_results = {}
waited_s = {}


_live = {}          # name -> Popen of a job that is running (ended by exact PID at shutdown)
_closing = False
import threading
_cv = threading.Condition()
_running = 0
_cap = WORKERS      # jobs allowed to run at once right now (lowered while a test with several GPU processes runs)


class reserve:
    """with reserve(n): ...  - a test that will have `n` processes on the GPU (itself included).  Lowers the number of oracle jobs
    that may run to GPU_PROCESS_LIMIT - n and waits until no more than that many are running."""

    def __init__(self, n: int):
        self.cap = max(0, GPU_PROCESS_LIMIT - int(n))

    def __enter__(self):
        global _cap
        with _cv:
            _cap = min(WORKERS, self.cap)
            while _running > _cap and not _closing:
                _cv.wait(timeout=1.0)
        return self

    def __exit__(self, *exc):
        global _cap
        with _cv:
            _cap = WORKERS
            _cv.notify_all()
        return False


def _run_child(name, out):
    global _running
    with _cv:
        while _running >= _cap and not _closing:
            _cv.wait(timeout=1.0)
        if _closing:
            raise RuntimeError("oracle pool is shutting down")
        _running += 1
    try:
        return _run_child_now(name, out)
    finally:
        with _cv:
            _running -= 1
            _cv.notify_all()


def _run_child_now(name, out):
    t0 = time.time()
    nthr = JOB_THREADS.get(name, THREADS)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(nthr))
    p = subprocess.Popen([sys.executable, "-m", "tests.oracle_jobs", name, out, str(nthr)], cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, env=env)
    _live[name] = p
    try:
        _, err = p.communicate()
    finally:
        _live.pop(name, None)
    if p.returncode != 0:
        raise RuntimeError(f"oracle job {name} failed (rc {p.returncode}):\n{err[-3000:]}")
    return out, time.time() - t0


def start(names):
    global _pool, _dir
    names = [n for n in dict.fromkeys(names) if n not in _futures]
    if not names:
        return
    if _pool is None:
        _pool = ThreadPoolExecutor(max_workers=WORKERS)
        _dir = tempfile.mkdtemp(prefix="t2_oracle_")
        # (the box's share is 16 cores per GPU, whatever nproc says: WORKERS x THREADS of them for the jobs, the rest for the oracle runs
        #  of the test process itself.  Measured: with torch's default of one thread per VISIBLE core - 256 on the box - the small
        #  oracle cases of the test process are slow enough to make the whole suite take 240 s instead of 150 s)
        cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
        torch.set_num_threads(max(2, min(torch.get_num_threads(), cores - WORKERS * THREADS)))
    for n in names:
        _futures[n] = _pool.submit(_run_child, n, os.path.join(_dir, n.replace(":", "_") + ".pt"))


def oracle(name):
    if name in _results:
        return _results[name]
    if name in _futures:
        t0 = time.time()
        out, took = _futures[name].result()
        waited_s[name] = (round(time.time() - t0, 1), round(took, 1))
        res = torch.load(out, map_location="cpu", weights_only=True)
        os.remove(out)
    else:
        from tests import oracle_jobs
        res = oracle_jobs.run(name)
    _results[name] = res
    return res


def release(name):
    """Drop a cached result (the judged-shape forwards hold ~100 MB each)."""
    _results.pop(name, None)


def shutdown():
    """End of the session (also after an early -x stop): queued jobs are cancelled, running ones are ended - by the exact process
    handles started here - so nothing is left computing behind a finished test run."""
    global _pool, _closing
    _closing = True
    with _cv:
        _cv.notify_all()
    if _pool is not None:
        for f in _futures.values():
            f.cancel()
        for p in list(_live.values()):
            if p.poll() is None:
                p.kill()
        _pool.shutdown(wait=True)
        _pool = None
    if _dir and os.path.isdir(_dir):
        for f in os.listdir(_dir):
            os.remove(os.path.join(_dir, f))
        os.rmdir(_dir)

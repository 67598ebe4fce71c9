import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_processes(n): the test has n processes on the GPU at once (itself included); more than 2 "
                                       "reserve their share of the box's limit of 6 from the background oracle jobs")
    config.addinivalue_line("markers", "oracle(name): the test reads the CPU-oracle job `name` of tests/oracle_jobs.py; the jobs of all "
                                       "selected tests are started as background CPU processes when the session starts "
                                       "('{L}' in the name is filled from the test's parameters)")


def pytest_collection_finish(session):
    """Start the CPU-oracle jobs of the SELECTED tests (after -m / -k deselection), in test order, beside the GPU tests.  Only on
    a machine with a GPU (device_count() does not initialise the runtime): without one the gpu tests are deselected or fail at
    their first line, and nothing should be left running."""
    names = []
    for item in session.items:
        for m in item.iter_markers("oracle"):
            params = getattr(getattr(item, "callspec", None), "params", {})
            names.append(m.args[0].format(**params))
    if not names:
        return
    import torch
    if torch.cuda.device_count() == 0 and os.environ.get("T2_ORACLE_POOL") != "1":
        return
    from tests import oracle_pool
    oracle_pool.start(names)


def pytest_sessionfinish(session, exitstatus):
    if "tests.oracle_pool" in sys.modules:
        pool = sys.modules["tests.oracle_pool"]
        if pool.waited_s:
            print("\n[oracle pool] job: (seconds the test waited, seconds the job took) " +
                  ", ".join(f"{k}: {v}" for k, v in pool.waited_s.items()))
        pool.shutdown()


@pytest.fixture(autouse=True)
def _gpu_process_share(request):
    m = request.node.get_closest_marker("gpu_processes")
    if m is None or "tests.oracle_pool" not in sys.modules:
        yield
        return
    with sys.modules["tests.oracle_pool"].reserve(int(m.args[0])):
        yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle runs beside the GPU tests.  A GPU box exposes every host thread (256) but grants a 16-core share: torch's default
    # intra-op thread count would oversubscribe it ~16x (the T = 3000 oracle runs took > 7 minutes instead of ~2).
    try:
        import torch
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    except Exception:
        pass


# Progress while the GPU suite runs (the long oracle comparisons print nothing for minutes, and pytest captures stderr): a line per
# minute with the running test's id, appended to gpurun_out/pytest_progress.log.
_current = {"id": None, "t0": 0.0}


def _heartbeat():
    import time
    out_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out")
    while True:
        time.sleep(60)
        if _current["id"]:
            try:
                os.makedirs(out_dir, exist_ok=True)
                with open(os.path.join(out_dir, "pytest_progress.log"), "a") as f:
                    f.write(f"{time.strftime('%H:%M:%S')} still running {_current['id']} ({time.time() - _current['t0']:.0f} s)\n")
            except OSError:
                pass


def pytest_sessionstart(session):
    if _has_gpu():
        import threading
        threading.Thread(target=_heartbeat, daemon=True).start()


def pytest_runtest_logstart(nodeid, location):
    import time
    _current["id"], _current["t0"] = nodeid, time.time()


def pytest_runtest_logfinish(nodeid, location):
    _current["id"] = None


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)

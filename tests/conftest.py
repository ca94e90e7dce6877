import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLD = os.path.join(ROOT, "tests", "golden")


_BEAT = {"done": 0, "now": "", "t0": 0.0, "stop": None}


def _heartbeat(stop, period, fd):
    # One line a minute on the real stdout, unbuffered: the GPU suite holds several tests that run for more than a minute each,
    # and behind a pipe pytest's dots are block-buffered -- a runner that takes seven silent minutes for a hang would kill a
    # healthy run.  (While a test runs, fd 1 itself is pytest's capture file; fd is the saved original.)
    import time
    while not stop.wait(period):
        try:
            os.write(fd, ("\n[sr tests] %.0f s, %d finished, in %s\n" % (time.time() - _BEAT["t0"], _BEAT["done"], _BEAT["now"])).encode())
        except OSError:
            return


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-second CPU test")


def pytest_sessionstart(session):
    config = session.config
    from stable_renderer_amd.hostcpu import limit_torch_threads
    limit_torch_threads()                                     # the oracle is torch-CPU: fit its pool to the cgroup quota (and the children's)
    period = float(os.environ.get("SR_TEST_HEARTBEAT_S", "60"))
    if period > 0 and _BEAT["stop"] is None:
        import threading
        import time
        fd = 1
        try:                                                  # fd capture: the original stdout is the capture's saved duplicate
            fd = config.pluginmanager.getplugin("capturemanager")._global_capturing.out.targetfd_save
        except Exception:
            pass
        _BEAT["t0"], _BEAT["stop"] = time.time(), threading.Event()
        threading.Thread(target=_heartbeat, args=(_BEAT["stop"], period, fd), daemon=True).start()


def pytest_runtest_logstart(nodeid, location):
    _BEAT["now"] = nodeid


def pytest_runtest_logfinish(nodeid, location):
    _BEAT["done"] += 1


def pytest_unconfigure(config):
    if _BEAT["stop"] is not None:
        _BEAT["stop"].set()


@pytest.fixture(scope="session")
def gold():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLD, name + ".npz"))
    return load

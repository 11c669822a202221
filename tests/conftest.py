import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def child_env(extra=None):
    """environment for a test subprocess that opens a HIP context of its own.  The persistent one-launch sweeps are reserved to the
    first context that takes the device's advisory lock; when THIS (pytest) process holds a context -- and with it, possibly, the lock,
    while launching nothing meanwhile -- the child is told to launch them regardless (HYP_PERSISTENT=1).  When the parent holds no
    context the child takes the lock the normal way, so that two pytest runs sharing a GPU exclude each other."""
    import os
    import sys
    env = dict(os.environ)
    if extra:
        env.update(extra)
    holds = any(getattr(m, "_ctx", None) is not None for n, m in list(sys.modules.items()) if n.endswith("_lib") and hasattr(m, "load_library"))
    if holds:
        env.setdefault("HYP_PERSISTENT", "1")
    return env

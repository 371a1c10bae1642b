import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The suites load libfcn8s_hip.so / libfcn8s_oracle.so from the tree; build them if a fresh checkout has not yet
    # (hipcc cross-compiles for gfx950 without a GPU; a no-op when the libraries exist).
    import glob
    import subprocess
    lib = os.path.join(ROOT, "fcn8s_tensorflow_amd", "libfcn8s_hip.so")
    orc = os.path.join(ROOT, "oracle", "libfcn8s_oracle.so")
    csrc = os.path.join(ROOT, "fcn8s_tensorflow_amd", "csrc")

    def stale(target, sources):
        return not os.path.exists(target) or any(os.path.getmtime(f) > os.path.getmtime(target) for f in sources if os.path.exists(f))

    jobs = []
    if stale(lib, glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(ROOT, "include", "fcn8s_hip.h")]):
        jobs.append(["make", "-C", csrc, "-j4"])
    if stale(orc, [os.path.join(ROOT, "oracle", "fcn8s_oracle.c")]):
        jobs.append(["make", "-C", os.path.join(ROOT, "oracle")])
    for cmd in jobs:                      # a missing or out-of-date library is rebuilt; a failed build fails the session loudly
        if os.path.exists("/opt/rocm/bin/hipcc") or "oracle" in cmd[2]:
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise pytest.UsageError("building %s failed:\n%s" % (cmd[2], (r.stdout + r.stderr)[-4000:]))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)

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
    lib = os.path.join(ROOT, "fcn8s_tensorflow_amd", "libfcn8s_hip.so")
    orc = os.path.join(ROOT, "oracle", "libfcn8s_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "fcn8s_tensorflow_amd", "csrc")], check=False, capture_output=True)
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=False, capture_output=True)


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

"""INTEGRATION.md section B shows the ctypes stub a maintainer of the reference would add next to fcn8s_tensorflow.py.  This test runs that
very text (extracted from the document, only the library path filled in): single-device session, then the data-parallel entry points with
the one RCCL rank this box has."""
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_namespace():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## B. Keep `fcn8s_tensorflow.py`"):]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    assert 'C.CDLL("libfcn8s_hip.so")' in code
    code = code.replace('C.CDLL("libfcn8s_hip.so")', 'C.CDLL(%r)' % os.path.join(ROOT, "fcn8s_tensorflow_amd", "libfcn8s_hip.so"))
    import torch  # noqa: F401  (one HIP runtime per process: torch's goes first, see fcn8s_tensorflow_amd/_lib.py)
    ns = {}
    exec(compile(code, "INTEGRATION.md#B", "exec"), ns)
    return ns


def test_the_documented_binding_runs():
    ns = _stub_namespace()
    Session, lib = ns["Session"], ns["lib"]
    s = Session(20)
    assert lib.fcn8s_init_params(s.h, 3) == 0
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (2, 64, 64, 3), dtype=np.uint8)
    onehot = np.eye(20, dtype=bool)[rng.integers(0, 20, (2, 64, 64))]
    loss, step = s.train_step(img, onehot, 1e-4, 0.5, 0.0)
    assert np.isfinite(loss) and step == 1
    s.reset_metrics(); s.eval_step(img, onehot, 0.0)
    mean_loss, miou, acc = s.metrics()
    assert np.isfinite(mean_loss) and 0.0 <= miou <= 1.0 and 0.0 <= acc <= 1.0
    pred = s.predict(img)
    assert pred.shape == (2, 64, 64) and pred.dtype == np.int64 and pred.min() >= 0 and pred.max() < 20
    sm = s.predict(img, argmax=False)
    assert sm.shape == (2, 64, 64, 20) and np.abs(sm.sum(-1) - 1).max() < 1e-5
    # data parallelism through the library's own RCCL rank (world 1 here: SUM over one rank, scale 1/1)
    s.join(Session.make_comm_id(), 0, 1)
    l2 = s.train_step_dp(img, onehot, 1e-4, 0.5, 0.0)
    assert np.isfinite(l2) and lib.fcn8s_global_step(s.h) == 2
    s.reset_metrics(); s.eval_step(img, onehot, 0.0)
    a = s.metrics(); b = s.metrics_dp()
    assert a == b
    s.close()

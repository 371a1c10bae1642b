"""examples/c_client.c is a plain C99 host of libfcn8s_hip.so, built by gcc alone (no hipcc, no HIP headers, no Python in the process):
the drop-in boundary is a C ABI (include/fcn8s_hip.h), and this is the proof that nothing behind it leans on PyTorch or the ctypes
layer.  CPU part: it builds, its --layout mode (pure host logic) agrees with the Python binding's view of the flat variable buffer, and
without a GPU it stops with the library's own message.  GPU part: the three hot sess.run sites of the reference, the split-phase step and
a state round trip through the C program, compared number by number with the same calls made through ctypes."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "c_client")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    assert os.path.exists(EXE)


def _kv(text):
    out = {}
    for line in text.splitlines():
        for tok in line.split():
            if "=" in tok:
                k, v = tok.split("=", 1)
                out.setdefault(k, []).append(v)
    return out


def test_c_client_builds_with_gcc_alone_and_knows_the_layout():
    _build()
    # the binary's dynamic dependencies: the library, libm / libc -- no Python, no torch
    needed = subprocess.run(["readelf", "-d", EXE], capture_output=True, text=True).stdout
    libs = [l.split("[")[1].split("]")[0] for l in needed.splitlines() if "(NEEDED)" in l]
    assert "libfcn8s_hip.so" in libs and not any("python" in l or "torch" in l for l in libs), libs
    out = subprocess.run([EXE, "--layout"], capture_output=True, text=True, check=True).stdout
    from fcn8s_tensorflow_amd.dp import layout
    specs, total, buckets = layout(20)
    lines = [l.split() for l in out.splitlines()]
    params = [l for l in lines if l[0] == "param"]
    assert len(params) == len(specs) == 42
    assert int(_kv(out)["param_floats"][0]) == total
    for l in params:
        name, off, shape = l[2], int(l[3].split("=")[1]), tuple(int(x) for x in l[4].split("=")[1].split("x"))
        assert tuple(specs[name][0]) == shape and int(specs[name][1]) == off, (name, specs[name], off, shape)
    got = [(int(l[2].split("=")[1]), int(l[3].split("=")[1])) for l in lines if l[0] == "bucket"]
    assert got == [(int(o), int(n)) for o, n in buckets]
    # the reference's variable names, in its order (fcn8s_tensorflow.py:331-350 for the ones it names)
    assert params[0][2] == "conv1_1/filter" and params[-1][2] == "fc7_pool4_pool3_conv2d_trans/bias"


def test_c_client_without_a_gpu_fails_with_the_librarys_message():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 2 and "fcn8s_create failed" in r.stderr and "no CPU fallback" in r.stderr, (r.returncode, r.stderr)


@pytest.mark.gpu
def test_c_client_runs_the_hot_path_and_matches_the_ctypes_binding():
    _build()
    N, H, W, steps = 2, 64, 96, 3
    r = subprocess.run([EXE, str(N), str(H), str(W), str(steps)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    kv = _kv(r.stdout)
    assert kv["c_client"] == ["ok"]
    c_losses = [float(x) for x in kv["loss"][:steps]]
    assert [int(x) for x in kv["train_step"]] == [1, 2, 3]
    assert int(kv["split_step"][0]) == steps + 1 and int(kv["buckets"][0]) == 4
    assert int(kv["resume_step"][0]) == steps + 2
    assert abs(float(kv["resume_loss"][0]) - float(kv["original_loss"][0])) <= 1e-6 * abs(float(kv["original_loss"][0]))
    # the same batch, generated the same way (Knuth's 64-bit LCG, top 31 bits), through the Python binding
    state = 42
    noise = np.empty(N * H * W * 3, np.uint8)
    for i in range(noise.size):
        state = (state * 6364136223846793005 + 1442695040888963407) & ((1 << 64) - 1)
        noise[i] = (state >> 33) & 31
    p = np.arange(N * H * W)
    x, y = p % W, (p // W) % H
    cls = ((x // 16) + 3 * (y // 16)) % 20
    img = ((cls[:, None] * 12 + 37 * np.arange(3)[None, :] + noise.reshape(-1, 3)) & 255).astype(np.uint8).reshape(N, H, W, 3)
    lab = cls.astype(np.uint8).reshape(N, H, W)
    from fcn8s_tensorflow_amd import _lib as L
    from fcn8s_tensorflow_amd.engine import Engine
    e = Engine(20, device_id=0, seed=1234)
    e.init_params(seed=7)
    py_losses = []
    for s in range(steps):
        loss, step = e.train_step(img, lab, 1e-4, keep_prob=0.5, l2_rate=1e-3, optimizer=L.OPT_TF_ADAM)
        assert step == s + 1
        py_losses.append(loss)
    # same library, same seeds, same bytes in: the same numbers out (up to the summation order of atomically reduced split launches)
    for a, b in zip(c_losses, py_losses):
        assert abs(a - b) <= 2e-5 * abs(b), (c_losses, py_losses)
    e.metrics_reset()
    e.eval_step(img, lab, l2_rate=1e-3)
    mloss, miou, acc = e.metrics_get()
    assert abs(float(kv["eval_loss"][0]) - mloss) <= 2e-5 * abs(mloss)
    assert abs(float(kv["eval_accuracy"][0]) - acc) <= 2e-3 and abs(float(kv["eval_mean_iou"][0]) - miou) <= 2e-2
    pred = e.predict(img)
    assert int(kv["predict_pixels"][0]) == pred.size
    assert abs(int(kv["predict_agree"][0]) - int((pred == lab).sum())) <= 0.002 * pred.size
    e.close()

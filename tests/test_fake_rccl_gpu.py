"""The library's own communicator (fcn8s_comm_* / fcn8s_allreduce_bucket / the watchdog, include/fcn8s_hip.h "data parallelism inside
the library") with MORE THAN ONE RANK on a one-GPU box: two processes share GPU 0 and exchange through tests/fake_rccl/librccl.so.1, a
shared-memory stand-in that keeps RCCL's contract where the code under test depends on it (asynchronous stream-ordered collectives that
block the stream until every peer arrived, ncclCommAbort releasing them, ncclCommGetAsyncError) and can make a rank stall or fail.
Everything on the library's side of the dlopen boundary is the product code, unchanged.  The same sequence over the real RCCL, one
rank per GPU, is tests/test_multigpu_gpu.py (gated on two GPUs).

Reference: the reference trains on one device (fcn8s_tensorflow.py:65); the exchange step is SURVEY 8e's."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_DIR = os.path.join(ROOT, "tests", "fake_rccl")
FAKE = os.path.join(FAKE_DIR, "librccl.so.1")
from oracle import fcn8s_oracle as orc  # noqa: E402  (checker only)

SMALL = (8, 16, 32, 64, 64, 128, 128)


@pytest.fixture(scope="module", autouse=True)
def fake_lib():
    if not os.path.exists(FAKE) or os.path.getmtime(FAKE) < os.path.getmtime(os.path.join(FAKE_DIR, "fake_rccl.c")):
        subprocess.check_call(["make", "-C", FAKE_DIR])
    return FAKE


def run_ranks(mode, tmp_path, world=2, env_all=None, env_rank=None, timeout=300):
    """Start `world` worker processes on GPU 0; returns their result dicts (a worker that does not end in `timeout` seconds fails the test:
    that is the hang the watchdog exists to prevent)."""
    idfile = str(tmp_path / "id.bin")
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env["FCN8S_RCCL_LIBRARY"] = FAKE
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(env_all or {})
        env.update((env_rank or {}).get(r, {}))
        out = str(tmp_path / ("out%d.json" % r))
        procs.append((subprocess.Popen([sys.executable, os.path.join(FAKE_DIR, "worker.py"), mode, str(r), str(world), idfile, out],
                                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), out))
    res = []
    t0 = time.time()
    try:
        for p, out in procs:
            text, _ = p.communicate(timeout=max(1.0, timeout - (time.time() - t0)))
            assert p.returncode == 0, text[-3000:]
            res.append(json.load(open(out)))
    finally:
        for p, _ in procs:
            if p.poll() is None:
                p.kill()
    return res


def test_two_native_ranks_step_equals_big_batch_step(tmp_path):
    """tests/test_multigpu_gpu.py::test_two_rank_rccl_step_equals_big_batch_step[native] with both ranks on one GPU: rank 1 receives rank 0's
    parameters through fcn8s_comm_broadcast_params, every bucket goes through fcn8s_allreduce_bucket while the backward pass runs, the
    replicas end bit-identical after three steps (two through the split-phase calls, one through plain fcn8s_train_step), the first
    update equals the update of one process stepping on the whole batch, and the evaluation metrics are summed over the ranks."""
    from fcn8s_tensorflow_amd.engine import Engine
    from fcn8s_tensorflow_amd import _lib as L
    from tests.test_facade_gpu import gen
    res = run_ranks("step", tmp_path)
    z0, z1 = np.load(str(tmp_path / "out0.json.npz")), np.load(str(tmp_path / "out1.json.npz"))
    np.testing.assert_array_equal(z0["params"], z1["params"])
    np.testing.assert_array_equal(z0["params2"], z1["params2"])
    assert not np.array_equal(z0["params"], z0["params2"])                   # (the fcn8s_train_step step did update)
    p0 = z0["params2"]
    for r in res:
        assert r["step"] == 1 and r["step2"] == 2 and r["step3"] == 3 and np.isfinite(r["loss"]) and np.isfinite(r["loss2"])
        assert r["conf_sum"] == 4 * 32 * 64 and r["loss_count"] == 2
    # the two SGD updates by ONE process on the whole batch (TF-Adam's update is +-lr whatever the gradient's size: not a fair comparison under re-ordered sums)
    P = orc.init_params(20, SMALL, seed=1, decoder_std_scale=30.0, bias_std=0.05)
    img, lab = next(gen(4, 32, 64, 4, onehot=False))
    e = Engine(20, widths=SMALL, seed=7); e.set_params(P)
    before = e.flat_params.cpu().numpy().copy()
    e.train_step(img, lab, 1e-2, keep_prob=1.0, l2_rate=1e-3, optimizer=L.OPT_SGD_MOMENTUM)
    e.train_step(img, lab, 1e-2, keep_prob=1.0, l2_rate=1e-3, optimizer=L.OPT_SGD_MOMENTUM)
    ref = e.flat_params.cpu().numpy()
    e.close()
    upd_ref, upd_dp = ref - before, p0 - before
    assert np.abs(upd_ref).max() > 0
    assert np.abs(upd_dp - upd_ref).max() <= 2e-3 * np.abs(upd_ref).max(), np.abs(upd_dp - upd_ref).max() / np.abs(upd_ref).max()


def test_four_native_ranks_stay_bit_identical(tmp_path):
    """Four ranks of the native communicator on the one GPU (the sums of four buffers in rank order: every replica gets the same bits), three steps each,
    the evaluation metrics summed over all four."""
    res = run_ranks("step", tmp_path, world=4, timeout=400)
    z = [np.load(str(tmp_path / ("out%d.json.npz" % r))) for r in range(4)]
    for r in range(1, 4):
        np.testing.assert_array_equal(z[0]["params"], z[r]["params"])
        np.testing.assert_array_equal(z[0]["params2"], z[r]["params2"])
    for r in res:
        assert r["step3"] == 3 and r["conf_sum"] == 8 * 32 * 64 and r["loss_count"] == 4, r


def test_bench_eight_native_ranks_on_one_gpu(tmp_path):
    """The driver's N = 8 command with `--comm native`, all eight ranks on GPU 0 over the stand-in: BASELINE config 4's rank count through the library's own
    communicator -- eight per-rank times, the 538 MB of gradients of the full-width network summed over eight ranks in four buckets, replicas bit-identical."""
    env = dict(os.environ); env["FCN8S_RCCL_LIBRARY"] = FAKE; env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--device", "0", "--comm", "native", "--steps", "2",
                        "--warmup", "1", "--repeats", "1", "--batch", "1", "--height", "64", "--width", "64", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out.get("error") is None and out["n_gpus"] == 8 and out["config"]["global_batch"] == 8 and len(out["per_rank_ms"]) == 8, out
    c = out["comm"]
    assert c["collectives_by"].startswith("libfcn8s_hip") and c["ranks"] == 8 and c["replicas_identical_after_timed_steps"] is True, c
    assert len(c["bucket_mb"]) == 4 and abs(sum(c["bucket_mb"]) - 537.9) < 1.0


def test_bench_and_run_dp_over_the_native_communicator_two_ranks_on_one_gpu(tmp_path):
    """`bench.py --gpus 2 --comm native` and `run_dp.py --gpus 2 --comm native` as tests/test_multigpu_gpu.py runs them on a two-GPU box, here with both ranks
    on GPU 0: the torch group (gloo) only carries the 128-byte id, every gradient bucket goes through fcn8s_allreduce_bucket of the library's own
    communicator (over the stand-in), the replicas stay bit-identical, four buckets leave in backward-production order."""
    env = dict(os.environ); env["FCN8S_RCCL_LIBRARY"] = FAKE; env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--device", "0", "--comm", "native", "--steps", "3",
                        "--warmup", "1", "--repeats", "1", "--batch", "2", "--height", "64", "--width", "64", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out.get("error") is None and out["n_gpus"] == 2 and out["config"]["global_batch"] == 4, out
    c = out["comm"]
    assert c["collectives_by"].startswith("libfcn8s_hip") and c["ranks"] == 2 and c["replicas_identical_after_timed_steps"] is True, c
    assert len(c["bucket_mb"]) == 4 and c["bucket_issue_ms"] == sorted(c["bucket_issue_ms"]), c
    assert out["value"] > 0 and np.isfinite(out["final_loss"])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run_dp.py"), "--gpus", "2", "--backend", "gloo", "--device", "0", "--comm", "native", "--batch", "2",
                        "--height", "64", "--width", "64", "--steps-per-epoch", "3", "--workers", "0"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "rank 0: 2 ranks x 2 images/step, global step 3" in r.stdout, r.stdout[-1000:]


def test_a_stalled_peer_fails_the_update_within_the_timeout(tmp_path):
    """Rank 1 never arrives at the all-reduce.  Both ranks' fcn8s_apply_update must return FCN8S_ERR_RCCL about comm_timeout_ms (1.5 s) after
    the buckets were queued -- not hang, and not apply an update to gradients that were never exchanged; every later collective call says
    the same; fcn8s_comm_destroy reports the failure once; afterwards the model trains alone."""
    res = run_ranks("stall_update", tmp_path, env_all={"FAKE_RCCL_STALL_RANK": "1", "TEST_COMM_TIMEOUT_MS": "1500"}, timeout=240)
    for r in res:
        assert r["raised"] and "fcn8s_apply_update" in r["msg"] and "comm_timeout_ms" in r["msg"] and "aborted" in r["msg"], r
        assert 1.0 <= r["elapsed_s"] <= 30.0, r
        assert r["step_after"] == 0 and r["params_untouched"], r
        assert r["second_raised"] and r["metrics_raised"] and r["destroy_raised"] and r["alone_loss_finite"], r


def test_an_asynchronous_rccl_error_fails_the_update_at_once(tmp_path):
    """ncclCommGetAsyncError reports ncclRemoteError on rank 0 (timeout 600 s: only the asynchronous-error poll can end its wait);
    rank 1, whose peer is gone, runs into its own 3 s timeout."""
    res = run_ranks("async_error", tmp_path, env_all={"FAKE_RCCL_ASYNC_ERROR_RANK": "0"},
                    env_rank={0: {"TEST_COMM_TIMEOUT_MS": "600000"}, 1: {"TEST_COMM_TIMEOUT_MS": "3000"}}, timeout=240)
    r0, r1 = res
    assert r0["raised"] and "asynchronous RCCL error" in r0["msg"] and "remote process exited" in r0["msg"], r0
    assert r0["elapsed_s"] < 20.0 and r0["params_untouched"] and r0["step_after"] == 0, r0
    assert r1["raised"] and "comm_timeout_ms" in r1["msg"] and r1["params_untouched"], r1


def test_destroy_with_collectives_against_a_dead_peer_returns(tmp_path):
    """fcn8s_destroy straight after queueing the all-reduces, the peer never arriving: returns FCN8S_ERR_RCCL after about the timeout."""
    res = run_ranks("stall_destroy", tmp_path, env_all={"FAKE_RCCL_STALL_RANK": "1", "TEST_COMM_TIMEOUT_MS": "1500"}, timeout=240)
    for r in res:
        assert r["rc"] == r["err_rccl"] and "aborted" in r["msg"], r
        assert 1.0 <= r["elapsed_s"] <= 30.0, r


def test_the_metrics_allreduce_is_watched_too(tmp_path):
    """Rank 1 never calls fcn8s_comm_allreduce_metrics: rank 0's call (which needs the sums on the host) ends with FCN8S_ERR_RCCL after the
    timeout instead of synchronising the stream for ever."""
    res = run_ranks("metrics_stall", tmp_path, env_all={"TEST_COMM_TIMEOUT_MS": "1500"}, timeout=240)
    r0 = res[0]
    assert r0["raised"] and "metrics all-reduce" in r0["msg"] and 1.0 <= r0["elapsed_s"] <= 30.0, r0


def test_a_librccl_of_another_major_version_is_refused(tmp_path):
    res = run_ranks("version", tmp_path, world=1, env_all={"FAKE_RCCL_VERSION": "30100"})
    assert res[0]["raised"] and "2.x" in res[0]["msg"], res

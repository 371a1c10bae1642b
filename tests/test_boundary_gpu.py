"""The boundary around the hot path on the GPU: asynchronous host staging, the frozen-parameter guard, bench.py as the
driver calls it (self-launching ranks, end-to-end mode), and the LDS-DMA GEMM kernels at every tile configuration."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fcn8s_oracle as orc  # noqa: E402  (checker only)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = (8, 16, 32, 64, 64, 128, 128)


def _engine(widths=SMALL, seed=0):
    from fcn8s_tensorflow_amd.engine import Engine
    return Engine(20, widths=widths, device_id=0, seed=seed)


def _batch(n, h, w, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8), rng.integers(0, 20, (n, h, w), dtype=np.uint8)


def test_staged_inputs_equal_direct_inputs():
    """fcn8s_stage_inputs / stage_wait / stage_release: a batch that travels through a pinned staging slot and the copy stream
    gives the results of the same batch passed as host arrays (bit-identical on the deterministic forward path; to round-off
    for the gradients, whose weight-gradient atomics commute only up to fp32 summation order), slots can be refilled
    while earlier steps are queued, and staging from another thread works (what the facade's feeder does)."""
    import threading
    P = orc.init_params(20, SMALL, seed=1, decoder_std_scale=30.0, bias_std=0.05)
    a, b = _engine(), _engine()
    a.set_params(P); b.set_params(P)
    batches = [_batch(2, 32, 64, s) for s in range(5)]
    staged = [None] * len(batches)

    def feeder():
        for i, (img, lab) in enumerate(batches[:3]):
            staged[i] = b.stage(img, lab, slot=i % 3)
    t = threading.Thread(target=feeder); t.start(); t.join()
    for i, (img, lab) in enumerate(batches):
        if staged[i] is None:
            staged[i] = b.stage(img, lab, slot=i % 3)          # refills a slot whose previous batch has been consumed
        la = a.forward_backward(img, lab, keep_prob=1.0)
        lb = b.forward_backward(staged[i], None, keep_prob=1.0)
        assert la == lb, (i, la, lb)                           # same weights, same batch: the forward pass is deterministic
        ga, gb = a.flat_grads, b.flat_grads
        assert float((ga - gb).abs().max()) <= 1e-4 * float(ga.abs().max())      # (weight-gradient atomics: summation order only)
    la, sa = a.train_step(batches[0][0], batches[0][1], 1e-3, keep_prob=1.0)
    lb, sb = b.train_step(b.stage(batches[0][0], batches[0][1], slot=2), None, 1e-3, keep_prob=1.0)
    assert la == lb and sa == sb == 1
    b.flat_params.copy_(a.flat_params)
    img, lab = batches[0]
    st = b.stage(img, None, slot=0)                              # images only: predict
    np.testing.assert_array_equal(a.predict(img), b.predict(st).cpu().numpy())
    a.metrics_reset(); b.metrics_reset()
    a.eval_step(img, lab); b.eval_step(b.stage(img, lab, slot=1), None)
    np.testing.assert_array_equal(a.metrics_raw()[0], b.metrics_raw()[0])
    with pytest.raises(Exception):
        b.stage(img, lab[:, :16], slot=0)
    a.close(); b.close()


def test_frozen_parameters_survive_a_write_through_the_side_door():
    """A torch write into the external parameter buffer while frozen (what a torch optimizer over views would do) is caught by the
    fingerprint guard: the next forward pass rebuilds the cached Winograd filter banks instead of using stale ones."""
    P = orc.init_params(20, SMALL, seed=3, decoder_std_scale=30.0, bias_std=0.05)
    img, _ = _batch(1, 128, 160, 0)
    e = _engine(); e.set_params(P)
    e.freeze(True)
    before = e.predict(img, argmax=False)
    np.testing.assert_array_equal(e.predict(img, argmax=False), before)      # cached banks in use
    e.flat_params.mul_(1.01)                                                 # the side door: no library call, no freeze(False)
    after = e.predict(img, argmax=False)
    e2 = _engine(); e2.flat_params.copy_(e.flat_params)
    np.testing.assert_array_equal(after, e2.predict(img, argmax=False))      # = a model that never cached anything
    assert np.abs(after - before).max() > 0
    e.close(); e2.close()


def _run_bench(*args, timeout=600):
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` without torchrun, exactly as the driver calls it for N > 1 (the ranks share this box's one GPU and
    exchange over gloo; on the 8-GPU node the same path runs one rank per GPU over RCCL)."""
    out = _run_bench("--gpus", "2", "--backend", "gloo", "--device", "0", "--steps", "2", "--warmup", "1", "--batch", "2",
                     "--height", "64", "--width", "64", "--no-cpu-baseline")
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    c = out["comm"]
    assert c["backend"] == "gloo" and c["rccl_ranks"] == 0 and out["rccl_ranks"] == 0 and c["ranks"] == 2       # gloo ranks are not RCCL ranks
    assert len(c["allreduce_ms_per_bucket_standalone"]) == 4 and len(out["per_rank_ms"]) == 2
    assert len(c["bucket_issue_ms"]) == 4 and all(d >= i for i, d in zip(c["bucket_issue_ms"], c["bucket_complete_ms"]))
    assert c["bucket_issue_ms"] == sorted(c["bucket_issue_ms"])                        # buckets leave in backward-production order
    assert out["value"] > 0 and np.isfinite(out["final_loss"])
    assert len(out["timed_regions_ms_per_step"]) == 3 and out["timed_regions"]["min_ms_per_step"] <= out["ms_per_step"] <= out["timed_regions"]["max_ms_per_step"]


def test_bench_eight_ranks_dry_run_on_one_gpu():
    """The first 8-GPU run must not be the first time eight ranks meet: `python bench.py --gpus 8` exactly as the driver calls it, the eight
    ranks sharing this box's one GPU over gloo.  Eight per-rank times, global batch 8 x 2, all four buckets traced in backward-production
    order, every rank bound (or knowingly not) to its GPU's NUMA node, and the replicas still bit-identical after the timed steps."""
    out = _run_bench("--gpus", "8", "--backend", "gloo", "--device", "0", "--steps", "2", "--warmup", "1", "--repeats", "1", "--batch", "2",
                     "--height", "64", "--width", "64", "--no-cpu-baseline", timeout=1500)
    assert out["n_gpus"] == 8 and out["config"]["global_batch"] == 16 and out["config"]["parallelism"] == "dp8"
    assert len(out["per_rank_ms"]) == 8 and all(t > 0 for t in out["per_rank_ms"])
    c = out["comm"]
    assert c["ranks"] == 8 and c["rccl_ranks"] == 0 and c["replicas_identical_after_timed_steps"] is True
    assert len(c["bucket_mb"]) == len(c["bucket_issue_ms"]) == len(c["bucket_complete_ms"]) == len(c["allreduce_ms_per_bucket_standalone"]) == 4
    assert abs(sum(c["bucket_mb"]) - 537.9) < 1.0 and c["bucket_mb"][1] > 400                      # fc6 alone
    assert c["bucket_issue_ms"] == sorted(c["bucket_issue_ms"]) and all(d >= i for i, d in zip(c["bucket_issue_ms"], c["bucket_complete_ms"]))
    assert c["exposed_comm_ms_per_step"] is not None and c["local_only_ms_per_step"] > 0 and isinstance(c["numa"], dict)
    assert out["value"] > 0 and np.isfinite(out["final_loss"])


def test_bench_measures_roofline_traffic_live():
    """The plain one-GPU training run re-measures roofline.traffic itself (two rocprofv3 --pmc passes of its own command after the timed
    regions); here forced on a small case: the line must say the figure is live and carry both counters of the dominant kernel."""
    import shutil
    if not shutil.which("rocprofv3"):
        pytest.skip("rocprofv3 not on PATH")
    out = _run_bench("--steps", "2", "--warmup", "1", "--batch", "2", "--height", "128", "--width", "128", "--no-cpu-baseline", "--live-traffic", timeout=900)
    r = out["roofline"]
    d = r["traffic_detail"]
    assert d and "error" not in d, d
    assert d["traffic_source"].startswith("live"), d
    assert r["traffic"] == round(d["hbm_mb_per_launch"] * 1e6) and d["fetch_mb"] > 0 and d["write_mb"] > 0
    quiet = _run_bench("--steps", "2", "--warmup", "1", "--batch", "2", "--height", "128", "--width", "128", "--no-cpu-baseline")
    src = (quiet["roofline"].get("traffic_detail") or {}).get("traffic_source", "committed")
    assert not src.startswith("live")               # side benches stay quick: live only on request or in the default headline run


def test_bench_under_torchrun_one_rank_rccl():
    """bench.py launched the way the driver launches it for N > 1 -- `python -m torch.distributed.run ... bench.py --gpus N` -- with the one
    rank this box has: the process group is RCCL (backend nccl), the four bucketed all-reduces run through it inside every step, and the
    line says so (rccl_ranks == 1).  More ranks than GPUs cannot run over RCCL here; the 2-rank path is covered over gloo above."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--height", "64", "--width", "64", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c = out["comm"]
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and c["backend"] == "nccl" and c["rccl_ranks"] == 1
    assert len(c["bucket_issue_ms"]) == 4 and all(d >= i for i, d in zip(c["bucket_issue_ms"], c["bucket_complete_ms"]))
    assert c["rccl_version"] and "nccl_env" in c and len(c["allreduce_gbs_per_bucket_standalone"]) == 4
    assert all(np.isfinite(x) and x >= 0 for x in c["allreduce_ms_per_bucket_standalone"])
    assert c["overlap_frac"] is None or 0.0 <= c["overlap_frac"] <= 1.0
    assert np.isfinite(out["final_loss"]) and out["value"] > 0


def test_run_dp_launches_two_ranks_and_trains():
    """run_dp.py (the launcher INTEGRATION.md names): self-launches 2 ranks, each runs FCN8s.train() on its shard of a generated PNG
    tree, gradients exchanged over gloo on this one-GPU box (RCCL on the 8-GPU node)."""
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run_dp.py"), "--gpus", "2", "--backend", "gloo", "--device", "0", "--batch", "2",
                        "--height", "64", "--width", "64", "--steps-per-epoch", "2", "--workers", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "rank 0: 2 ranks x 2 images/step, global step 2" in r.stdout, r.stdout[-1000:]


def test_run_dp_eight_ranks_dry_run():
    """run_dp.py with eight ranks on the one GPU (gloo): FCN8s.train() on eight file shards, evaluation all-reduced, replica guard on."""
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run_dp.py"), "--gpus", "8", "--backend", "gloo", "--device", "0", "--batch", "1",
                        "--height", "64", "--width", "64", "--steps-per-epoch", "2", "--workers", "0"], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "rank 0: 8 ranks x 1 images/step, global step 2" in r.stdout, r.stdout[-1000:]


def test_run_dp_trains_through_the_native_communicator():
    """run_dp.py --comm native with the one rank this box has: FCN8s.train() end to end, every gradient bucket all-reduced by the library's own RCCL
    communicator (no torch.distributed process group exists in this run at all), evaluation metrics summed through fcn8s_comm_allreduce_metrics."""
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run_dp.py"), "--gpus", "1", "--comm", "native", "--batch", "2", "--height", "64", "--width", "64",
                        "--steps-per-epoch", "3", "--workers", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "rank 0: 1 ranks x 2 images/step, global step 3" in r.stdout, r.stdout[-1000:]


def test_bench_under_torchrun_one_rank_native_rccl():
    """`--comm native`: the gradient buckets are all-reduced by the library's own RCCL communicator (fcn8s_comm_init /
    fcn8s_allreduce_bucket behind the C ABI), torch.distributed only carried the 128-byte id.  One rank is all this box has."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--height", "64", "--width", "64", "--no-cpu-baseline", "--comm", "native"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c = out["comm"]
    assert c["collectives_by"].startswith("libfcn8s_hip") and c["rccl_version"]
    assert np.isfinite(out["final_loss"]) and out["value"] > 0 and out["backward_check"]["ok"]


def test_native_rccl_communicator_one_rank():
    """fcn8s_comm_* through the C ABI with the one rank this box has (ncclCommInitRank, world 1): a training step whose four buckets go
    through fcn8s_allreduce_bucket (SUM over one rank = identity, scale 1/1) equals the plain step; the call-order errors are reported;
    fcn8s_bucket_wait makes a foreign stream wait for a bucket; the metrics all-reduce leaves one rank's counts unchanged."""
    from fcn8s_tensorflow_amd import _lib as L
    P = orc.init_params(20, SMALL, seed=1, decoder_std_scale=30.0, bias_std=0.05)
    img, lab = _batch(2, 32, 64, 3)
    a, b = _engine(), _engine()
    a.set_params(P); b.set_params(P)
    assert b.comm_info()["world"] == 0
    assert L.lib.fcn8s_allreduce_bucket(b.h, 0) == L.ERR_STATE and b"fcn8s_comm_init" in L.lib.fcn8s_last_error(b.h)
    b.comm_init_native()
    info = b.comm_info()
    assert info["world"] == 1 and info["rank"] == 0 and info["rccl_version"] >= 20000 and b.world_size == 1
    assert L.lib.fcn8s_comm_init(b.h, b"\0" * 128, 128, 0, 1) == L.ERR_STATE            # one communicator per model
    nb = b.num_buckets
    assert nb == 4 and [int(L.lib.fcn8s_bucket_complete_after(b.h, r)) for r in range(nb)] == list(range(nb))
    b._sync_stream()
    assert L.lib.fcn8s_bucket_wait(b.h, 0, None) == L.ERR_STATE                           # nothing queued yet
    la, sa = a.train_step(img, lab, 1e-3, keep_prob=1.0, optimizer=L.OPT_SGD_MOMENTUM)
    lb, sb = b.train_step(img, lab, 1e-3, keep_prob=1.0, optimizer=L.OPT_SGD_MOMENTUM)
    assert la == lb and sa == sb == 1
    assert float((a.flat_params - b.flat_params).abs().max()) <= 1e-6 * float(a.flat_params.abs().max())     # (weight-gradient atomics: order only)
    # a foreign stream can wait for each bucket after its call, and a bucket cannot be reduced twice in one pass
    b.forward_backward(img, lab, keep_prob=1.0)
    side = torch.cuda.Stream()
    for r in range(nb):
        assert L.lib.fcn8s_bucket_wait(b.h, r, C.c_void_p(side.cuda_stream)) == 0
    assert L.lib.fcn8s_allreduce_bucket(b.h, 1) == 0 and L.lib.fcn8s_allreduce_bucket(b.h, 1) == L.ERR_STATE
    assert L.lib.fcn8s_comm_wait(b.h) == 0
    side.synchronize(); torch.cuda.synchronize()
    b.metrics_reset(); b.eval_step(img, lab); before = b.metrics_raw()
    b.metrics_allreduce(); after = b.metrics_raw()
    np.testing.assert_array_equal(before[0], after[0]); assert before[1:] == after[1:] and after[0].sum() == lab.size
    b.broadcast_params(0)
    np.testing.assert_array_equal(b.get_params()["conv1_1/filter"], b.get_params()["conv1_1/filter"])
    assert L.lib.fcn8s_comm_destroy(b.h) == 0 and b.comm_info()["world"] == 0
    a.close(); b.close()
    # the launcher's NUMA binding starts from the GPU's PCI address and ends in a CPU set (or a reason why not)
    import re
    from fcn8s_tensorflow_amd import dp
    buf = C.create_string_buffer(32)
    assert L.lib.fcn8s_device_pci_bus_id(0, buf, 32) == 0 and re.match(r"^[0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-7]$", buf.value.decode()), buf.value
    assert L.lib.fcn8s_device_pci_bus_id(0, buf, 4) == L.ERR_BAD_ARG
    before = os.sched_getaffinity(0)
    try:
        info = dp.bind_to_gpu_numa(0, 1)
        assert info["bound"] or "why" in info, info
        if info["bound"]:
            assert len(os.sched_getaffinity(0)) == info["cpus"] > 0 and os.sched_getaffinity(0) <= before
    finally:
        os.sched_setaffinity(0, before)


def test_fused_step_marks_every_bucket_final():
    """after the fused fcn8s_train_step (no per-bucket calls) every gradient bucket -- bucket 2 included -- must be waitable
    (fcn8s_bucket_wait) and, with a communicator, reducible; the gradient readers order themselves behind pending all-reduces."""
    from fcn8s_tensorflow_amd import _lib as L
    P = orc.init_params(20, SMALL, seed=1, decoder_std_scale=30.0, bias_std=0.05)
    img, lab = _batch(2, 32, 64, 3)
    e = _engine(); e.set_params(P)
    loss, step = e.train_step(img, lab, 1e-3, keep_prob=1.0)          # TF-Adam, one rank, no process group: the fused C entry point
    assert step == 1 and np.isfinite(loss)
    side = torch.cuda.Stream()
    for b in range(e.num_buckets):
        assert L.lib.fcn8s_bucket_wait(e.h, b, C.c_void_p(side.cuda_stream)) == 0, (b, L.lib.fcn8s_last_error(e.h))
    side.synchronize()
    e.comm_init_native()                                               # world 1
    loss2, step2 = e.train_step(img, lab, 1e-3, keep_prob=1.0)
    assert step2 == 2 and np.isfinite(loss2)
    e.forward_backward(img, lab, keep_prob=1.0)
    for b in range(e.num_buckets):
        assert L.lib.fcn8s_allreduce_bucket(e.h, b) == 0
    g = e.get_grads()                                                  # (waits for the four all-reduces itself)
    assert all(np.isfinite(v).all() for v in g.values())
    assert e.get_option("comm_timeout_ms") == 600000
    e.set_option("comm_timeout_ms", 1234); assert e.get_option("comm_timeout_ms") == 1234
    e.comm_destroy()
    assert not e.native_comm and e.comm_info()["world"] == 0
    e.close()


def test_bench_single_gpu_line_and_end_to_end_mode():
    out = _run_bench("--steps", "2", "--warmup", "1", "--batch", "2", "--height", "64", "--width", "64", "--no-cpu-baseline")
    assert out["n_gpus"] == 1 and out["roofline"]["bound"] == "mfma" and 0 < out["roofline"]["frac"] < 1
    out = _run_bench("--mode", "e2e", "--steps", "2", "--warmup", "1", "--batch", "2", "--height", "64", "--width", "64", "--workers", "2")
    assert out["value"] > 0 and out["resident_input_images_per_sec"] > 0 and "end to end" in out["metric"]


@pytest.mark.parametrize("N,H,W,Cin,Cout", [
    (1, 15, 20, 256, 128),      # M = 300: <64,64> tiles, rows clamped in the last tile
    (1, 160, 161, 64, 128),     # M = 25 760: <128,128> tiles (>= 200 blocks), last tile partial
    (1, 160, 161, 128, 64),     # Cout = 64: <128,64> tiles
    (2, 40, 41, 512, 128),      # <64,128> tiles (between 200 blocks of 64x128 and 200 of 128x128)
    (1, 7, 9, 4096, 256),       # long K (256 K-tiles), few rows
])
def test_plain_row_gemm_kernels(N, H, W, Cin, Cout):
    """1x1 convolutions are the plain-row GEMMs of the LDS-DMA kernels (gemm_glds_kernel forward / data gradient with the fused
    bias + ReLU epilogue, wgrad_glds_kernel for the weight gradient, including its register-path K tail when the row count is
    not a multiple of 16): against a float64 oracle, fp32 summation-order tolerance."""
    from fcn8s_tensorflow_amd import _lib as L
    rng = np.random.default_rng(3)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((1, 1, Cin, Cout)) / np.sqrt(Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    dy = rng.standard_normal((N, H, W, Cout)).astype(np.float32)
    x64, w64 = x.reshape(-1, Cin).astype(np.float64), w.reshape(Cin, Cout).astype(np.float64)
    y_ref = x64 @ w64 + b
    dx_ref = dy.reshape(-1, Cout).astype(np.float64) @ w64.T
    dw_ref = x64.T @ dy.reshape(-1, Cout).astype(np.float64)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    ptr = lambda t: C.c_void_p(t.data_ptr())
    xd, wd, bd, dyd = dev(x), dev(w), dev(b), dev(dy)
    y_, dx_, dw_, db_ = torch.empty(N, H, W, Cout).cuda(), torch.empty(N, H, W, Cin).cuda(), torch.empty(1, 1, Cin, Cout).cuda(), torch.empty(Cout).cuda()
    L.check(L.lib.fcn8s_op_conv2d(None, ptr(xd), ptr(wd), ptr(bd), ptr(y_), N, H, W, Cin, Cout, 1, 1))
    L.check(L.lib.fcn8s_op_conv2d_bwd(None, ptr(xd), ptr(wd), ptr(dyd), ptr(dx_), ptr(dw_), ptr(db_), N, H, W, Cin, Cout, 1))
    torch.cuda.synchronize()
    rel = lambda a, r: float(np.abs(a.astype(np.float64) - r).max() / (np.abs(r).max() + 1e-30))
    assert rel(y_.cpu().numpy().reshape(-1, Cout), np.maximum(y_ref, 0)) < 2e-5
    assert rel(dx_.cpu().numpy().reshape(-1, Cin), dx_ref) < 2e-5
    assert rel(dw_.cpu().numpy().reshape(Cin, Cout), dw_ref) < 2e-5
    assert rel(db_.cpu().numpy(), dy.reshape(-1, Cout).astype(np.float64).sum(0)) < 2e-5


def test_gpu_resize_scale_translate_match_the_host_generator():
    """fcn8s_op_resample_u8 against the host path of this repo's BatchGenerator (`_apply` through cv2_compat: cv2.resize INTER_LINEAR /
    INTER_NEAREST and the integer-translation warpAffine of data_generator/batch_generator.py:328-384 in OpenCV's 8-bit arithmetic) on the
    same parameters: labels AND images bit-exact."""
    from fcn8s_tensorflow_amd.batch_generator import BatchGenerator
    e = _engine()
    rng = np.random.default_rng(11)
    N, H, W, void = 4, 64, 96, 7
    img = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
    lab = rng.integers(0, 20, (N, H, W), dtype=np.uint8)
    imgd, labd = torch.from_numpy(img).cuda(), torch.from_numpy(lab).cuda()

    def host(i, draw, resize=False):
        return BatchGenerator._apply(img[i], lab[i], draw, void, False, False, resize, False)

    def check(gi, gl, hi, hl, what):
        np.testing.assert_array_equal(gl, hl, err_msg=what)
        np.testing.assert_array_equal(gi, hi, err_msg=what)

    # resize (:328-331): every image to the same size, down and up
    for rs in ((45, 67), (83, 125), (32, 48), (64, 96), (32, 96), (21, 31), (128, 192), (200, 97)):       # (32, 48) = the exact 2x shrink
        gi, gl = e.resample(imgd, labd, out_hw=rs)
        for i in range(N):
            hi, hl = host(i, {}, resize=rs)
            check(gi[i].cpu().numpy(), gl[i].cpu().numpy(), hi, hl, "resize %s" % (rs,))
    # scale (:358-384): per-image factor, canvas placement for f <= 1, centre crop for f > 1
    factors = [0.61, 0.93, 1.27, 1.9]
    sizes = [(int(H * f), int(W * f)) for f in factors]
    offs = []
    for f, (sh, sw) in zip(factors, sizes):
        yo, xo = abs(int((H - sh) / 2)), abs(int((W - sw) / 2))
        offs.append((yo, xo) if f <= 1 else (-yo, -xo))
    gi, gl = e.resample(imgd, labd, out_hw=(H, W), sizes=sizes, offsets=offs, void_class_id=void)
    for i, f in enumerate(factors):
        hi, hl = host(i, {'factor': f})
        check(gi[i].cpu().numpy(), gl[i].cpu().numpy(), hi, hl, "scale %.2f" % f)
    # translate (:344-356): integer shifts, exact copy inside, void / zero outside
    shifts = [(5, -3), (-17, 0), (0, 11), (40, 30)]          # (x_shift, y_shift)
    gi, gl = e.resample(imgd, labd, out_hw=(H, W), sizes=[(H, W)] * N, offsets=[(dy, dx) for dx, dy in shifts], void_class_id=void)
    for i, sh in enumerate(shifts):
        hi, hl = host(i, {'shift': sh})
        np.testing.assert_array_equal(gi[i].cpu().numpy(), hi)
        np.testing.assert_array_equal(gl[i].cpu().numpy(), hl)
    # labels only / images only
    _, gl2 = e.resample(imgd, labd, out_hw=(45, 67))
    assert gl2.shape == (N, 45, 67)
    gi3, gl3 = e.resample(imgd, None, out_hw=(45, 67))
    assert gl3 is None and gi3.shape == (N, 45, 67, 3)
    e.close()

"""world_size-2 data-parallel path on CPU (gloo): the bucketed SUM all-reduce + 1/world scaling
reproduces the gradient of the global-mean loss.  Gradients come from the oracle (checker); the
code under test is fcn8s_tensorflow_amd.dp (bucket table from the C library, BucketReducer)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

WIDTHS = (4, 4, 8, 8, 8, 16, 16)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _flat(specs, total, grads):
    flat = np.zeros(total, np.float32)
    for k, (shape, off) in specs.items():
        flat[off:off + grads[k].size] = grads[k].reshape(-1)
    return flat


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import fcn8s_oracle as orc
    from fcn8s_tensorflow_amd import dp
    specs, total, buckets = dp.layout(4, WIDTHS, 3)
    P = orc.init_params(4, WIDTHS, fc6_ksize=3, seed=0, decoder_std_scale=50.0, bias_std=0.1)
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (2 * world, 32, 32, 3), dtype=np.uint8)
    lab = rng.integers(0, 4, (2 * world, 32, 32), dtype=np.uint8)
    sl = slice(2 * rank, 2 * rank + 2)                                 # this rank's shard
    _, g_local, _ = orc.loss_and_grads(P, img[sl], orc.one_hot(lab[sl], 4).astype(np.float32), l2_rate=0.01)
    flat = torch.from_numpy(_flat(specs, total, g_local))
    red = dp.BucketReducer(flat, buckets)
    for b in range(len(buckets)):
        red.reduce_bucket(b)
    red.wait()
    flat *= red.grad_scale()
    _, g_global, _ = orc.loss_and_grads(P, img, orc.one_hot(lab, 4).astype(np.float32), l2_rate=0.01)
    want = _flat(specs, total, g_global)
    err = float(np.abs(flat.numpy() - want).max() / np.abs(want).max())
    # confusion-matrix / loss reduction used by evaluate(): sums over ranks
    cm = torch.tensor(orc.confusion_matrix(lab[sl], (lab[sl] + rank) % 4, 4).reshape(-1)).double()
    dist.all_reduce(cm)
    out[rank] = (err, float(cm.sum()), red.grad_scale())
    dist.destroy_process_group()


def test_bucketed_allreduce_equals_global_batch_gradient():
    world = 2
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert len(out) == world
    for r in range(world):
        err, cm_total, scale = out[r]
        assert err < 1e-5, err
        assert cm_total == 2 * world * 32 * 32 and scale == 0.5


def test_bucket_reducer_is_a_noop_without_process_group():
    from fcn8s_tensorflow_amd import dp
    t = torch.ones(10)
    red = dp.BucketReducer(t, [(0, 5), (5, 5)])
    red.reduce_bucket(0); red.reduce_bucket(1); red.wait()
    assert red.grad_scale() == 1.0 and (t == 1).all()


def _guard_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fcn8s_tensorflow_amd import dp
    g = torch.Generator().manual_seed(3)
    flat = torch.randn(300000, generator=g)
    res = [dp.check_replicas(flat, 7)]                                  # identical replicas pass
    for mutate in ("step", "param", "swap", "nan"):
        f, step = flat.clone(), 7
        if rank == 1:
            if mutate == "step":
                step = 8
            elif mutate == "param":
                st = f.numel() // (1 << 16); i0 = step % st              # (the sample starts at global_step % stride)
                f[i0 + st * 5] += 1e-7 * (1 + abs(float(f[i0 + st * 5])))
            elif mutate == "swap":
                st = f.numel() // (1 << 16); i0 = step % st
                a, b = float(f[i0 + st * 10]), float(f[i0 + st * 20]); f[i0 + st * 10] = b; f[i0 + st * 20] = a
            else:
                f[step % (f.numel() // (1 << 16))] = float("nan")
        try:
            dp.check_replicas(f, step)
            res.append("passed")
        except RuntimeError as e:
            res.append(str(e))
    out[rank] = res
    dist.destroy_process_group()


def test_replica_guard_raises_on_every_rank_when_replicas_differ():
    """FCN8s.train's data-parallel guard (`global_step` identical on all ranks, replicas identical up to nothing): a
    differing step, a one-ulp parameter difference, two swapped values and a NaN are each seen by BOTH ranks."""
    world = 2
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_guard_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        res = out[r]
        assert res[0] is True
        assert "global step (7 .. 8)" in res[1], res[1]
        for msg in res[2:]:
            assert "parameter checksum" in msg, msg


def test_numa_share_and_cpulist():
    """Host logic of dp.bind_to_gpu_numa: the CPUs of a GPU's NUMA node, cut evenly among the ranks whose GPUs share the node."""
    from fcn8s_tensorflow_amd import dp
    assert dp.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and dp.parse_cpulist("") == []
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]                                   # eight GPUs on two sockets
    cpus = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    shares = [dp.numa_share(nodes, cpus, d) for d in range(8)]
    assert all(len(s_) == 32 for s_ in shares)
    assert sorted(sum(shares[:4], [])) == sorted(cpus[0]) and sorted(sum(shares[4:], [])) == sorted(cpus[1])      # disjoint, complete
    assert dp.numa_share([-1], cpus, 0) is None                                                # node unknown: leave the affinity alone
    assert dp.numa_share([0], {0: [0, 1, 2, 3]}, 0, allowed={2, 3, 9}) == [2, 3]                  # never outside the inherited affinity
    assert dp.numa_share([0, 0, 0], {0: [0, 1]}, 1) is None                                    # fewer CPUs than ranks: do not bind
    import torch
    if not torch.cuda.is_available():
        assert dp.bind_to_gpu_numa(0)["bound"] is False                                        # no GPU: reports why, never raises


def test_replica_fingerprint_sample_moves_with_the_step():
    """The guard looks at a strided sample; its offset is the global step modulo the stride, so a corrupted element outside one
    check's sample is inside a later one's."""
    import torch
    from fcn8s_tensorflow_amd import dp
    a = torch.arange(1000, dtype=torch.float32)
    b = a.clone(); b[7] += 1.0                                           # stride 10: element 7 is in the sample of steps 7, 17, ...
    same = [bool((dp.replica_fingerprint(a, t, samples=100) == dp.replica_fingerprint(b, t, samples=100)).all()) for t in range(10)]
    assert same == [True] * 7 + [False] + [True] * 2
    assert not bool((dp.replica_fingerprint(a, 0, samples=1000) == dp.replica_fingerprint(b, 0, samples=1000)).all())   # samples = numel: everything


def test_bench_prints_an_error_line_when_its_ranks_cannot_run():
    """`bench.py --gpus 2` on a box without a GPU: no rank can create its engine ("no CPU fallback").  The driver must still get exactly ONE
    JSON line on stdout -- value null, an `error` field -- and a non-zero exit code, not a hang or an empty stdout.
    On a GPU box the same contract is tested with a rank that dies mid-run (tests/test_multigpu_gpu.py)."""
    import json
    import subprocess
    import sys
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: covered by tests/test_multigpu_gpu.py")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in (["--gpus", "2", "--backend", "gloo", "--device", "0"], []):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1", "--batch", "1", "--height", "32", "--width", "32",
                            "--no-cpu-baseline", "--rank-timeout", "30"] + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        out = json.loads(lines[0])
        assert out["value"] is None and out["error"] and out["n_gpus"] == (2 if extra else 1) and out["metric"].startswith("training images/sec")

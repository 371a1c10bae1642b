"""Data-parallel plumbing of the FCN-8s path (new functionality: the reference is single-device).

One process per GPU; the minibatch is sharded over ranks (independent images) and the
only exchange is a SUM all-reduce of the flat gradient buffer, issued bucket by bucket in
backward-production order so that RCCL moves bucket b over xGMI while the compute stream runs
bucket b+1's backward kernels.  With equal shard sizes, (1/world) * sum of the local mean-loss
gradients equals the gradient of the global mean loss; the 1/world factor is folded into the
optimizer kernel (`grad_scale`), not applied as a separate pass over the 538 MB buffer.

Backend-agnostic (`nccl` = RCCL on the GPU box, `gloo` in the CPU tests).
"""
from __future__ import annotations

import ctypes as C


def dist_or_none():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def layout(num_classes, widths=None, fc6_ksize=7):
    """(specs, total_floats, buckets) of the flat variable buffer -- host logic of the C library,
    needs no GPU.  specs: name -> (shape, offset); buckets: [(offset, n)] in backward order."""
    from collections import OrderedDict
    from . import _lib as L
    cfg = L.Config()
    cfg.num_classes = int(num_classes); cfg.fc6_ksize = int(fc6_ksize)
    for i in range(7):
        cfg.widths[i] = int(widths[i]) if widths else 0
    total = L.lib.fcn8s_param_floats(C.byref(cfg))
    if total == 0:
        raise ValueError("invalid model configuration")
    specs = OrderedDict()
    for i in range(L.lib.fcn8s_layout_num_params(C.byref(cfg))):
        name = C.create_string_buffer(64); nd = C.c_int32(); shp = (C.c_int64 * 4)(); off = C.c_int64()
        L.check(L.lib.fcn8s_layout_param(C.byref(cfg), i, name, C.byref(nd), C.byref(shp), C.byref(off)))
        specs[name.value.decode()] = (tuple(int(shp[k]) for k in range(nd.value)), int(off.value))
    buckets = []
    for b in range(L.NUM_BUCKETS):
        o = C.c_size_t(); n = C.c_size_t()
        L.check(L.lib.fcn8s_layout_bucket(C.byref(cfg), b, C.byref(o), C.byref(n)))
        buckets.append((o.value, n.value))
    return specs, total, buckets


class BucketReducer:
    """Issues one asynchronous SUM all-reduce per gradient bucket and waits for all of them.
    trace: optional list; then every bucket appends (issue event on the compute stream, completion event on a side stream that
    waits for nothing but that all-reduce) -- bench.py turns them into issue -> complete timestamps per bucket."""

    def __init__(self, flat_grads, buckets, group=None, trace=None, always=False):
        self.flat = flat_grads
        self.buckets = buckets
        self.group = group
        self.works = []
        self.trace = trace
        self.always = always          # run the collectives in a one-rank group too (bench.py under torchrun with one rank)
        self._side = None

    def reduce_bucket(self, b):
        d = dist_or_none()
        if d is None or (d.get_world_size(self.group) == 1 and not self.always):
            return
        off, n = self.buckets[b]
        ev = None
        if self.trace is not None and self.flat.is_cuda:
            import torch
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
        w = d.all_reduce(self.flat[off:off + n], op=d.ReduceOp.SUM, group=self.group, async_op=True)
        self.works.append(w)
        if ev is not None:
            import torch
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.flat.device)
            done = torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(self._side):
                w.wait()                      # (NCCL/RCCL: the side stream waits for the collective; gloo: the host does)
                done.record()
            self.trace.append((b, ev, done))

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []

    def grad_scale(self):
        d = dist_or_none()
        return 1.0 if d is None else 1.0 / d.get_world_size(self.group)


def replica_fingerprint(flat_params, global_step, samples=1 << 16):
    """(global step, integer checksum of a strided sample of the parameter bits) of this rank's replica, as an int64 tensor on the
    parameters' device.  Bit patterns, not values: replicas must be *identical* (same all-reduced gradients, same update kernel),
    so any difference -- a rank that skipped a step, a diverged dropout-free buffer, a NaN -- changes the checksum."""
    import torch
    n = flat_params.numel()
    stride = max(1, n // int(samples))
    bits = flat_params.view(torch.int32)[::stride].to(torch.int64)
    w = torch.arange(1, bits.numel() + 1, dtype=torch.int64, device=bits.device)       # position-weighted: a swap of two values shows
    return torch.stack([torch.tensor(int(global_step), dtype=torch.int64, device=bits.device), (bits * w).sum()])


def check_replicas(flat_params, global_step, group=None):
    """Raises RuntimeError on every rank when the replicas of a data-parallel run are not identical (global step or parameter
    checksum differ between ranks).  One 32-byte all-reduce; FCN8s.train calls it every `replica_check_every` steps."""
    d = dist_or_none()
    if d is None or d.get_world_size(group) == 1:
        return True
    import torch
    fp = replica_fingerprint(flat_params, global_step)
    both = torch.cat([fp, -fp])                      # MAX of (x, -x) gives max and -min in one collective
    d.all_reduce(both, op=d.ReduceOp.MAX, group=group)
    both = both.cpu()
    hi, lo = both[:2], -both[2:]
    if not bool((hi == lo).all()):
        what = "global step (%d .. %d)" % (int(lo[0]), int(hi[0])) if int(hi[0]) != int(lo[0]) else "parameter checksum"
        raise RuntimeError("data-parallel replicas have diverged: %s differs between ranks (this rank: step %d)" % (what, int(global_step)))
    return True

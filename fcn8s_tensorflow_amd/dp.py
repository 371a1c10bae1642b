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
    for b in range(L.lib.fcn8s_layout_num_buckets(C.byref(cfg))):
        o = C.c_size_t(); n = C.c_size_t()
        L.check(L.lib.fcn8s_layout_bucket(C.byref(cfg), b, C.byref(o), C.byref(n)))
        buckets.append((o.value, n.value))
    return specs, total, buckets


def parse_cpulist(text):
    """'0-3,8,10-11' (sysfs cpulist syntax) -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def numa_share(node_of_device, cpus_of_node, device, allowed=None):
    """The CPUs rank `device` should run on: the CPUs of its GPU's NUMA node (restricted to `allowed`, the affinity the process was
    started with), cut into equal contiguous slices among the devices that sit on the same node.  Pure host logic (tests/test_dp_gloo.py).
    node_of_device: list, entry d = NUMA node of device d (-1 = unknown); cpus_of_node: dict node -> [cpu].  None = leave the affinity alone."""
    node = node_of_device[device]
    if node < 0 or not cpus_of_node.get(node):
        return None
    cpus = [c for c in cpus_of_node[node] if allowed is None or c in allowed]
    peers = [d for d, n in enumerate(node_of_device) if n == node]
    per = len(cpus) // len(peers)
    if per < 1:
        return None
    i = peers.index(device)
    return cpus[i * per:(i + 1) * per]


def bind_to_gpu_numa(device, local_world=None):
    """Bind this process -- EVERY thread it has (sched_setaffinity acts on one thread: the threads listed in /proc/self/task are bound one by one)
    and everything it starts later (threads and forked children inherit the caller's mask: the feeder thread, the BatchGenerator decode
    workers) -- to the CPUs of the NUMA node its GPU hangs off: pinned staging copies and PNG decode then stay on the memory controller next to
    the GPU's PCIe root.  Call it BEFORE creating the process group and the Engine, so that their helper threads are born with the mask; threads
    that exist already are re-bound here, but memory they have touched stays where it is.  Ranks whose GPUs share a node split its CPUs evenly.
    Returns a dict for the log / bench line; FCN8S_NUMA_BIND=0 switches it off."""
    import os
    from . import _lib as L
    info = {"bound": False}
    if os.environ.get("FCN8S_NUMA_BIND", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        info["why"] = "disabled"
        return info
    try:
        n = int(local_world if local_world is not None else os.environ.get("LOCAL_WORLD_SIZE", "1"))
        nodes, cpus = [], {}
        for d in range(max(n, device + 1)):
            buf = C.create_string_buffer(32)
            if L.lib.fcn8s_device_pci_bus_id(d, buf, 32) != 0:
                nodes.append(-1)
                continue
            base = "/sys/bus/pci/devices/" + buf.value.decode().lower()
            try:
                node = int(open(base + "/numa_node").read())
                if node >= 0 and node not in cpus:
                    cpus[node] = parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read())
            except OSError:
                node = -1
            nodes.append(node)
        allowed = os.sched_getaffinity(0)
        mine = numa_share(nodes, cpus, device, allowed)
        info.update({"numa_node": nodes[device], "devices_on_node": sum(1 for x in nodes if x == nodes[device] and x >= 0)})
        if mine:
            os.sched_setaffinity(0, mine)
            n_threads = 1
            try:                                        # the other threads of this process (torch's pools, a process group's watchdogs, ...)
                me = getattr(__import__("threading"), "get_native_id", lambda: None)()
                for tid in os.listdir("/proc/self/task"):
                    if me is not None and int(tid) == me:
                        continue
                    try:
                        os.sched_setaffinity(int(tid), mine)
                        n_threads += 1
                    except OSError:                     # (a thread that ended meanwhile)
                        pass
            except OSError:
                pass
            info.update({"bound": True, "cpus": len(mine), "first_cpu": mine[0], "last_cpu": mine[-1], "threads_bound": n_threads})
        else:
            info["why"] = "no NUMA node reported for this GPU" if nodes[device] < 0 else "no CPUs left to bind to"
    except Exception as ex:            # binding is an optimisation: never let it stop a run
        info["why"] = repr(ex)
    return info


class BucketReducer:
    """Issues one asynchronous SUM all-reduce per gradient bucket and waits for all of them.
    trace: optional list; then every bucket appends (issue event on the compute stream, completion event on a side stream that
    waits for nothing but that all-reduce) -- bench.py turns them into issue -> complete timestamps per bucket."""

    def __init__(self, flat_grads, buckets, group=None, trace=None, always=False, ready=None):
        self.flat = flat_grads
        self.buckets = buckets
        self.group = group
        self.works = []
        self.trace = trace
        self.always = always          # run the collectives in a one-rank group too (bench.py under torchrun with one rank)
        self._side = None
        self.ready = ready            # callable b -> torch stream that waits for bucket b's last gradient kernel (Engine._bucket_ready), or None

    def reduce_bucket(self, b):
        d = dist_or_none()
        if d is None or (d.get_world_size(self.group) == 1 and not self.always):
            return
        off, n = self.buckets[b]
        ev = None
        st = self.ready(b) if (self.ready is not None and self.flat.is_cuda) else None
        if self.trace is not None and self.flat.is_cuda:
            import torch
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(st) if st is not None else ev.record()      # (on the bucket's stream: the moment its last gradient kernel ended)
        if st is not None:
            # issued with the bucket's stream current: RCCL's stream then waits for the bucket's last gradient kernel only, not for whatever
            # the compute stream has queued behind it (fc6's bucket is final 3 ms of data-gradient GEMMs before its backward call ends)
            import torch
            with torch.cuda.stream(st):
                w = d.all_reduce(self.flat[off:off + n], op=d.ReduceOp.SUM, group=self.group, async_op=True)
        else:
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
    """(global step, integer checksum of a strided SAMPLE of the parameter bits) of this rank's replica, as an int64 tensor on the
    parameters' device.  Bit patterns, not values: replicas must be *identical* (same all-reduced gradients, same update kernel), so a
    rank that skipped a step, a diverged buffer or a NaN that has spread changes the checksum.  It is a sample (65 536 of 134 M elements
    per check): a difference confined to elements outside it passes this check; the sample's offset is the global step modulo the
    stride, so successive checks look at different elements, and `samples=numel` checks everything."""
    import torch
    n = flat_params.numel()
    stride = max(1, n // int(samples))
    first = int(global_step) % stride            # the sample moves with the step: over `stride` checks every element has been looked at
    bits = flat_params.view(torch.int32)[first::stride].to(torch.int64)
    w = torch.arange(1, bits.numel() + 1, dtype=torch.int64, device=bits.device)       # position-weighted: a swap of two values shows
    return torch.stack([torch.tensor(int(global_step), dtype=torch.int64, device=bits.device), (bits * w).sum()])


def check_replicas(flat_params, global_step, group=None):
    """Raises RuntimeError on every rank when the replicas of a data-parallel run are seen to differ (global step, or the checksum of
    the sampled parameter bits -- see replica_fingerprint -- differ between ranks).  One 32-byte all-reduce; FCN8s.train calls it every `replica_check_every` steps."""
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

"""Thin Python owner of one `fcn8s_model` handle (one process, one GPU).

PyTorch-ROCm is plumbing here: it owns the flat parameter / gradient buffers
(so that `torch.distributed` = RCCL can all-reduce them and a torch optimizer
can update them) and provides the HIP stream.  All arithmetic of the FCN-8s
path happens inside libfcn8s_hip.so.

Data-parallel training: one Engine per rank, the minibatch is
sharded by the caller, gradients are all-reduced in four buckets in
backward-production order ({fc7, decoder} | {fc6} | {conv5, conv4} | {conv3..1});
a bucket's all-reduce starts behind the last kernel that writes into it
(fcn8s_bucket_wait) and runs on RCCL's stream while the rest of the backward
pass runs on the compute stream.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _lib as L
from .dp import BucketReducer


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


_CLASS_AXES = {   # axes of the decoder variables that run over classes (fcn8s_tensorflow.py:173-233)
    'pool3_1x1/kernel': (3,), 'pool3_1x1/bias': (0,), 'pool4_1x1/kernel': (3,), 'pool4_1x1/bias': (0,),
    'fc7_1x1/kernel': (3,), 'fc7_1x1/bias': (0,),
    'fc7_conv2d_trans/kernel': (2, 3), 'fc7_conv2d_trans/bias': (0,),
    'fc7_pool4_conv2d_trans/kernel': (2, 3), 'fc7_pool4_conv2d_trans/bias': (0,),
    'fc7_pool4_pool3_conv2d_trans/kernel': (2, 3), 'fc7_pool4_pool3_conv2d_trans/bias': (0,)}
_PAD_LOGIT = -1e30   # bias of a padding class in the last layer: its softmax probability is exactly 0, so is every gradient that touches it


def padded_classes(num_classes):
    """The library's class dimension: the next multiple of 4 (its C-channel tensors are accessed 16 bytes at a time)."""
    return (int(num_classes) + 3) // 4 * 4


def _pad_classes(name, a, c, cpad, slot=False):
    """Embed a decoder variable with `c` classes into the `cpad`-class shape: padding weights 0 (a padding channel then carries
    exactly 0 through the decoder and receives exactly 0 gradient -- the subspace is invariant under training), padding entries of
    the last bias _PAD_LOGIT."""
    axes = _CLASS_AXES.get(name)
    if axes is None or c == cpad:
        return a
    a = np.asarray(a)
    pad = [(0, cpad - c) if i in axes else (0, 0) for i in range(a.ndim)]
    fill = _PAD_LOGIT if (name == 'fc7_pool4_pool3_conv2d_trans/bias' and not slot) else 0.0
    return np.pad(a, pad, constant_values=fill)


def _unpad_classes(name, a, c, cpad):
    axes = _CLASS_AXES.get(name)
    if axes is None or c == cpad:
        return a
    sl = tuple(slice(0, c) if i in axes else slice(None) for i in range(np.ndim(a)))
    return np.ascontiguousarray(np.asarray(a)[sl])


class Staged:
    """One host batch on its way to the GPU through a staging slot (Engine.stage): pinned copy + H2D on the library's copy
    stream, started by whoever called stage() -- typically the feeder thread, while the compute stream runs the previous step."""
    __slots__ = ("slot", "img_ptr", "lab_ptr", "dtype", "nhw")

    def __init__(self, slot, img_ptr, lab_ptr, dtype, nhw):
        self.slot, self.img_ptr, self.lab_ptr, self.dtype, self.nhw = slot, img_ptr, lab_ptr, dtype, nhw


class Engine:
    def __init__(self, num_classes, widths=None, fc6_ksize=7, device_id=0, seed=0, process_group=None, precision='fp32', logical_classes=None,
                 options=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("fcn8s_tensorflow_amd needs an AMD GPU (gfx950); there is no CPU fallback for the hot path")
        self.torch = torch
        self.device = torch.device("cuda", device_id)
        torch.cuda.set_device(self.device)
        self.num_classes = int(num_classes)
        # classes the caller sees when `num_classes` was padded to the library's multiple of 4 (FCN8s facade): one-hot labels carry this many channels
        self.logical_classes = int(logical_classes) if logical_classes else int(num_classes)
        self.pg = process_group
        cfg = L.Config()
        cfg.num_classes = int(num_classes)
        cfg.fc6_ksize = int(fc6_ksize)
        for i in range(7):
            cfg.widths[i] = int(widths[i]) if widths else 0
        cfg.device_id = device_id
        cfg.seed = int(seed)
        n = L.lib.fcn8s_param_floats(C.byref(cfg))
        if n == 0:
            raise ValueError("invalid model configuration (num_classes must be a positive multiple of 4)")
        self.flat_params = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.flat_grads = torch.zeros(n, dtype=torch.float32, device=self.device)
        cfg.ext_params = self.flat_params.data_ptr()
        cfg.ext_grads = self.flat_grads.data_ptr()
        h = C.c_void_p()
        L.check(L.lib.fcn8s_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.widths = tuple(widths) if widths else (64, 128, 256, 512, 512, 4096, 4096)
        self.set_precision(precision)
        for k, v in (options or {}).items():        # algorithm variants for parity reports (include/fcn8s_hip.h: fcn8s_set_option)
            self.set_option(k, v)
        self.specs = OrderedDict()          # name -> (shape, offset)
        for i in range(L.lib.fcn8s_num_params(self.h)):
            name = C.c_char_p(); nd = C.c_int32(); shp = (C.c_int64 * 4)(); off = C.c_int64()
            L.check(L.lib.fcn8s_param_info(self.h, i, C.byref(name), C.byref(nd), C.byref(shp), C.byref(off)), self.h)
            self.specs[name.value.decode()] = (tuple(int(shp[k]) for k in range(nd.value)), int(off.value))
        self.buckets = []
        self.num_buckets = int(L.lib.fcn8s_num_buckets(self.h))
        self.native_comm = False                    # True: gradients are exchanged by the library's own RCCL communicator (comm_init_native)
        self._side = None
        for b in range(self.num_buckets):
            o = C.c_size_t(); m = C.c_size_t()
            L.check(L.lib.fcn8s_bucket_range(self.h, b, C.byref(o), C.byref(m)), self.h)
            self.buckets.append((o.value, m.value))
        self._label_checks_left = 2
        self._label_calls = 0
        self._bad = None
        self.replica_check_every = 100      # data-parallel runs: compare global step + parameter checksum across ranks every so many steps (0 = never)
        self._sync_stream()

    # ---- plumbing ---------------------------------------------------------------------
    def _sync_stream(self):
        L.check(L.lib.fcn8s_set_stream(self.h, C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)), self.h)

    def close(self):
        if getattr(self, "h", None) is not None:
            self.torch.cuda.synchronize(self.device)
            L.lib.fcn8s_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def world_size(self):
        if getattr(self, "native_comm", False):
            return self.comm_world
        d = _dist()
        return d.get_world_size(self.pg) if d else 1

    @property
    def rank(self):
        if getattr(self, "native_comm", False):
            return self.comm_rank
        d = _dist()
        return d.get_rank(self.pg) if d else 0

    # ---- variables -----------------------------------------------------------------------
    def param_view(self, name):
        shape, off = self.specs[name]
        return self.flat_params[off:off + int(np.prod(shape))].view(*shape)

    def logical_param_view(self, name):
        """param_view without the padding classes (a strided view when num_classes was padded to a multiple of 4)."""
        v = self.param_view(name)
        if self.logical_classes != self.num_classes:
            for ax in _CLASS_AXES.get(name, ()):
                v = v.narrow(ax, 0, self.logical_classes)
        return v

    def grad_view(self, name):
        shape, off = self.specs[name]
        return self.flat_grads[off:off + int(np.prod(shape))].view(*shape)

    def set_params(self, params):
        for k, v in params.items():
            if k not in self.specs:
                raise ValueError("unknown variable '%s'" % k)
            a = np.ascontiguousarray(self.pad(k, np.asarray(v, dtype=np.float32)), dtype=np.float32)
            if tuple(a.shape) != self.specs[k][0]:
                raise ValueError("variable '%s' has shape %s, expected %s" % (k, a.shape, self.specs[k][0]))
            L.check(L.lib.fcn8s_set_param(self.h, k.encode(), a.ctypes.data_as(C.c_void_p), a.size), self.h)

    def get_params(self):
        out = OrderedDict()
        for k, (shape, _) in self.specs.items():
            a = np.empty(shape, np.float32)
            L.check(L.lib.fcn8s_get_param(self.h, k.encode(), a.ctypes.data_as(C.c_void_p), a.size), self.h)
            out[k] = self.unpad(k, a)
        return out

    def get_grads(self):
        out = OrderedDict()
        for k, (shape, _) in self.specs.items():
            a = np.empty(shape, np.float32)
            L.check(L.lib.fcn8s_get_grad(self.h, k.encode(), a.ctypes.data_as(C.c_void_p), a.size), self.h)
            out[k] = self.unpad(k, a)
        return out

    def pad(self, name, a, slot=False):
        """A variable (or, slot=True, one of its optimizer slots) with `logical_classes` classes in the library's padded shape."""
        return _pad_classes(name, a, self.logical_classes, self.num_classes, slot)

    def unpad(self, name, a):
        return _unpad_classes(name, a, self.logical_classes, self.num_classes)

    def init_params(self, seed=0):
        self._sync_stream()
        L.check(L.lib.fcn8s_init_params(self.h, int(seed)), self.h)
        if self.logical_classes != self.num_classes:           # padding classes: zero weights, probability-0 bias in the last layer
            c = self.logical_classes
            for name, axes in _CLASS_AXES.items():
                v = self.param_view(name)
                fill = _PAD_LOGIT if name == 'fc7_pool4_pool3_conv2d_trans/bias' else 0.0
                for ax in axes:
                    v.narrow(ax, c, self.num_classes - c).fill_(fill)

    def broadcast_params(self, src=0):
        if self.native_comm:
            self._sync_stream()
            L.check(L.lib.fcn8s_comm_broadcast_params(self.h, int(src)), self.h)
            return
        d = _dist()
        if d and self.world_size > 1:
            self.freeze(False)
            d.broadcast(self.flat_params, src=src, group=self.pg)

    @property
    def global_step(self):
        return int(L.lib.fcn8s_global_step(self.h))

    @global_step.setter
    def global_step(self, v):
        L.check(L.lib.fcn8s_set_global_step(self.h, int(v)), self.h)

    def get_opt_state(self):
        n = self.flat_params.numel()
        m = np.empty(n, np.float32); v = np.empty(n, np.float32)
        L.check(L.lib.fcn8s_get_opt_state(self.h, m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), n), self.h)
        return m, v

    def set_opt_state(self, m, v):
        m = np.ascontiguousarray(m, np.float32); v = np.ascontiguousarray(v, np.float32)
        L.check(L.lib.fcn8s_set_opt_state(self.h, m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), m.size), self.h)

    # ---- input marshalling -----------------------------------------------------------------
    def _images(self, images, force_device=False):
        """-> (keepalive, pointer, dtype code, where, (N,H,W))"""
        torch = self.torch
        if isinstance(images, (list, tuple)):
            images = np.asarray(images)
        if isinstance(images, torch.Tensor):
            t = images
            if t.dim() != 4 or t.shape[-1] != 3:
                raise ValueError("images must have shape (batch, height, width, 3)")
            if t.dtype not in (torch.uint8, torch.float32):
                t = t.float()
            if t.is_cuda or force_device:
                t = t.to(self.device).contiguous()
                return t, C.c_void_p(t.data_ptr()), (L.IMG_U8 if t.dtype == torch.uint8 else L.IMG_F32), L.DEVICE, tuple(t.shape[:3])
            images = t.numpy()
        a = np.asarray(images)
        if a.ndim != 4 or a.shape[-1] != 3:
            raise ValueError("images must be an array-like of rank 4 with shape (batch, height, width, 3)")
        if a.dtype != np.uint8:
            a = a.astype(np.float32, copy=False)
        a = np.ascontiguousarray(a)
        if force_device:
            t = torch.from_numpy(a).to(self.device)
            return t, C.c_void_p(t.data_ptr()), (L.IMG_U8 if a.dtype == np.uint8 else L.IMG_F32), L.DEVICE, a.shape[:3]
        return a, a.ctypes.data_as(C.c_void_p), (L.IMG_U8 if a.dtype == np.uint8 else L.IMG_F32), L.HOST, a.shape[:3]

    def _check_ids(self, ids):
        """Class ids that would select a padding class (num_classes padded to a multiple of 4 by the facade) are an error, as
        `np.eye(num_classes)[ids]` is in the reference (helpers/ground_truth_conversion_utils.py:84-88: IndexError); ids at or above
        the padded count stay what include/fcn8s_hip.h says they are: 'ignore'.  Only costs anything when classes are padded."""
        if self.logical_classes == self.num_classes:
            return
        lo, hi = self.logical_classes, self.num_classes
        if isinstance(ids, np.ndarray):
            bad = bool(((ids >= lo) & (ids < hi)).any())
        else:
            bad = bool(((ids >= lo) & (ids < hi)).any().item())
        if bad:
            raise ValueError("class ids must be below num_classes = %d (ids from %d on mean 'ignore')" % (lo, hi))

    def _onehot_to_ids(self, t, nhw):
        """One-hot rows (N,H,W,C) on the device -> uint8 class ids on the device (library kernel).
        The first batches are also checked for being one-hot (costs one host sync each)."""
        torch = self.torch
        if t.shape[-1] != self.logical_classes:
            raise ValueError("one-hot labels must have %d channels, got %d" % (self.logical_classes, t.shape[-1]))
        if tuple(t.shape[:3]) != tuple(nhw):
            raise ValueError("labels shape %s does not match images %s" % (tuple(t.shape), tuple(nhw)))
        if t.dtype in (torch.bool, torch.uint8, torch.int8):
            eb = 1
        elif t.dtype in (torch.int32, torch.float32):
            eb = 4
        else:
            t = t.to(torch.int32); eb = 4
        t = t.to(self.device).contiguous()
        npix = int(nhw[0]) * int(nhw[1]) * int(nhw[2])
        ids = torch.empty(tuple(int(x) for x in nhw), dtype=torch.uint8, device=self.device)
        # every batch adds its count of rows that are not one-hot to a device counter (no sync); the host looks at it on the first
        # two batches and then every 64th (all-zero rows train as "ignore", multi-hot rows as their first class -- see include/fcn8s_hip.h)
        if self._bad is None:
            self._bad = torch.zeros(1, dtype=torch.int32, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        L.check(L.lib.fcn8s_onehot_to_ids(stream, C.c_void_p(t.data_ptr()), eb, npix, self.logical_classes,
                                          C.c_void_p(ids.data_ptr()), C.c_void_p(self._bad.data_ptr())))
        self._label_calls += 1
        if self._label_checks_left > 0 or self._label_calls % 64 == 0:
            self._label_checks_left = max(0, self._label_checks_left - 1)
            nbad = int(self._bad.item())
            if nbad:
                self._bad.zero_()
                raise ValueError("labels must be one-hot along the last axis (%d rows of the last %s were not); "
                                 "soft labels are not supported" % (nbad, "batch" if self._label_calls <= 2 else "64 batches"))
        return ids

    def _inputs(self, images, labels):
        """Marshals one (images, labels) feed.  One-hot labels (the reference's format, fcn8s_tensorflow.py:110,
        429-433) are shipped to the GPU and reduced to uint8 class ids there -- an np.argmax over the
        168 MB one-hot batch on the host would cost more than the whole training step."""
        torch = self.torch
        if isinstance(images, Staged):
            st = images
            L.check(L.lib.fcn8s_stage_wait(self.h, st.slot), self.h)
            return st, C.c_void_p(st.img_ptr), st.dtype, C.c_void_p(st.lab_ptr) if st.lab_ptr else None, L.DEVICE, st.nhw
        lab_is_tensor = isinstance(labels, torch.Tensor)
        lab_rank = labels.dim() if lab_is_tensor else np.ndim(labels)
        onehot = lab_rank == 4
        ka_i, pi, dt, where, nhw = self._images(images, force_device=onehot or (lab_is_tensor and labels.is_cuda))
        if onehot:
            t = labels if lab_is_tensor else torch.from_numpy(np.ascontiguousarray(labels))
            ids = self._onehot_to_ids(t, nhw)
            return (ka_i, ids), pi, dt, C.c_void_p(ids.data_ptr()), where, nhw
        if lab_rank != 3:
            raise ValueError("labels must be one-hot (batch, height, width, num_classes) or class ids (batch, height, width)")
        if lab_is_tensor:
            t = labels
            if tuple(t.shape) != tuple(nhw):
                raise ValueError("labels shape %s does not match images %s" % (tuple(t.shape), tuple(nhw)))
            t = t.to(torch.uint8).contiguous()
            if where == L.DEVICE:
                t = t.to(self.device)
                self._check_ids(t)
                return (ka_i, t), pi, dt, C.c_void_p(t.data_ptr()), where, nhw
            labels = t.cpu().numpy()
        a = np.asarray(labels)
        if tuple(a.shape) != tuple(nhw):
            raise ValueError("labels shape %s does not match images %s" % (tuple(a.shape), tuple(nhw)))
        a = np.ascontiguousarray(a, dtype=np.uint8)
        self._check_ids(a)
        if where == L.DEVICE:
            t = torch.from_numpy(a).to(self.device)
            return (ka_i, t), pi, dt, C.c_void_p(t.data_ptr()), where, nhw
        return (ka_i, a), pi, dt, a.ctypes.data_as(C.c_void_p), where, nhw

    def stage(self, images, label_ids=None, slot=0):
        """Start moving one host batch to the GPU (include/fcn8s_hip.h: fcn8s_stage_inputs) and return a `Staged` handle that
        train_step / eval_step / predict accept in place of the arrays.  images: uint8 or float32 array (N,H,W,3); label_ids:
        uint8 class ids (N,H,W) or None.  Thread-safe against the compute calls of this engine (it only touches the staging slot
        and the copy stream), so a feeder thread can stage batch k+1 while step k runs."""
        a = np.asarray(images)
        if a.ndim != 4 or a.shape[-1] != 3:
            raise ValueError("images must be an array-like of rank 4 with shape (batch, height, width, 3)")
        if a.dtype != np.uint8:
            a = a.astype(np.float32, copy=False)
        a = np.ascontiguousarray(a)
        lab = None
        if label_ids is not None:
            lab = np.ascontiguousarray(label_ids, dtype=np.uint8)
            if tuple(lab.shape) != tuple(a.shape[:3]):
                raise ValueError("labels shape %s does not match images %s" % (tuple(lab.shape), tuple(a.shape[:3])))
            self._check_ids(lab)
        pi, pl = C.c_void_p(), C.c_void_p()
        N, H, W = (int(x) for x in a.shape[:3])
        dt = L.IMG_U8 if a.dtype == np.uint8 else L.IMG_F32
        L.check(L.lib.fcn8s_stage_inputs(self.h, int(slot), a.ctypes.data_as(C.c_void_p), dt,
                                         lab.ctypes.data_as(C.c_void_p) if lab is not None else None, N, H, W, C.byref(pi), C.byref(pl)), self.h)
        return Staged(int(slot), pi.value, pl.value, dt, (N, H, W))

    def _release(self, ka):
        if isinstance(ka, Staged):
            L.check(L.lib.fcn8s_stage_release(self.h, ka.slot), self.h)

    # ---- hot path ----------------------------------------------------------------------------
    def freeze(self, frozen=True):
        """Promise (or withdraw the promise) that the parameters stay constant: inside evaluate() / predict loops the library then
        keeps the Winograd-transformed filter banks across calls.  Library calls that change parameters unfreeze on their own;
        code that writes into `flat_params` through torch must call `freeze(False)` first."""
        L.check(L.lib.fcn8s_freeze_params(self.h, 1 if frozen else 0), self.h)

    def set_option(self, key, value):
        """fcn8s_set_option: 'winograd_min_cin', 'winograd_tile', 'winograd_fc6', 'tconv_gemm' (defaults = the measured winners)."""
        L.check(L.lib.fcn8s_set_option(self.h, key.encode(), int(value)), self.h)

    def get_option(self, key):
        v = C.c_int64()
        L.check(L.lib.fcn8s_get_option(self.h, key.encode(), C.byref(v)), self.h)
        return int(v.value)

    def set_precision(self, precision):
        """'fp32' (the reference's arithmetic), 'bf16_fc' (BASELINE config 5: forward fc6 / fc7 with bf16-rounded
        operands on the bf16 MFMA, fp32 accumulation; the rest of the step stays fp32), 'f32x3' (every large GEMM on the
        bf16 MFMA with each fp32 operand split exactly into three bf16 pieces: fp32 accuracy, not bit-identical to fp32) or
        'bf16_fwd' (config 5 taken further: conv3_1 .. conv5_3, fc6 and fc7 forward with bf16-rounded operands, every other GEMM
        in the f32x3 arithmetic: all matrix work on the bf16 MFMA); 'f32x2' / 'bf16_fwd_x2' = 'f32x3' / 'bf16_fwd' with two bf16
        pieces per operand instead of three (16 significand bits enter each product: reduced precision, half the matrix work);
        'bf16_train' = mixed-precision training as it is usually meant: conv1_2 .. conv5_3, fc6 and fc7 as direct convolutions whose operands are
        rounded to bf16 in the forward pass, the data gradient AND the weight gradient (fp32 accumulate, fp32 master weights, everything else
        exact fp32; no Winograd transforms)."""
        modes = {'fp32': L.PREC_F32, 'bf16_fc': L.PREC_BF16_FC, 'f32x3': L.PREC_F32X3, 'bf16_fwd': L.PREC_BF16_FWD,
                 'f32x2': L.PREC_F32X2, 'bf16_fwd_x2': L.PREC_BF16_FWD_X2, 'bf16_train': L.PREC_BF16_TRAIN}
        if precision not in modes:
            raise ValueError("`precision` must be 'fp32', 'bf16_fc', 'f32x3', 'bf16_fwd', 'f32x2', 'bf16_fwd_x2' or 'bf16_train', but is '{}'.".format(precision))
        L.check(L.lib.fcn8s_set_precision(self.h, modes[precision]), self.h)
        self.precision = precision

    def train_step(self, images, labels, learning_rate, keep_prob=0.5, l2_rate=0.0,
                   optimizer=L.OPT_TF_ADAM, fetch_loss=True, reduce=True):
        """sess.run([train_op, total_loss, global_step]) (fcn8s_tensorflow.py:554-572).
        `reduce=False` skips the gradient exchange of a data-parallel run (measurement only: bench.py's local-only leg)."""
        self._sync_stream()
        ka, pi, dt, pl, where, nhw = self._inputs(images, labels)
        N, H, W = (int(x) for x in nhw)
        ws = self.world_size
        loss = C.c_float(0.0)
        always = bool(getattr(self, "dp_always", False)) and _dist() is not None
        if ws == 1 and optimizer == L.OPT_TF_ADAM and not always and not self.native_comm:
            step = C.c_int64(0)
            L.check(L.lib.fcn8s_train_step(self.h, pi, dt, pl, N, H, W, float(learning_rate), float(keep_prob),
                                           float(l2_rate), where, None, C.byref(step)), self.h)
            self._release(ka)
            if fetch_loss:
                L.check(L.lib.fcn8s_read_loss(self.h, C.byref(loss)), self.h)
            return (float(loss.value) if fetch_loss else None), int(step.value)
        trace = getattr(self, "comm_trace", None)          # bench.py: a list collects (step start, per-bucket issue / completion events)
        if trace is not None:                              # t0 in front of the forward pass: the traced times are "ms after the step's first kernel"
            t0 = self.torch.cuda.Event(enable_timing=True); t0.record()
            trace.append(("step", t0, None))
        L.check(L.lib.fcn8s_forward_loss(self.h, pi, dt, pl, N, H, W, float(keep_prob), float(l2_rate), where), self.h)
        self._release(ka)
        red = BucketReducer(self.flat_grads, self.buckets, self.pg, trace=trace, always=always, ready=self._bucket_ready)
        # bucket b is final once the call `ready_after[b]` has returned (include/fcn8s_hip.h) -- by default call b itself: {fc7, decoder},
        # {fc6}, {conv4, conv5}, {conv1 .. conv3} leave in that order, each while the rest of the backward pass runs
        nb = self.num_buckets
        ready_after = [int(L.lib.fcn8s_bucket_complete_after(self.h, b)) for b in range(nb)]
        for b in range(nb):
            L.check(L.lib.fcn8s_backward_bucket(self.h, b), self.h)
            if reduce:
                for r in range(nb):
                    if ready_after[r] == b:
                        if self.native_comm:
                            L.check(L.lib.fcn8s_allreduce_bucket(self.h, r), self.h)
                        else:
                            red.reduce_bucket(r)
        red.wait()
        scale = (1.0 / self.comm_world) if (self.native_comm and reduce) else (red.grad_scale() if reduce else 1.0)
        L.check(L.lib.fcn8s_apply_update(self.h, optimizer, float(learning_rate), scale), self.h)     # (waits for the native all-reduces itself)
        if trace is not None:
            t1 = self.torch.cuda.Event(enable_timing=True); t1.record()
            trace.append(("end", t1, None))
        if fetch_loss:
            L.check(L.lib.fcn8s_read_loss(self.h, C.byref(loss)), self.h)
        step = self.global_step
        every = int(getattr(self, "replica_check_every", 0) or 0)
        if reduce and ws > 1 and every > 0 and step % every == 0:
            self.check_replicas()
        return (float(loss.value) if fetch_loss else None), step

    def _bucket_ready(self, b):
        """A torch side stream that waits for exactly the last kernel writing into gradient bucket b (fcn8s_bucket_wait): a collective
        issued with that stream current starts when the bucket is final, not when everything queued behind it has run."""
        torch = self.torch
        if self._side is None:
            self._side = [torch.cuda.Stream(device=self.device) for _ in range(self.num_buckets)]
        st = self._side[b]
        L.check(L.lib.fcn8s_bucket_wait(self.h, b, C.c_void_p(st.cuda_stream)), self.h)
        return st

    def comm_init_native(self, unique_id=None, rank=None, world=None):
        """Give this model a RCCL rank of its own inside libfcn8s_hip.so (fcn8s_comm_init): from then on train_step exchanges the gradient
        buckets through fcn8s_allreduce_bucket instead of torch.distributed.  Without arguments the unique id is made by rank 0 of the
        current torch.distributed group and handed round through it (any backend: the group only carries 128 bytes, once)."""
        d = _dist()
        if unique_id is None:
            rank = d.get_rank(self.pg) if d else 0
            world = d.get_world_size(self.pg) if d else 1
            buf = C.create_string_buffer(L.COMM_ID_BYTES)
            if rank == 0:
                L.check(L.lib.fcn8s_comm_unique_id(buf, L.COMM_ID_BYTES))
            if d and world > 1:
                box = [bytes(buf.raw)]
                d.broadcast_object_list(box, src=d.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)
                buf = C.create_string_buffer(box[0], L.COMM_ID_BYTES)
            unique_id = buf.raw
        self._sync_stream()
        L.check(L.lib.fcn8s_comm_init(self.h, unique_id, len(unique_id), int(rank), int(world)), self.h)
        self.native_comm, self.comm_world, self.comm_rank = True, int(world), int(rank)
        return unique_id

    def comm_destroy(self):
        """fcn8s_comm_destroy: give the library's RCCL rank back; train_step exchanges through torch.distributed again (or not at all).
        Raises with RCCL's / the watchdog's text if the communicator had failed (a dead or hung peer: include/fcn8s_hip.h)."""
        self._sync_stream()
        self.native_comm, self.comm_world, self.comm_rank = False, 1, 0
        L.check(L.lib.fcn8s_comm_destroy(self.h), self.h)

    def comm_info(self):
        r, w, v = C.c_int(), C.c_int(), C.c_int()
        L.check(L.lib.fcn8s_comm_info(self.h, C.byref(r), C.byref(w), C.byref(v)), self.h)
        return {"rank": r.value, "world": w.value, "rccl_version": v.value}

    def check_replicas(self):
        """Data-parallel guard (dp.check_replicas): every rank must hold the same global step and the same bits in a sample of the
        parameters (the sample moves with the step; replicas are identical by construction, this catches the day they are not)."""
        from .dp import check_replicas          # (the library works on torch's current stream: the reads below are ordered behind the update)
        return check_replicas(self.flat_params, self.global_step, self.pg)

    def forward_backward(self, images, labels, keep_prob=1.0, l2_rate=0.0):
        """Gradients only (no update): returns the loss; gradients via get_grads()/grad_view()."""
        self._sync_stream()
        ka, pi, dt, pl, where, nhw = self._inputs(images, labels)
        N, H, W = (int(x) for x in nhw)
        L.check(L.lib.fcn8s_forward_loss(self.h, pi, dt, pl, N, H, W, float(keep_prob), float(l2_rate), where), self.h)
        self._release(ka)
        for b in range(self.num_buckets):
            L.check(L.lib.fcn8s_backward_bucket(self.h, b), self.h)
        loss = C.c_float(0.0)
        L.check(L.lib.fcn8s_read_loss(self.h, C.byref(loss)), self.h)
        return float(loss.value)

    def apply_update(self, learning_rate, optimizer=L.OPT_TF_ADAM, grad_scale=1.0):
        self._sync_stream()
        L.check(L.lib.fcn8s_apply_update(self.h, optimizer, float(learning_rate), float(grad_scale)), self.h)

    def eval_step(self, images, labels, l2_rate=0.0):
        """sess.run(metric_update_ops) (fcn8s_tensorflow.py:685-689), keep_prob = 1."""
        self._sync_stream()
        ka, pi, dt, pl, where, nhw = self._inputs(images, labels)
        N, H, W = (int(x) for x in nhw)
        L.check(L.lib.fcn8s_eval_step(self.h, pi, dt, pl, N, H, W, float(l2_rate), where), self.h)
        self._release(ka)

    def metrics_reset(self):
        self._sync_stream()
        L.check(L.lib.fcn8s_metrics_reset(self.h), self.h)

    def _metrics_raw_padded(self):
        Cn = self.num_classes
        cm = np.zeros((Cn, Cn), np.int64); ls = C.c_double(); lc = C.c_int64()
        L.check(L.lib.fcn8s_metrics_raw(self.h, cm.ctypes.data_as(C.c_void_p), C.byref(ls), C.byref(lc)), self.h)
        return cm, float(ls.value), int(lc.value)

    def metrics_raw(self):
        """(confusion matrix [label, prediction] over the caller's classes, sum of the per-batch losses, number of batches)"""
        cm, ls, lc = self._metrics_raw_padded()
        c = self.logical_classes
        return np.ascontiguousarray(cm[:c, :c]), ls, lc

    def metrics_allreduce(self):
        """Sum the confusion matrix and the per-batch loss samples over ranks."""
        if self.native_comm:
            self._sync_stream()
            L.check(L.lib.fcn8s_comm_allreduce_metrics(self.h), self.h)
            return
        d = _dist()
        if not d or self.world_size == 1:
            return
        torch = self.torch
        cm, ls, lc = self._metrics_raw_padded()
        t = torch.cat([torch.from_numpy(cm.reshape(-1)).double(), torch.tensor([ls, float(lc)], dtype=torch.float64)]).to(self.device)
        d.all_reduce(t, group=self.pg)
        t = t.cpu()
        cm = t[:-2].round().long().numpy().reshape(cm.shape).astype(np.int64)
        L.check(L.lib.fcn8s_metrics_set_raw(self.h, np.ascontiguousarray(cm).ctypes.data_as(C.c_void_p), float(t[-2]), int(round(float(t[-1])))), self.h)

    def metrics_get(self, all_classes=False):
        """(mean loss, mean IoU, accuracy).  all_classes: average the IoU over all classes, absent ones counting 0 (what the
        tutorial's TF 1.3.0 tf.metrics.mean_iou does) instead of over the classes that occur (later TF 1.x)."""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        L.check(L.lib.fcn8s_metrics_get_ex(self.h, C.byref(a), C.byref(b), C.byref(c), 1 if all_classes else 0), self.h)
        if all_classes and self.logical_classes != self.num_classes:      # the padding classes never occur: they must not count as absent ones
            b = C.c_double(b.value * self.num_classes / self.logical_classes)
        return float(a.value), float(b.value), float(c.value)

    def augment(self, images, labels=None, out_hw=None, offsets=None, flips=None, gains=None, void_class_id=0):
        """GPU-side augmentation of a uint8 batch already on the device (not in the reference, whose BatchGenerator augments on
        the host: batch_generator.py:293-379).  images: uint8 cuda tensor [N,H,W,3]; labels: uint8 cuda tensor [N,H,W] or None.
        Per image: `offsets[n] = (y, x)` top-left corner of the output window in the source (negative = place the source inside a
        larger canvas, like `random_crop` does), `flips[n]` horizontal flip, `gains[n]` brightness factor (None / NaN = no brightness step
        for that image; a factor, 1.0 included, runs the reference's 8-bit HSV round trip, which is not the identity).  Bit-exact with
        the host BatchGenerator (cv2_compat.py).  Returns the augmented (images, labels) as new cuda tensors of size `out_hw`."""
        torch = self.torch
        self._sync_stream()
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[-1] != 3 or not images.is_cuda:
            raise ValueError("`images` must be a uint8 cuda tensor of shape (N, H, W, 3)")
        N, H, W = (int(x) for x in images.shape[:3])
        Ho, Wo = (int(x) for x in (out_hw or (H, W)))
        par = np.zeros((N, 4), np.int32)
        if offsets is not None:
            par[:, :2] = np.asarray(offsets, np.int32).reshape(N, 2)
        if flips is not None:
            par[:, 2] = np.asarray(flips).astype(np.int32).reshape(N)
        vl = None
        if gains is not None:
            g = np.asarray([np.nan if x is None else x for x in gains], np.float64).reshape(N)
            par[:, 3] = ~np.isnan(g)
            v = np.arange(256, dtype=np.uint8)[None, :] * np.where(np.isnan(g), 1.0, g)[:, None]      # the reference's float64 `hsv[:,:,2] * random_br`
            vl = torch.from_numpy(np.where(v > 255, 255, v).astype(np.uint8)).to(self.device)         # ... saturated, truncated on the uint8 store
        pd = torch.from_numpy(par).to(self.device)
        images = images.contiguous()
        out = torch.empty((N, Ho, Wo, 3), dtype=torch.uint8, device=self.device)
        lab_out = None; lp = lo = None
        if labels is not None:
            if labels.dtype != torch.uint8 or tuple(labels.shape) != (N, H, W) or not labels.is_cuda:
                raise ValueError("`labels` must be a uint8 cuda tensor of shape (N, H, W)")
            labels = labels.contiguous()
            lab_out = torch.empty((N, Ho, Wo), dtype=torch.uint8, device=self.device)
            lp, lo = C.c_void_p(labels.data_ptr()), C.c_void_p(lab_out.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        L.check(L.lib.fcn8s_op_augment_u8(stream, C.c_void_p(images.data_ptr()), lp, C.c_void_p(out.data_ptr()), lo,
                                          C.c_void_p(pd.data_ptr()), C.c_void_p(vl.data_ptr()) if vl is not None else None,
                                          N, H, W, Ho, Wo, int(void_class_id)))
        return out, lab_out

    def resample(self, images, labels=None, out_hw=None, sizes=None, offsets=None, void_class_id=0):
        """GPU-side resize / scale / translate of a uint8 batch already on the device (fcn8s_op_resample_u8; the reference's
        BatchGenerator does these on the host, batch_generator.py:328-384).  images: uint8 cuda tensor [N,H,W,3]; labels: uint8 cuda
        tensor [N,H,W] or None.  Per image: `sizes[n] = (rh, rw)` size the source is resized to (default: out_hw), `offsets[n] =
        (oy, ox)` where its top-left corner lands in the output (negative = crop).  Images follow cv2.resize(INTER_LINEAR), labels
        cv2.resize(INTER_NEAREST), bit-exact with the host BatchGenerator (cv2_compat.py).  Returns new (images, labels) of size out_hw."""
        torch = self.torch
        self._sync_stream()
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[-1] != 3 or not images.is_cuda:
            raise ValueError("`images` must be a uint8 cuda tensor of shape (N, H, W, 3)")
        N, H, W = (int(x) for x in images.shape[:3])
        Ho, Wo = (int(x) for x in (out_hw or (H, W)))
        par = np.zeros((N, 4), np.int32)
        par[:, :2] = np.asarray(sizes, np.int32).reshape(N, 2) if sizes is not None else (Ho, Wo)
        if offsets is not None:
            par[:, 2:] = np.asarray(offsets, np.int32).reshape(N, 2)
        pd = torch.from_numpy(par).to(self.device)
        images = images.contiguous()
        out = torch.empty((N, Ho, Wo, 3), dtype=torch.uint8, device=self.device)
        lab_out = None; lp = lo = None
        if labels is not None:
            if labels.dtype != torch.uint8 or tuple(labels.shape) != (N, H, W) or not labels.is_cuda:
                raise ValueError("`labels` must be a uint8 cuda tensor of shape (N, H, W)")
            labels = labels.contiguous()
            lab_out = torch.empty((N, Ho, Wo), dtype=torch.uint8, device=self.device)
            lp, lo = C.c_void_p(labels.data_ptr()), C.c_void_p(lab_out.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        L.check(L.lib.fcn8s_op_resample_u8(stream, C.c_void_p(images.data_ptr()), lp, C.c_void_p(out.data_ptr()), lo,
                                           C.c_void_p(pd.data_ptr()), N, H, W, Ho, Wo, int(void_class_id)))
        return out, lab_out

    def predict(self, images, argmax=True):
        """sess.run(predictions_argmax | softmax_output) (fcn8s_tensorflow.py:764-770)."""
        self._sync_stream()
        if isinstance(images, Staged):
            ka_i, pi, dt, _, where, nhw = self._inputs(images, None)
        else:
            ka_i, pi, dt, where, nhw = self._images(images)
        N, H, W = (int(x) for x in nhw)
        torch = self.torch
        if where == L.DEVICE:
            out = torch.empty((N, H, W), dtype=torch.int64, device=self.device) if argmax else \
                torch.empty((N, H, W, self.num_classes), dtype=torch.float32, device=self.device)
            L.check(L.lib.fcn8s_predict(self.h, pi, dt, N, H, W, int(bool(argmax)), C.c_void_p(out.data_ptr()), where), self.h)
            self._release(ka_i)
            return out if argmax or self.logical_classes == self.num_classes else out[..., :self.logical_classes].contiguous()
        out = np.empty((N, H, W), np.int64) if argmax else np.empty((N, H, W, self.num_classes), np.float32)
        L.check(L.lib.fcn8s_predict(self.h, pi, dt, N, H, W, int(bool(argmax)), out.ctypes.data_as(C.c_void_p), where), self.h)
        return out if argmax or self.logical_classes == self.num_classes else np.ascontiguousarray(out[..., :self.logical_classes])

    # ---- introspection (tests, bench) -----------------------------------------------------------
    def activation(self, name, shape, missing_ok=False):
        """fcn8s_get_activation.  A conv whose output transform was fused with the next conv's input transform (option `fuse_out_in`)
        has no activation tensor; the library says so (Fcn8sError), or, with `missing_ok`, this returns None."""
        a = np.empty(shape, np.float32)
        rc = L.lib.fcn8s_get_activation(self.h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size)
        if rc == L.ERR_STATE and missing_ok and b"not materialised" in (L.lib.fcn8s_last_error(self.h) or b""):
            return None
        L.check(rc, self.h)
        return a

    def relu_branches(self, nhw):
        """Which ReLU units were on in the last forward pass: {name: bool array (N,h,w,c)} for the convs that feed another conv, each
        block's pool (a block's last conv is never materialised; max(relu(z)) > 0 says the same for the window's maximum), fc6 and fc7
        (with keep_prob 1).  Parity tests hand these to the checker so that both sides differentiate along the same branches."""
        N, H, W = (int(x) for x in nhw)
        out = OrderedDict()
        h, w = H, W
        for b, nconv in enumerate((2, 2, 3, 3, 3), start=1):
            for i in range(1, nconv):
                name = "conv%d_%d" % (b, i)
                try:
                    out[name] = self.activation(name, (N, h, w, self.widths[b - 1])) > 0
                except L.Fcn8sError:          # not materialised (its output transform wrote the next conv's input transform directly): the ReLU record
                    rec = np.empty((N, h, w, self.widths[b - 1]), np.uint8)
                    L.check(L.lib.fcn8s_get_relu_record(self.h, name.encode(), rec.ctypes.data_as(C.c_void_p), rec.size), self.h)
                    out[name] = rec.astype(bool)
            h //= 2; w //= 2
            out["pool%d" % b] = self.activation("pool%d" % b, (N, h, w, self.widths[b - 1])) > 0
        out["fc6"] = self.activation("fc6", (N, h, w, self.widths[5])) > 0
        out["fc7"] = self.activation("fc7", (N, h, w, self.widths[6])) > 0
        return out

    def pool_routes(self, nhw):
        """Where the last training forward/backward pass routes each max-pool gradient: {"pool<b>": uint8 (N,h/2,w/2,c)}, 0..3 = window
        element (2*row + col), 4 = ReLU off (include/fcn8s_hip.h: fcn8s_get_pool_routing).  For the parity checker."""
        N, H, W = (int(x) for x in nhw)
        out = OrderedDict()
        for b in range(1, 6):
            a = np.empty((N, H >> b, W >> b, self.widths[b - 1]), np.uint8)
            L.check(L.lib.fcn8s_get_pool_routing(self.h, b, a.ctypes.data_as(C.c_void_p), a.size), self.h)
            out["pool%d" % b] = a
        return out

    def dropout_masks(self, shape6, shape7):
        m6 = np.empty(shape6, np.float32); m7 = np.empty(shape7, np.float32)
        L.check(L.lib.fcn8s_get_dropout_masks(self.h, m6.ctypes.data_as(C.c_void_p), m6.size, m7.ctypes.data_as(C.c_void_p), m7.size), self.h)
        return m6, m7

    def profile(self, on=True):
        L.check(L.lib.fcn8s_profile_enable(self.h, int(on)), self.h)     # 2 = per-layer groups

    def profile_reset(self):
        L.check(L.lib.fcn8s_profile_reset(self.h), self.h)

    def profile_results(self):
        out = OrderedDict()
        for g in range(L.lib.fcn8s_profile_num_groups(self.h)):
            name = C.c_char_p(); ms = C.c_double(); n = C.c_int64(); fl = C.c_double(); by = C.c_double()
            L.check(L.lib.fcn8s_profile_get(self.h, g, C.byref(name), C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), self.h)
            out[name.value.decode()] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
        return out

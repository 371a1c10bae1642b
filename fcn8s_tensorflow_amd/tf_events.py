"""TensorBoard event files without TensorFlow: what the reference's `tf.summary.FileWriter`s put on disk
(fcn8s_tensorflow.py:324-369 `_build_summary_ops`, :531-535 the two writers, :563 / :607 `add_summary`).

Format (restated from the published definitions -- tensorflow/core/lib/io/record_writer, tensorflow/core/util/event.proto,
tensorflow/core/framework/summary.proto; TF itself is not available offline):

  file  `events.out.tfevents.<unix seconds>.<hostname>` = a sequence of TFRecords
  TFRecord = uint64 length | uint32 masked_crc32c(length bytes) | data | uint32 masked_crc32c(data)      (little endian)
  data     = Event { double wall_time = 1; int64 step = 2; oneof { string file_version = 3; Summary summary = 5; } }
  Summary  = repeated Value value = 1;   Value { string tag = 1; oneof { float simple_value = 2; HistogramProto histo = 5; } }
  HistogramProto { double min = 1, max = 2, num = 3, sum = 4, sum_squares = 5; repeated double bucket_limit = 6 [packed], bucket = 7 [packed] }

The first record of a file is Event{file_version: "brain.Event:2"}.  Histograms use TensorFlow's default bucket limits
(tensorflow/core/lib/histogram: +-1e-12 * 1.1^k up to 1e20, 0, DBL_MAX) and its encoding (runs of empty buckets collapse into one entry).

`add_variable_summaries` mirrors helpers/tf_variable_summaries.py:3-20: mean, stddev (population), max, min, histogram of a variable,
computed on the device the variable lives on (fc6's kernel is 411 MB: only five scalars and the bucket counts travel to the host).
Tags are the ones TF 1.x generates for that code: `<scope>/mean`, `<scope>/stddev_1` (the op scope 'stddev' on line 15 takes the plain
name, so the scalar summary created after it is uniquified [INFERRED: TF name-scope rules]), `<scope>/max`, `<scope>/min`,
`<scope>/histogram`.

PARITY NOTE: no TF-written event file is available offline; this module is pinned by a reader for the same format (round trip), by
the RFC-3720 CRC-32C vectors (tf_bundle.crc32c) and by hand-assembled records in tests/test_tf_events.py.
"""
from __future__ import annotations

import os
import socket
import struct
import sys
import time

import numpy as np

from .tf_bundle import _get_varint, _mask_crc, _parse_proto, _put_varint, crc32c

_DBL_MAX = sys.float_info.max


def default_bucket_limits():
    """TensorFlow's default histogram bucket limits (upper edges, ascending; the last one is DBL_MAX)."""
    pos = []
    v = 1e-12
    while v < 1e20:
        pos.append(v)
        v *= 1.1
    return np.array([-x for x in reversed(pos)] + [0.0] + pos + [_DBL_MAX], np.float64)


_LIMITS = default_bucket_limits()


# ---- protobuf writing --------------------------------------------------------------------------------------------------------------
def _key(field, wire):
    return _put_varint((field << 3) | wire)


def _f_double(field, v):
    return _key(field, 1) + struct.pack("<d", float(v))


def _f_float(field, v):
    return _key(field, 5) + struct.pack("<f", float(v))


def _f_varint(field, v):
    return _key(field, 0) + _put_varint(int(v) & 0xFFFFFFFFFFFFFFFF)


def _f_bytes(field, b):
    return _key(field, 2) + _put_varint(len(b)) + bytes(b)


def _packed_doubles(field, values):
    return _f_bytes(field, np.asarray(values, "<f8").tobytes())


def histogram_proto(hmin, hmax, num, total, sum_squares, bucket_limit, bucket):
    return (_f_double(1, hmin) + _f_double(2, hmax) + _f_double(3, num) + _f_double(4, total) + _f_double(5, sum_squares) +
            _packed_doubles(6, bucket_limit) + _packed_doubles(7, bucket))


def encode_buckets(counts):
    """TF's Histogram::EncodeToProto: every non-empty bucket, and one entry (the last limit) per run of empty ones."""
    limits, out_l, out_c = _LIMITS, [], []
    i, n = 0, len(counts)
    while i < n:
        end, c = limits[i], counts[i]
        i += 1
        if c <= 0:
            while i < n and counts[i] <= 0:
                end, c = limits[i], counts[i]
                i += 1
        out_l.append(end); out_c.append(float(c))
    return out_l, out_c


def scalar_value(tag, v):
    return _f_bytes(1, _f_bytes(1, tag.encode()) + _f_float(2, v))


def histogram_value(tag, histo_bytes):
    return _f_bytes(1, _f_bytes(1, tag.encode()) + _f_bytes(5, histo_bytes))


def event(wall_time, step=None, file_version=None, summary=None):
    b = _f_double(1, wall_time)
    if step is not None:
        b += _f_varint(2, step)
    if file_version is not None:
        b += _f_bytes(3, file_version.encode())
    if summary is not None:
        b += _f_bytes(5, summary)
    return b


def tfrecord(data):
    head = struct.pack("<Q", len(data))
    return head + struct.pack("<I", _mask_crc(crc32c(head))) + data + struct.pack("<I", _mask_crc(crc32c(data)))


# ---- variable statistics -----------------------------------------------------------------------------------------------------------
def variable_stats(t):
    """(mean, stddev, max, min, sum, sum of squares, n, bucket counts over TF's default limits) of a torch tensor or numpy array.
    Bucket b counts the values in (limit[b-1], limit[b]] ... as TF's Histogram::Add does with upper_bound: the first limit > value."""
    try:
        import torch
    except ImportError:          # pragma: no cover
        torch = None
    if torch is not None and isinstance(t, torch.Tensor):
        # in chunks of 8 M elements (64 MB as float64: fc6's 103 M weights never exist as one 0.8 GB float64 copy), every partial result
        # kept on the tensor's device and fetched in ONE transfer at the end (one host sync per variable instead of five)
        flat = t.detach().reshape(-1)
        lim = torch.from_numpy(_LIMITS).to(flat.device)
        n = flat.numel()
        counts_t = torch.zeros(lim.numel(), dtype=torch.int64, device=flat.device)
        acc = torch.zeros(2, dtype=torch.float64, device=flat.device)
        ext = torch.tensor([-float("inf"), float("inf")], dtype=torch.float64, device=flat.device)      # running (max, min)
        for a in range(0, n, 1 << 23):
            x = flat[a:a + (1 << 23)].to(torch.float64)
            idx = torch.bucketize(x, lim, right=True).clamp_(max=lim.numel() - 1)
            counts_t += torch.bincount(idx, minlength=lim.numel())
            acc += torch.stack([x.sum(), (x * x).sum()])
            ext = torch.stack([torch.maximum(ext[0], x.max()), torch.minimum(ext[1], x.min())])
        got = torch.cat([acc, ext, counts_t.to(torch.float64)]).cpu().numpy()
        s, ss, mx, mn = float(got[0]), float(got[1]), float(got[2]), float(got[3])
        counts = got[4:].astype(np.float64)
    else:
        x = np.asarray(t, np.float64).reshape(-1)
        idx = np.minimum(np.searchsorted(_LIMITS, x, side="right"), len(_LIMITS) - 1)
        counts = np.bincount(idx, minlength=len(_LIMITS)).astype(np.float64)
        s, ss, mx, mn, n = float(x.sum()), float((x * x).sum()), float(x.max()), float(x.min()), x.size
    mean = s / n
    std = float(np.sqrt(max(ss / n - mean * mean, 0.0)))
    return dict(mean=mean, stddev=std, max=mx, min=mn, sum=s, sum_squares=ss, num=float(n), counts=counts)


def add_variable_summaries(variable, scope):
    """helpers/tf_variable_summaries.py:3-20 -> the serialized Summary.Value entries for one variable."""
    st = variable_stats(variable)
    lim, cnt = encode_buckets(st["counts"])
    histo = histogram_proto(st["min"], st["max"], st["num"], st["sum"], st["sum_squares"], lim, cnt)
    return (scalar_value(scope + "/mean", st["mean"]) + scalar_value(scope + "/stddev_1", st["stddev"]) +
            scalar_value(scope + "/max", st["max"]) + scalar_value(scope + "/min", st["min"]) +
            histogram_value(scope + "/histogram", histo))


# the ten weight / bias pairs the reference watches (fcn8s_tensorflow.py:331-350): (variable name, summary scope)
WATCHED_VARIABLES = [
    ("pool3_1x1/kernel", "pool3_1x1/kernel"), ("pool3_1x1/bias", "pool3_1x1/bias"),
    ("pool4_1x1/kernel", "pool4_1x1/kernel"), ("pool4_1x1/bias", "pool4_1x1/bias"),
    ("fc7_1x1/kernel", "fc7_1x1/kernel"), ("fc7_1x1/bias", "fc7_1x1/bias"),
    ("fc7_conv2d_trans/kernel", "fc7_conv2d_trans/kernel"), ("fc7_conv2d_trans/bias", "fc7_conv2d_trans/bias"),
    ("fc7_pool4_conv2d_trans/kernel", "fc7_pool4_conv2d_trans/kernel"), ("fc7_pool4_conv2d_trans/bias", "fc7_pool4_conv2d_trans/bias"),
    ("fc7_pool4_pool3_conv2d_trans/kernel", "fc7_pool4_pool3_conv2d_trans/kernel"), ("fc7_pool4_pool3_conv2d_trans/bias", "fc7_pool4_pool3_conv2d_trans/bias"),
    ("fc7/weights", "fc7/kernel"), ("fc7/biases", "fc7/bias"),
    ("fc6/weights", "fc6/kernel"), ("fc6/biases", "fc6/bias"),
    ("conv4_3/filter", "conv4_3/kernel"), ("conv4_3/biases", "conv4_3/bias"),
    ("conv3_3/filter", "conv3_3/kernel"), ("conv3_3/biases", "conv3_3/bias"),
]


class EventFileWriter:
    """tf.summary.FileWriter(logdir): creates `<logdir>/events.out.tfevents.<time>.<host>` and appends one record per add_*()."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, "events.out.tfevents.%010d.%s" % (int(time.time()), socket.gethostname()))
        self.f = open(self.path, "ab")
        self._write(event(time.time(), file_version="brain.Event:2"))

    def _write(self, ev):
        self.f.write(tfrecord(ev))
        self.f.flush()

    def add_summary(self, summary, global_step):
        """summary: serialized Summary (concatenated Value entries from scalar_value / histogram_value / add_variable_summaries)."""
        self._write(event(time.time(), step=global_step, summary=summary))

    def add_scalars(self, global_step, **scalars):
        self.add_summary(b"".join(scalar_value(k, v) for k, v in scalars.items()), global_step)

    def close(self):
        if self.f:
            self.f.close()
            self.f = None


# ---- reading (tests, and a quick look at a run without TensorBoard) ---------------------------------------------------------------------
def read_events(path, verify_crc=True):
    """-> list of dicts {wall_time, step, file_version?, scalars: {tag: value}, histograms: {tag: {min,max,num,sum,sum_squares,bucket_limit,bucket}}}"""
    out = []
    with open(path, "rb") as f:
        blob = f.read()
    pos = 0
    while pos < len(blob):
        head = blob[pos:pos + 8]
        (n,) = struct.unpack("<Q", head)
        (c1,) = struct.unpack("<I", blob[pos + 8:pos + 12])
        data = blob[pos + 12:pos + 12 + n]
        (c2,) = struct.unpack("<I", blob[pos + 12 + n:pos + 16 + n])
        if verify_crc and (c1 != _mask_crc(crc32c(head)) or c2 != _mask_crc(crc32c(data))):
            raise ValueError("corrupt record at byte %d of %s" % (pos, path))
        pos += 16 + n
        ev = {"scalars": {}, "histograms": {}, "step": 0}
        for field, wire, val in _parse_proto(data):
            if field == 1:
                ev["wall_time"] = struct.unpack("<d", val)[0] if isinstance(val, (bytes, bytearray)) else struct.unpack("<d", struct.pack("<Q", val))[0]
            elif field == 2:
                ev["step"] = val
            elif field == 3:
                ev["file_version"] = bytes(val).decode()
            elif field == 5:
                for f2, _, v2 in _parse_proto(val):
                    if f2 != 1:
                        continue
                    tag, simple, histo = None, None, None
                    for f3, w3, v3 in _parse_proto(v2):
                        if f3 == 1:
                            tag = bytes(v3).decode()
                        elif f3 == 2:
                            simple = struct.unpack("<f", v3)[0] if isinstance(v3, (bytes, bytearray)) else struct.unpack("<f", struct.pack("<I", v3))[0]
                        elif f3 == 5:
                            h = {}
                            names = {1: "min", 2: "max", 3: "num", 4: "sum", 5: "sum_squares"}
                            for f4, w4, v4 in _parse_proto(v3):
                                if f4 in names:
                                    h[names[f4]] = struct.unpack("<d", v4)[0] if isinstance(v4, (bytes, bytearray)) else struct.unpack("<d", struct.pack("<Q", v4))[0]
                                elif f4 == 6:
                                    h["bucket_limit"] = np.frombuffer(bytes(v4), "<f8").copy()
                                elif f4 == 7:
                                    h["bucket"] = np.frombuffer(bytes(v4), "<f8").copy()
                            histo = h
                    if simple is not None:
                        ev["scalars"][tag] = simple
                    if histo is not None:
                        ev["histograms"][tag] = histo
        out.append(ev)
    return out

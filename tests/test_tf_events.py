"""TensorBoard event files and the video helper (fcn8s_tensorflow.py:324-369, helpers/tf_variable_summaries.py:3-20,
helpers/visualization_utils.py:102-120).  CPU only.  No TF-written event file exists offline: the writer is pinned by hand-assembled
bytes (protobuf wire format + TFRecord framing + CRC-32C masks restated here independently), by the reader, and by numpy statistics."""
import os
import struct

import numpy as np
import pytest

from fcn8s_tensorflow_amd import tf_events as tfe
from fcn8s_tensorflow_amd.tf_bundle import crc32c


def _mask(c):
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


def test_record_framing_and_scalar_event_bytes_by_hand():
    # Event{wall_time = 2.0 (field 1, fixed64), step = 300 (field 2, varint), summary (field 5) { value (1) { tag (1) "a/b", simple_value (2) 0.5 } }}
    value = b"\x0a\x03a/b" + b"\x15" + struct.pack("<f", 0.5)
    summary = b"\x0a" + bytes([len(value)]) + value
    want = b"\x09" + struct.pack("<d", 2.0) + b"\x10\xac\x02" + b"\x2a" + bytes([len(summary)]) + summary
    got = tfe.event(2.0, step=300, summary=tfe.scalar_value("a/b", 0.5))
    assert got == want
    rec = tfe.tfrecord(want)
    head = struct.pack("<Q", len(want))
    assert rec == head + struct.pack("<I", _mask(crc32c(head))) + want + struct.pack("<I", _mask(crc32c(want)))
    assert crc32c(b"123456789") == 0xE3069283                       # the CRC-32C check value (RFC 3720 appendix B.4)


def test_default_buckets_and_run_length_encoding():
    lim = tfe.default_bucket_limits()
    assert lim[-1] == np.finfo(np.float64).max and (np.diff(lim) > 0).all()
    zero = int(np.where(lim == 0.0)[0][0])
    assert lim[zero + 1] == 1e-12 and lim[zero - 1] == -1e-12
    assert abs(lim[zero + 2] / lim[zero + 1] - 1.1) < 1e-12 and lim[-2] < 1e20 <= lim[-2] * 1.1
    assert len(lim) == 2 * (zero) + 2                                  # symmetric: negatives, 0, positives, DBL_MAX
    counts = np.zeros(len(lim)); counts[5] = 2; counts[6] = 1; counts[20] = 7
    l, c = tfe.encode_buckets(counts)
    # empty run [0..4] -> one entry (limit 4, 0), then the two full buckets, empty run [7..19] -> (limit 19, 0), the full one, the empty tail
    assert c == [0.0, 2.0, 1.0, 0.0, 7.0, 0.0]
    assert l == [lim[4], lim[5], lim[6], lim[19], lim[20], lim[-1]]


def test_variable_summaries_round_trip(tmp_path):
    import torch
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((3, 3, 16, 8)) * 0.01).astype(np.float32)
    w[0, 0, 0, 0] = 0.0
    st_np, st_t = tfe.variable_stats(w), tfe.variable_stats(torch.from_numpy(w))
    for k in ("mean", "stddev", "max", "min", "sum", "sum_squares", "num"):
        assert abs(st_np[k] - st_t[k]) <= 1e-12 * max(1.0, abs(st_np[k])), k
    np.testing.assert_array_equal(st_np["counts"], st_t["counts"])
    x = w.astype(np.float64)
    assert abs(st_np["mean"] - x.mean()) < 1e-15 and abs(st_np["stddev"] - x.std()) < 1e-12        # population stddev, as tf.sqrt(reduce_mean(square(v - mean)))
    lim = tfe.default_bucket_limits()
    assert st_np["counts"].sum() == w.size
    b = int(np.searchsorted(lim, 0.0, side="right"))                    # TF's upper_bound: 0.0 falls into the bucket whose limit is 1e-12
    assert lim[b] == 1e-12 and st_np["counts"][b] == 1
    v = float(x.max())
    assert st_np["counts"][int(np.searchsorted(lim, v, side="right"))] >= 1

    wr = tfe.EventFileWriter(str(tmp_path / "run"))
    wr.add_summary(tfe.add_variable_summaries(torch.from_numpy(w), "conv3_3/kernel") + tfe.scalar_value("total_loss", 1.25), 10)
    wr.add_scalars(20, mean_loss=0.5, mean_iou=0.25, accuracy=0.75)
    wr.close()
    assert os.path.basename(wr.path).startswith("events.out.tfevents.")
    evs = tfe.read_events(wr.path)
    assert evs[0]["file_version"] == "brain.Event:2" and len(evs) == 3
    e = evs[1]
    assert e["step"] == 10 and e["scalars"]["total_loss"] == 1.25
    assert set(e["scalars"]) == {"conv3_3/kernel/mean", "conv3_3/kernel/stddev_1", "conv3_3/kernel/max", "conv3_3/kernel/min", "total_loss"}
    assert abs(e["scalars"]["conv3_3/kernel/mean"] - x.mean()) < 1e-9 and e["scalars"]["conv3_3/kernel/max"] == np.float32(x.max())
    h = e["histograms"]["conv3_3/kernel/histogram"]
    assert h["num"] == w.size and h["min"] == x.min() and h["max"] == x.max() and abs(h["sum"] - x.sum()) < 1e-9
    assert h["bucket"].sum() == w.size and len(h["bucket"]) == len(h["bucket_limit"]) and (np.diff(h["bucket_limit"]) > 0).all()
    assert evs[2]["scalars"] == {"mean_loss": 0.5, "mean_iou": 0.25, "accuracy": 0.75} and evs[2]["step"] == 20
    # a flipped byte is caught
    blob = bytearray(open(wr.path, "rb").read()); blob[40] ^= 1
    open(wr.path, "wb").write(bytes(blob))
    with pytest.raises(ValueError):
        tfe.read_events(wr.path)


def test_watched_variables_are_the_references_ten_pairs():
    from fcn8s_tensorflow_amd import dp
    specs, _, _ = dp.layout(20)
    assert len(tfe.WATCHED_VARIABLES) == 20
    for name, scope in tfe.WATCHED_VARIABLES:
        assert name in specs, name
    assert dict(tfe.WATCHED_VARIABLES)["fc6/weights"] == "fc6/kernel" and dict(tfe.WATCHED_VARIABLES)["conv4_3/biases"] == "conv4_3/bias"


def test_video_helper_writes_a_parsable_mjpeg_avi(tmp_path):
    from PIL import Image
    import io
    from fcn8s_tensorflow_amd.fcn8s import create_video_from_images
    rng = np.random.default_rng(1)
    for i in range(3):
        Image.fromarray(rng.integers(0, 256, (24, 32, 3), dtype=np.uint8)).save(tmp_path / ("f%02d.png" % i))
    with pytest.raises(ValueError):
        create_video_from_images(str(tmp_path / "v"), str(tmp_path), image_file_extension="jpg")
    out = create_video_from_images(str(tmp_path / "v"), str(tmp_path), frame_rate=25.0)
    blob = open(out, "rb").read()
    if out.endswith(".mp4"):
        return                                                         # moviepy present: nothing of ours to parse
    assert blob[:4] == b"RIFF" and blob[8:12] == b"AVI " and struct.unpack("<I", blob[4:8])[0] == len(blob) - 8
    p = blob.index(b"avih") + 8
    usec, _, _, flags, nframes, _, nstreams, _, w, h = struct.unpack("<10I", blob[p:p + 40])
    assert usec == 40000 and nframes == 3 and nstreams == 1 and (w, h) == (32, 24) and flags & 0x10
    movi = blob.index(b"movi")
    idx = blob.index(b"idx1", movi)
    n_idx = struct.unpack("<I", blob[idx + 4:idx + 8])[0] // 16
    assert n_idx == 3
    for k in range(3):
        cid, fl, off, ln = struct.unpack("<4sIII", blob[idx + 8 + 16 * k:idx + 24 + 16 * k])
        assert cid == b"00dc"
        c = movi + off
        assert blob[c:c + 4] == b"00dc" and struct.unpack("<I", blob[c + 4:c + 8])[0] == ln
        im = Image.open(io.BytesIO(blob[c + 8:c + 8 + ln]))
        assert im.format == "JPEG" and im.size == (32, 24)

"""TensorFlow tensor-bundle reader/writer (the on-disk format of the reference's VGG-16 SavedModel and of
its own checkpoints).  No TF-written fixture exists offline, so the format is pinned by hand-assembled
LevelDB blocks / protobuf bytes, known CRC-32C vectors and round trips.  CPU only."""
import os
import struct

import numpy as np
import pytest

from fcn8s_tensorflow_amd import tf_bundle as tb


def test_crc32c_known_vectors():
    assert tb.crc32c(b"") == 0
    assert tb.crc32c(b"123456789") == 0xE3069283                 # the standard CRC-32C check value
    assert tb.crc32c(bytes(32)) == 0x8A9136AA                    # RFC 3720 B.4: 32 bytes of zeros
    assert tb.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43           # RFC 3720 B.4: 32 bytes of ones
    big = np.random.default_rng(0).integers(0, 256, 100003, dtype=np.uint8).tobytes()
    c_fast = tb.crc32c(big)                                      # C slice-by-8 path (>= 4096 bytes)
    c_slow = tb.crc32c(big[:2000], 0); c_slow = tb.crc32c(big[2000:4000], c_slow)   # pure-Python path, chained
    from fcn8s_tensorflow_amd import _lib
    buf = np.frombuffer(big, np.uint8)
    assert int(_lib.lib.fcn8s_crc32c(buf.ctypes.data, 4000, 0)) == c_slow
    assert int(_lib.lib.fcn8s_crc32c(buf[4000:].ctypes.data, buf.size - 4000, c_slow)) == c_fast
    assert tb._mask_crc(0) == 0xa282ead8                         # LevelDB mask of 0


def test_varint_and_entry_proto_bytes():
    assert tb._put_varint(300) == b"\xac\x02" and tb._get_varint(b"\xac\x02", 0) == (300, 2)
    # BundleEntryProto{dtype: DT_FLOAT(1), shape{dim{size:3} dim{size:64}}, offset: 1024, size: 768, crc32c: 0xDEADBEEF}
    raw = (b"\x08\x01" + b"\x12\x08" + b"\x12\x02\x08\x03" + b"\x12\x02\x08\x40" + b"\x20\x80\x08" + b"\x28\x80\x06"
           + b"\x35" + struct.pack("<I", 0xDEADBEEF))
    e = tb._parse_entry(raw)
    assert e["dtype"] == 1 and e["shape"] == [3, 64] and e["offset"] == 1024 and e["size"] == 768 and e["crc32c"] == 0xDEADBEEF
    assert tb._entry_proto(1, (3, 64), 1024, 768, 0xDEADBEEF) == raw


def test_hand_assembled_table_is_read(tmp_path):
    """A minimal LevelDB table written byte by byte here (one data block with prefix-compressed keys)."""
    t = np.arange(6, dtype=np.float32).reshape(2, 3)
    data = t.tobytes()
    (tmp_path / "ck.data-00000-of-00001").write_bytes(data)
    header = b"\x08\x01\x1a\x02\x08\x01"
    ent = tb._entry_proto(1, (2, 3), 0, len(data), tb._mask_crc(tb.crc32c(data)))
    ent2 = tb._entry_proto(3, (), 0, 0, 0)                       # a scalar int32 of size 0 is never read back below
    # entries: "" , "conv1_1/biases" (shared 0), "conv1_1/filter" (shared 8 = len("conv1_1/"))
    blk = (b"\x00\x00" + bytes([len(header)]) + header
           + b"\x00\x0e" + bytes([len(ent2)]) + b"conv1_1/biases" + ent2
           + b"\x08\x06" + bytes([len(ent)]) + b"filter" + ent
           + struct.pack("<II", 0, 1))
    def framed(b):
        return b + b"\x00" + struct.pack("<I", tb._mask_crc(tb.crc32c(b + b"\x00")))
    meta = struct.pack("<II", 0, 1)
    idx_entry_val = tb._put_varint(0) + tb._put_varint(len(blk))
    idx = b"\x00\x01" + bytes([len(idx_entry_val)]) + b"d" + idx_entry_val + struct.pack("<II", 0, 1)
    body = framed(blk)
    moff = len(body); body += framed(meta)
    ioff = len(body); body += framed(idx)
    footer = tb._put_varint(moff) + tb._put_varint(len(meta)) + tb._put_varint(ioff) + tb._put_varint(len(idx))
    body += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57)
    (tmp_path / "ck.index").write_bytes(body)
    hdr, entries = tb.read_index(str(tmp_path / "ck.index"))
    assert hdr["num_shards"] == 1 and list(entries) == ["conv1_1/biases", "conv1_1/filter"]
    out = tb.read_bundle(str(tmp_path / "ck"), names={"conv1_1/filter"}, verify_crc=True)
    np.testing.assert_array_equal(out["conv1_1/filter"], t)


def test_round_trip_many_tensors(tmp_path):
    rng = np.random.default_rng(1)
    tensors = {"v%03d/weights" % i: rng.standard_normal((3, 3, 4, 5)).astype(np.float32) for i in range(150)}   # > one data block
    tensors["optimizer/global_step"] = np.asarray(1234, dtype=np.int32)
    tensors["big"] = rng.standard_normal(70000).astype(np.float32)
    tensors["flags"] = np.array([True, False, True])
    tb.write_bundle(str(tmp_path / "variables" / "variables"), tensors)
    assert tb.find_bundle_prefix(str(tmp_path)) == str(tmp_path / "variables" / "variables")
    back = tb.read_bundle(tb.find_bundle_prefix(str(tmp_path)), verify_crc=True)
    assert sorted(back) == sorted(tensors)
    for k in tensors:
        np.testing.assert_array_equal(back[k], tensors[k]); assert back[k].dtype == tensors[k].dtype
    raw = bytearray((tmp_path / "variables" / "variables.data-00000-of-00001").read_bytes()); raw[10] ^= 1
    (tmp_path / "variables" / "variables.data-00000-of-00001").write_bytes(bytes(raw))
    with pytest.raises(ValueError, match="crc32c"):
        tb.read_bundle(str(tmp_path / "variables" / "variables"), verify_crc=True)
    with pytest.raises(ValueError, match="magic"):
        (tmp_path / "bad.index").write_bytes(b"x" * 100); tb.read_index(str(tmp_path / "bad.index"))

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


def _framed(b):
    return b + b"\x00" + struct.pack("<I", tb._mask_crc(tb.crc32c(b + b"\x00")))


def _table(entries):
    """A LevelDB table assembled byte by byte: ONE data block, no key prefix sharing (shared = 0 for every entry), one restart point."""
    blk = b""
    for key, val in entries:
        assert len(key) < 128 and len(val) < 128
        blk += b"\x00" + bytes([len(key)]) + bytes([len(val)]) + key + val
    blk += struct.pack("<II", 0, 1)
    meta = struct.pack("<II", 0, 1)
    hv = tb._put_varint(0) + tb._put_varint(len(blk))
    idx = b"\x00\x01" + bytes([len(hv)]) + b"\xff" + hv + struct.pack("<II", 0, 1)
    body = _framed(blk)
    moff = len(body); body += _framed(meta)
    ioff = len(body); body += _framed(idx)
    footer = tb._put_varint(moff) + tb._put_varint(len(meta)) + tb._put_varint(ioff) + tb._put_varint(len(idx))
    return body + footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57)


def test_hand_assembled_two_shard_bundle_is_read(tmp_path):
    """BundleHeaderProto{num_shards: 2}; "a" lives in shard 1 at offset 8, "b" in shard 0 at offset 0 -- entry protos written out byte by byte
    (field 3 = shard_id, field 4 = offset).  A missing shard file and a shard id outside the header's count are reported as such."""
    a = np.array([1.5, -2.0, 3.25], np.float32); b = np.array([[7, 8], [9, 10]], np.int32)
    (tmp_path / "m.data-00000-of-00002").write_bytes(b.tobytes())
    (tmp_path / "m.data-00001-of-00002").write_bytes(b"\xAA" * 8 + a.tobytes())
    crc = lambda x: struct.pack("<I", tb._mask_crc(tb.crc32c(x.tobytes())))
    header = b"\x08\x02" + b"\x1a\x02\x08\x01"
    ent_a = b"\x08\x01" + b"\x12\x04\x12\x02\x08\x03" + b"\x18\x01" + b"\x20\x08" + b"\x28\x0c" + b"\x35" + crc(a)            # DT_FLOAT [3], shard 1, offset 8, 12 bytes
    ent_b = b"\x08\x03" + b"\x12\x08\x12\x02\x08\x02\x12\x02\x08\x02" + b"\x28\x10" + b"\x35" + crc(b)                          # DT_INT32 [2,2], shard 0 (default), offset 0, 16 bytes
    (tmp_path / "m.index").write_bytes(_table([(b"", header), (b"a", ent_a), (b"b", ent_b)]))
    hdr, entries = tb.read_index(str(tmp_path / "m.index"))
    assert hdr["num_shards"] == 2 and entries["a"]["shard_id"] == 1 and entries["a"]["offset"] == 8 and entries["b"]["shard_id"] == 0
    out = tb.read_bundle(str(tmp_path / "m"), verify_crc=True)
    np.testing.assert_array_equal(out["a"], a); np.testing.assert_array_equal(out["b"], b)
    assert out["a"].dtype == np.float32 and out["b"].dtype == np.int32
    os.remove(tmp_path / "m.data-00001-of-00002")
    with pytest.raises(FileNotFoundError, match="data-00001-of-00002"):
        tb.read_bundle(str(tmp_path / "m"))
    ent_bad = ent_a.replace(b"\x18\x01", b"\x18\x05")
    (tmp_path / "m.index").write_bytes(_table([(b"", header), (b"a", ent_bad), (b"b", ent_b)]))
    with pytest.raises(ValueError, match="shard 5"):
        tb.read_bundle(str(tmp_path / "m"))


def test_ordered_code_vectors_and_slice_key():
    """tensorflow/core/lib/strings/ordered_code: the published examples of the signed encoding (0 -> 80, -1 -> 7f, 63 -> bf, 64 -> c0 40,
    -64 -> 40, -65 -> 3f bf, 8191 -> df ff, 8192 -> e0 20 00) and the key checkpoint::EncodeTensorNameSlice builds from them."""
    enc = tb._oc_signed_num_increasing
    assert enc(0) == b"\x80" and enc(-1) == b"\x7f" and enc(63) == b"\xbf" and enc(-64) == b"\x40"
    assert enc(64) == b"\xc0\x40" and enc(-65) == b"\x3f\xbf" and enc(8191) == b"\xdf\xff" and enc(8192) == b"\xe0\x20\x00" and enc(-8193) == b"\x1f\xdf\xff"
    assert [enc(v) for v in sorted([-70000, -65, -64, -1, 0, 5, 63, 64, 8191, 8192, 1 << 40])] == sorted(enc(v) for v in [-70000, -65, -64, -1, 0, 5, 63, 64, 8191, 8192, 1 << 40])
    assert tb._oc_num_increasing(0) == b"\x00" and tb._oc_num_increasing(2) == b"\x01\x02" and tb._oc_num_increasing(300) == b"\x02\x01\x2c"
    assert tb._oc_string(b"w\x00\xff") == b"w\x00\xff\xff\x00\x00\x01"
    # name "v", rank 2, rows [2, 2+3), all columns
    assert tb.slice_key("v", [(2, 3), (0, -1)]) == b"\x00" + b"v\x00\x01" + b"\x01\x02" + b"\x82\x83" + b"\x80\x7f"


def test_hand_assembled_sliced_entry_is_read(tmp_path):
    """A partitioned variable "v" of shape [5, 2] stored as two slices (rows 0..1 and rows 2..4), every byte written here: the full-shape
    entry carries two TensorSliceProtos (field 7) and no bytes; each slice's bytes sit under its EncodeTensorNameSlice key.  A bundle whose
    slices leave rows uncovered, and one whose slice key is absent, are refused."""
    v = np.arange(10, dtype=np.float32).reshape(5, 2)
    top, bot = v[:2], v[2:]
    (tmp_path / "p.data-00000-of-00001").write_bytes(top.tobytes() + bot.tobytes())
    crc = lambda x: struct.pack("<I", tb._mask_crc(tb.crc32c(np.ascontiguousarray(x).tobytes())))
    header = b"\x08\x01" + b"\x1a\x02\x08\x01"
    shape52 = b"\x12\x08\x12\x02\x08\x05\x12\x02\x08\x02"
    # TensorSliceProto: extent{length: 2} extent{}   (start 0 is omitted; an empty Extent = the whole dimension)
    sl_top = b"\x0a\x02\x10\x02" + b"\x0a\x00"
    # extent{start: 2, length: 3} extent{}
    sl_bot = b"\x0a\x04\x08\x02\x10\x03" + b"\x0a\x00"
    full = b"\x08\x01" + shape52 + b"\x3a" + bytes([len(sl_top)]) + sl_top + b"\x3a" + bytes([len(sl_bot)]) + sl_bot
    e_top = b"\x08\x01" + b"\x12\x08\x12\x02\x08\x02\x12\x02\x08\x02" + b"\x28\x10" + b"\x35" + crc(top)                         # [2,2] at offset 0
    e_bot = b"\x08\x01" + b"\x12\x08\x12\x02\x08\x03\x12\x02\x08\x02" + b"\x20\x10" + b"\x28\x18" + b"\x35" + crc(bot)          # [3,2] at offset 16
    k_top = b"\x00" + b"v\x00\x01" + b"\x01\x02" + b"\x80\x82" + b"\x80\x7f"
    k_bot = b"\x00" + b"v\x00\x01" + b"\x01\x02" + b"\x82\x83" + b"\x80\x7f"
    (tmp_path / "p.index").write_bytes(_table([(b"", header), (k_top, e_top), (k_bot, e_bot), (b"v", full)]))
    hdr, entries = tb.read_index(str(tmp_path / "p.index"))
    assert list(entries) == ["v"] and entries["v"]["sliced"] and entries["v"]["slices"] == [[(0, 2), (0, -1)], [(2, 3), (0, -1)]]
    out = tb.read_bundle(str(tmp_path / "p"), verify_crc=True)
    np.testing.assert_array_equal(out["v"], v)
    (tmp_path / "p.index").write_bytes(_table([(b"", header), (k_top, e_top), (b"v", full)]))
    with pytest.raises(ValueError, match="no entry in the index"):
        tb.read_bundle(str(tmp_path / "p"))
    only_top = b"\x08\x01" + shape52 + b"\x3a" + bytes([len(sl_top)]) + sl_top
    (tmp_path / "p.index").write_bytes(_table([(b"", header), (k_top, e_top), (b"v", only_top)]))
    with pytest.raises(ValueError, match="do not cover"):
        tb.read_bundle(str(tmp_path / "p"))


def test_round_trip_shards_and_partitions(tmp_path):
    rng = np.random.default_rng(2)
    tensors = {"w%02d" % i: rng.standard_normal((7, 3, 5)).astype(np.float32) for i in range(40)}
    tensors["emb"] = rng.standard_normal((1000, 16)).astype(np.float32)
    tensors["step"] = np.asarray(7, np.int64)
    tb.write_bundle(str(tmp_path / "ck"), tensors, shards=3, partition={"emb": 4, "w05": 7})
    assert sorted(f for f in os.listdir(tmp_path) if ".data-" in f) == ["ck.data-0000%d-of-00003" % i for i in range(3)]
    hdr, entries = tb.read_index(str(tmp_path / "ck.index"))
    assert hdr["num_shards"] == 3 and entries["emb"]["sliced"] and len(entries["emb"]["slices"]) == 4 and len(hdr["_slice_entries"]) == 11
    assert {e["shard_id"] for e in entries.values()} | {e["shard_id"] for e in hdr["_slice_entries"].values()} == {0, 1, 2}
    back = tb.read_bundle(str(tmp_path / "ck"), verify_crc=True)
    assert sorted(back) == sorted(tensors)
    for k in tensors:
        np.testing.assert_array_equal(back[k], tensors[k])


def test_snappy_block_is_refused_with_a_clear_message(tmp_path):
    t = np.zeros(2, np.float32)
    tb.write_bundle(str(tmp_path / "s"), {"t": t})
    raw = bytearray((tmp_path / "s.index").read_bytes())
    n = len(raw)
    # the index block's handle sits in the footer; flip the compression-type byte behind the FIRST data block (offset = its size)
    hdr, entries = tb.read_index(str(tmp_path / "s.index"))
    footer = bytes(raw[n - 48:])
    _, p = tb._get_varint(footer, 0); _, p = tb._get_varint(footer, p)
    ioff, p = tb._get_varint(footer, p); isz, _ = tb._get_varint(footer, p)
    raw[ioff + isz] = 1
    (tmp_path / "s.index").write_bytes(bytes(raw))
    with pytest.raises(NotImplementedError, match="snappy"):
        tb.read_index(str(tmp_path / "s.index"))

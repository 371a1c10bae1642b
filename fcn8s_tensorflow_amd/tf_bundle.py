"""Reader / writer for TensorFlow's checkpoint "tensor bundle" format, without TensorFlow.

This is the on-disk format on either side of the hot path: the pretrained VGG-16
SavedModel the reference loads (`<vgg16_dir>/variables/variables.{index,data-00000-of-00001}`,
fcn8s_tensorflow.py:134), the SavedModels `save(saver='saved_model')` writes (:922-925) and the
`tf.train.Saver` checkpoints of `saver='train_saver'` (:927-934) all store their variables this way.

Format (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table -- restated from the published
format, TF itself is not available offline):
  <prefix>.index                  an SSTable in LevelDB table format.  key "" -> BundleHeaderProto,
                                  key <tensor name> -> BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}
  <prefix>.data-NNNNN-of-MMMMM    raw little-endian tensor bytes at [offset, offset + size)
LevelDB table: data blocks + metaindex block + index block + 48-byte footer (two varint BlockHandles, padding,
magic 0xdb4775248b80fb57).  A block is prefix-compressed entries (varint shared, varint non_shared, varint
value_len, key delta, value), a uint32 restart array and its length, followed by a 1-byte compression type
(0 = none; snappy is not produced by TF's BundleWriter and is not supported here) and a masked crc32c.

Beyond BundleWriter's defaults: bundles merged from several writers have more than one data shard (`shard_id` picks the file), and a
partitioned variable (tf.get_variable(..., partitioner=...), or a Saver with sharded=True over one) is stored as a full-shape entry whose
field 7 lists TensorSliceProtos, the bytes of each slice sitting under a key of its own made by checkpoint::EncodeTensorNameSlice
(tensorflow/core/util/saved_tensor_slice_util.cc): OrderedCode(0) + OrderedCode string(name) + OrderedCode(rank) + per dimension the signed
ordered codes of (start, length), a full dimension being (0, -1).  Both are read here; snappy-compressed table blocks (never written by
BundleWriter) are refused with a message that says so.

PARITY NOTE: no TF-written fixture is available offline, so this module is pinned by round-trip tests and by
hand-assembled blocks only (tests/test_tf_bundle.py: one shard, two shards, a sliced entry); the first contact with a real checkpoint is untested.
"""
from __future__ import annotations

import os
import struct
from collections import OrderedDict

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


# ---- varints / minimal protobuf ------------------------------------------------------------------
def _get_varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]; pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf):
    """-> list of (field number, wire type, value) ; value is int or bytes."""
    pos, out = 0, []
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]; pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n]); pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]; pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.append((field, wt, v))
    return out


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_slice(buf):
    """TensorSliceProto { repeated Extent extent = 1 { int64 start = 1; oneof { int64 length = 2 } } } -> [(start, length)], length -1 = the whole dimension."""
    ext = []
    for field, _, v in _parse_proto(buf):
        if field == 1:
            start, length = 0, -1
            for f2, _, x in _parse_proto(v):
                if f2 == 1:
                    start = _signed64(x)
                elif f2 == 2:
                    length = _signed64(x)
            ext.append((start, length))
    return ext


# ---- OrderedCode (tensorflow/core/lib/strings/ordered_code.cc): the key of a tensor slice ------------------------------
_HEADER_BITS = {1: (0x80, 0), 2: (0xc0, 0), 3: (0xe0, 0), 4: (0xf0, 0), 5: (0xf8, 0), 6: (0xfc, 0), 7: (0xfe, 0), 8: (0xff, 0), 9: (0xff, 0x80), 10: (0xff, 0xc0)}


def _oc_num_increasing(v):
    body = v.to_bytes((v.bit_length() + 7) // 8, "big") if v else b""
    return bytes([len(body)]) + body


def _oc_string(b):
    return b"".join(b"\x00\xff" if c == 0 else b"\xff\x00" if c == 0xff else bytes([c]) for c in b) + b"\x00\x01"


def _oc_signed_num_increasing(v):
    x = ~v if v < 0 else v
    if x < 64:
        return bytes([(0x80 ^ v) & 0xff])
    n = x.bit_length() // 7 + 1                     # SignedEncodingLength: kBitsToLength[Log2Floor64(x) + 1]
    buf = bytearray((v & ((1 << 80) - 1)).to_bytes(10, "big"))       # sign-extended to 10 bytes, network order
    begin = 10 - n
    buf[begin] ^= _HEADER_BITS[n][0]
    buf[begin + 1] ^= _HEADER_BITS[n][1]
    return bytes(buf[begin:])


def slice_key(name, extents):
    """checkpoint::EncodeTensorNameSlice: the table key under which the bytes of one slice of a partitioned variable are stored."""
    out = _oc_num_increasing(0) + _oc_string(name.encode()) + _oc_num_increasing(len(extents))
    for start, length in extents:
        out += _oc_signed_num_increasing(start) + _oc_signed_num_increasing(length)
    return out


def _parse_entry(buf):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "sliced": False, "slices": []}
    for field, wt, v in _parse_proto(buf):
        if field == 1:
            e["dtype"] = v
        elif field == 2:                                   # TensorShapeProto { repeated Dim dim = 2 { int64 size = 1 } }
            for f2, _, dim in _parse_proto(v):
                if f2 == 2:
                    size = 0
                    for f3, _, s in _parse_proto(dim):
                        if f3 == 1:
                            size = _signed64(s)
                    e["shape"].append(size)
        elif field == 3:
            e["shard_id"] = v
        elif field == 4:
            e["offset"] = v
        elif field == 5:
            e["size"] = v
        elif field == 6:
            e["crc32c"] = v
        elif field == 7:
            e["sliced"] = True
            e["slices"].append(_parse_slice(v))
    return e


# ---- crc32c (Castagnoli), masked as in LevelDB / TF ---------------------------------------------------
_CRC_TABLE = None


def crc32c(data, crc=0):
    """CRC-32C; large buffers go through the C library's slice-by-8 routine (host code, no GPU needed)."""
    global _CRC_TABLE
    if len(data) >= 4096:
        try:
            from . import _lib
            buf = np.frombuffer(data, dtype=np.uint8)
            return int(_lib.lib.fcn8s_crc32c(buf.ctypes.data, buf.size, crc))
        except Exception:
            pass
    if _CRC_TABLE is None:
        tbl = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tbl.append(c)
        _CRC_TABLE = tbl
    c = crc ^ 0xFFFFFFFF
    for b in bytes(data):
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _mask_crc(c):
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


# ---- LevelDB table -------------------------------------------------------------------------------------
def _read_block(f, offset, size):
    f.seek(offset)
    raw = f.read(size + 5)
    if len(raw) < size + 1:
        raise ValueError("truncated table block")
    if raw[size] != 0:
        raise NotImplementedError("compressed table block (type %d%s): TF's BundleWriter writes uncompressed blocks and this reader has no "
                                  "decompressor; re-save the checkpoint with TensorFlow's own Saver" % (raw[size], ", snappy" if raw[size] == 1 else ""))
    return raw[:size]


def _block_entries(block):
    nrestarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared]); pos += non_shared
        yield key, bytes(block[pos:pos + vlen]); pos += vlen


def read_index(index_path):
    """-> (header dict, OrderedDict name -> entry dict)"""
    with open(index_path, "rb") as f:
        f.seek(0, os.SEEK_END)
        n = f.tell()
        if n < 48:
            raise ValueError("%s is too small to be a tensor-bundle index" % index_path)
        f.seek(n - 48)
        footer = f.read(48)
        if struct.unpack_from("<Q", footer, 40)[0] != _MAGIC:
            raise ValueError("%s is not a LevelDB-format table (bad magic)" % index_path)
        pos = 0
        _, pos = _get_varint(footer, pos); _, pos = _get_varint(footer, pos)        # metaindex handle
        ioff, pos = _get_varint(footer, pos); isz, pos = _get_varint(footer, pos)   # index handle
        entries = OrderedDict()
        header, slices = {}, {}
        for _, handle in _block_entries(_read_block(f, ioff, isz)):
            boff, p2 = _get_varint(handle, 0); bsz, _ = _get_varint(handle, p2)
            for key, value in _block_entries(_read_block(f, boff, bsz)):
                if key == b"":
                    for field, _, v in _parse_proto(value):
                        header[{1: "num_shards", 2: "endianness", 3: "version"}.get(field, field)] = v
                elif key[:1] == b"\x00":                     # an encoded (name, slice) key: the bytes of one slice of a partitioned variable
                    slices[bytes(key)] = _parse_entry(value)
                else:
                    entries[key.decode()] = _parse_entry(value)
    header["_slice_entries"] = slices
    return header, entries


def read_bundle(prefix, names=None, verify_crc=False):
    """Reads the variables of the bundle `<prefix>.index` / `<prefix>.data-*` -> OrderedDict name -> ndarray."""
    header, entries = read_index(prefix + ".index")
    if header.get("endianness", 0) != 0:
        raise NotImplementedError("big-endian tensor bundle")
    nshards = header.get("num_shards", 1)
    slice_entries = header.get("_slice_entries", {})
    files, out = {}, OrderedDict()

    def payload(name, e):
        sid = e["shard_id"]
        if not 0 <= sid < nshards:
            raise ValueError("'%s' lives in shard %d of a bundle with %d shard(s)" % (name, sid, nshards))
        if sid not in files:
            path = "%s.data-%05d-of-%05d" % (prefix, sid, nshards)
            if not os.path.isfile(path):
                raise FileNotFoundError("data shard %s (holding '%s') is missing" % (path, name))
            files[sid] = open(path, "rb")
        f = files[sid]
        f.seek(e["offset"])
        raw = f.read(e["size"])
        if len(raw) != e["size"]:
            raise ValueError("truncated data file for '%s'" % name)
        if verify_crc and e["crc32c"] is not None and _mask_crc(crc32c(raw)) != e["crc32c"]:
            raise ValueError("crc32c mismatch for '%s'" % name)
        return raw

    try:
        for name, e in entries.items():
            if names is not None and name not in names:
                continue
            if e["dtype"] not in _DTYPES:
                continue                                       # strings / resources: not variables of this model
            dt = np.dtype(_DTYPES[e["dtype"]]).newbyteorder("<")
            if e["sliced"]:
                # a partitioned variable: the full-shape entry lists its slices, each stored under its own encoded key
                full = np.zeros(e["shape"], dtype=_DTYPES[e["dtype"]])
                covered = np.zeros(e["shape"], dtype=bool)
                for ext in e["slices"]:
                    if len(ext) != len(e["shape"]):
                        raise ValueError("slice of '%s' has rank %d, the variable rank %d" % (name, len(ext), len(e["shape"])))
                    key = slice_key(name, ext)
                    se = slice_entries.get(key)
                    if se is None:
                        raise ValueError("slice %s of partitioned variable '%s' has no entry in the index" % (ext, name))
                    idx = tuple(slice(st, (st + ln) if ln >= 0 else dim) for (st, ln), dim in zip(ext, e["shape"]))
                    part_shape = full[idx].shape
                    full[idx] = np.frombuffer(payload(name, se), dtype=dt).reshape(part_shape)
                    covered[idx] = True
                if not covered.all():
                    raise ValueError("the slices of partitioned variable '%s' do not cover its shape %s" % (name, e["shape"]))
                out[name] = full
                continue
            out[name] = np.frombuffer(payload(name, e), dtype=dt).reshape(e["shape"]).copy()
    finally:
        for f in files.values():
            f.close()
    return out


# ---- writer (one shard, uncompressed blocks) ----------------------------------------------------------------
def _slice_proto(extents):
    out = b""
    for start, length in extents:
        ext = (b"\x08" + _put_varint(start) if start else b"") + (b"\x10" + _put_varint(length) if length >= 0 else b"")
        out += b"\x0a" + _put_varint(len(ext)) + ext
    return out


def _entry_proto(dtype_id, shape, offset, size, crc, shard_id=0, slices=None):
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(s & ((1 << 64) - 1)) for s in shape))
    out = b"\x08" + _put_varint(dtype_id)
    out += b"\x12" + _put_varint(len(dims)) + dims
    if shard_id:
        out += b"\x18" + _put_varint(shard_id)
    if offset:
        out += b"\x20" + _put_varint(offset)
    if slices is None:
        out += b"\x28" + _put_varint(size)
        out += b"\x35" + struct.pack("<I", crc)
    else:                                             # the full-shape entry of a partitioned variable carries no bytes of its own
        for ext in slices:
            sp = _slice_proto(ext)
            out += b"\x3a" + _put_varint(len(sp)) + sp
    return out


def _build_block(items, restart_interval=16):
    out, restarts, last, n = bytearray(), [], b"", 0
    for key, value in items:
        if n % restart_interval == 0:
            restarts.append(len(out)); shared = 0
        else:
            shared = 0
            while shared < min(len(last), len(key)) and last[shared] == key[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        last = key; n += 1
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_bundle(prefix, tensors, shards=1, partition=None):
    """Writes `tensors` (name -> ndarray) as a tensor bundle readable by tf.train.Saver / tf.train.load_checkpoint (variables are stored
    sorted by name, as TF does).  shards > 1 spreads the variables round-robin over that many data files (what merging the bundles of
    several writers produces); partition = {name: n} stores that variable as n slices along its first axis (a partitioned variable)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    names = sorted(tensors)
    shards = max(1, int(shards))
    items = [(b"", b"\x08" + _put_varint(shards) + b"\x1a\x02\x08\x01")]      # BundleHeaderProto{num_shards, version{producer=1}}
    offsets = [0] * shards
    files = [open("%s.data-%05d-of-%05d" % (prefix, i, shards), "wb") for i in range(shards)]
    try:
        nth = 0
        for name in names:
            a = np.ascontiguousarray(tensors[name])
            if a.dtype not in _DTYPE_IDS:
                raise ValueError("unsupported dtype %s for '%s'" % (a.dtype, name))
            parts = [(None, a)]
            nparts = (partition or {}).get(name, 0)
            if nparts:
                edges = np.linspace(0, a.shape[0], nparts + 1).astype(int)
                parts = [([(int(lo), int(hi - lo))] + [(0, -1)] * (a.ndim - 1), a[lo:hi]) for lo, hi in zip(edges[:-1], edges[1:])]
                items.append((name.encode(), _entry_proto(_DTYPE_IDS[a.dtype], a.shape, 0, 0, 0, slices=[p[0] for p in parts])))
            for ext, part in parts:
                sid = nth % shards; nth += 1
                part = np.ascontiguousarray(part)
                raw = part.astype(part.dtype.newbyteorder("<"), copy=False).tobytes()
                files[sid].write(raw)
                key = name.encode() if ext is None else slice_key(name, ext)
                items.append((key, _entry_proto(_DTYPE_IDS[a.dtype], part.shape, offsets[sid], len(raw), _mask_crc(crc32c(raw)), shard_id=sid)))
                offsets[sid] += len(raw)
    finally:
        for f in files:
            f.close()
    items = [items[0]] + sorted(items[1:], key=lambda kv: kv[0])      # table order = byte order of the keys (slice keys start with 0x00)

    def emit(f, block):
        pos = f.tell()
        trailer = b"\x00"
        f.write(block + trailer + struct.pack("<I", _mask_crc(crc32c(block + trailer))))
        return _put_varint(pos) + _put_varint(len(block))
    with open(prefix + ".index", "wb") as f:
        index_items = []
        for i in range(0, len(items), 64):                 # several data blocks, like a real (4 KiB block) table
            chunk = items[i:i + 64]
            handle = emit(f, _build_block(chunk))
            index_items.append((chunk[-1][0] + b"\x00" if i + 64 < len(items) else chunk[-1][0] + b"\xff", handle))
        meta = emit(f, _build_block([]))
        index = emit(f, _build_block(index_items, restart_interval=1))
        footer = meta + index
        f.write(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC))


# ---- mapping onto the FCN-8s variables ---------------------------------------------------------------------
ADAM_M_SUFFIX = "/adam_optimizer"        # tf.train.AdamOptimizer(name='adam_optimizer') slot names (fcn8s_tensorflow.py:256)
ADAM_V_SUFFIX = "/adam_optimizer_1"


def find_bundle_prefix(path):
    """Accepts a SavedModel dir, its variables/ dir, or a checkpoint prefix."""
    for cand in (os.path.join(path, "variables", "variables"), os.path.join(path, "variables"), path):
        if os.path.isfile(cand + ".index"):
            return cand
    return None

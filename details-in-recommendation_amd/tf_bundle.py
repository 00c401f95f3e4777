"""tf_bundle.py -- reader / writer of TensorFlow checkpoint bundles (the "V2" format of tf.train.Saver / Estimator model_dir:
`<prefix>.index` + `<prefix>.data-00000-of-00001`, and the `checkpoint` state file).

SURVEY.md 8(f) rank 4.  The reference's Estimators checkpoint into `model_dir` (models/DeepFM/deepFM.py:56,138-140;
models/DeepCrossNetwork/DeepCrossNetwork.py:36,115; train.py:170-175 exports through FinalExporter, whose SavedModel keeps its
variables in the same bundle format under variables/).  With this module a checkpoint written by the reference can be loaded
into the modules here by variable name (checkpoint.load_tf_checkpoint) and vice versa -- no TensorFlow needed on either side.

The arithmetic lives in a third-party dependency that is not vendored in /root/reference and not installable here (TensorFlow
1.x, version unpinned by the reference: no requirements file), so the format is restated from its published sources:
  * tensorflow/core/util/tensor_bundle/tensor_bundle.{h,cc} + protobuf/tensor_bundle.proto -- the index maps "" to a
    BundleHeaderProto {num_shards = 1, endianness = 2 (LITTLE = 0), version = 3 {producer = 1}} and every tensor name to a
    BundleEntryProto {dtype = 1, shape = 2, shard_id = 3, offset = 4, size = 5, crc32c = 6 (fixed32, MASKED crc32c of the tensor
    bytes), slices = 7}; tensor bytes are row-major little-endian in the data shard at [offset, offset + size);
  * tensorflow/core/lib/io/{table_builder,format,block_builder}.cc -- the index is a leveldb-format sorted table: blocks of
    prefix-compressed entries (varint32 shared, non_shared, value_len; restart array + count), each block followed by a 5-byte
    trailer (compression type 0, masked crc32c of block + type), an index block of (separator key -> BlockHandle), a metaindex
    block, and a 48-byte footer (two BlockHandles, zero padding, magic 0xdb4775248b80fb57);
  * crc32c masking (lib/hash/crc32c.h): ((crc >> 15) | (crc << 17)) + 0xa282ead8.
PARITY NOTE: no TensorFlow checkpoint file exists in this container to read back, so the implementation is pinned by the format
invariants above (tests/test_host_logic.py: footer magic, block CRCs, prefix compression with restart points, round trips) -- not
by a TF-written fixture.
Partitioned variables.  Under a partitioner scope (models/DeepFM/deepFM.py:163-175 with num_ps_replicas > 0) TensorFlow creates
`.../embedding_weights/part_N` variables carrying a SaveSliceInfo, and the Saver stores them as SLICES of the full tensor: the index holds
the full name -> BundleEntryProto {dtype, FULL shape, slices = [TensorSliceProto ...]} (no data of its own) and, per slice, a key
checkpoint::EncodeTensorNameSlice(name, slice) -> BundleEntryProto {dtype, slice shape, shard_id, offset, size, crc32c}
(tensorflow/core/util/saved_tensor_slice_util.cc; the key is an OrderedCode string: NumIncreasing(0), String(name),
NumIncreasing(rank), then SignedNumIncreasing(start), SignedNumIncreasing(length) per dimension -- TensorFlow's encoder may write a full
dimension as (0, -1), which read_bundle accepts; write_bundle writes the explicit (0, dim) extents since round 4;
tensorflow/core/lib/strings/ordered_code.cc).  read_bundle assembles such a variable from its slices; write_bundle(partitions=) cuts a
tensor along axis 0 into the 'div' row ranges of shard.div_range and writes it that way.  Restated from the published sources like the
rest of the format; KATs of the key encoding in tests/test_host_logic.py.
"""
import os
import struct

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8
# tensorflow/core/framework/types.proto
_DT = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 6: np.int8, 9: np.int64, 10: np.bool_, 5: np.int16}
_NP2DT = {np.dtype(v): k for k, v in _DT.items()}
_BLOCK_SIZE = 256 << 10       # table::Options block_size used by BundleWriter
_RESTART = 16


def _crc32c(data, crc=0):
    """CRC-32C through the C ABI's host function (falls back to a table loop when the library is not built)."""
    try:
        from . import _lib
        lib = _lib.load()
        buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data.reshape(-1).view(np.uint8)
        if buf.size == 0:
            return crc
        buf = np.ascontiguousarray(buf)
        return int(lib.dir_crc32c(crc, buf.ctypes.data, buf.size))
    except (RuntimeError, OSError, AttributeError):      # library not built / a stale build without dir_crc32c
        return _crc32c_py(bytes(data), crc)


_PYT = None


def _crc32c_py(data, crc=0):
    """Table-driven fallback (~10 MB/s in CPython: fine for index blocks; for multi-GB tensor shards build the library, or read with
    verify=False)."""
    global _PYT
    if _PYT is None:
        _PYT = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            _PYT.append(c)
    c = crc ^ 0xffffffff
    for b in data:
        c = _PYT[(c ^ b) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


def mask_crc(crc):
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xffffffff


def unmask_crc(m):
    rot = (m - _MASK_DELTA) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ---- varints / the two protos (hand-rolled: three message types, a dozen fields) ------------------------------------------------------
def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7f
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    n = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        n |= (b & 0x7f) << shift
        if not b & 0x80:
            return n, pos
        shift += 7


def _fields(buf):
    """(field number, wire type, value) of a serialized proto; length-delimited values as bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        tag, pos = _read_varint(buf, pos)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield f, wt, v


def _header_proto():
    version = b"\x08" + _varint(1)                           # VersionDef.producer = 1
    return b"\x08" + _varint(1) + b"\x1a" + _varint(len(version)) + version      # num_shards = 1, endianness LITTLE (default), version


def _shape_proto(shape):
    out = b""
    for d in shape:
        dim = b"\x08" + _varint(int(d))                      # Dim.size
        out += b"\x12" + _varint(len(dim)) + dim             # TensorShapeProto.dim
    return out


def _entry_proto(dtype, shape, offset, size, crc_masked, slices=None):
    out = b"\x08" + _varint(dtype)
    sp = _shape_proto(shape)
    out += b"\x12" + _varint(len(sp)) + sp
    if slices is not None:                                   # a partitioned variable's header entry: full shape + its slices, no data
        for ext in slices:
            sl = _slice_proto(ext)
            out += b"\x3a" + _varint(len(sl)) + sl
        return out
    if offset:
        out += b"\x20" + _varint(offset)                     # shard_id = 0 (default) is not written
    out += b"\x28" + _varint(size)
    out += b"\x35" + struct.pack("<I", crc_masked)
    return out


def _parse_entry(buf):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "slices": []}
    for f, wt, v in _fields(buf):
        if f == 1:
            e["dtype"] = v
        elif f == 2:
            for f2, _, v2 in _fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1:
                            size = v3 if v3 < (1 << 63) else v3 - (1 << 64)
                    e["shape"].append(size)
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = v
        elif f == 7:
            e["slices"].append(_parse_slice(v))
    return e


# ---- tensor slices (partitioned variables) -------------------------------------------------------------------------------------------
def _parse_slice(buf):
    """TensorSliceProto {repeated Extent extent = 1}, Extent {int64 start = 1; oneof {int64 length = 2}} -> [(start, length | -1)]
    (-1 = the whole dimension: TensorSlice::kFullExtent)."""
    ext = []
    for f, _, v in _fields(buf):
        if f == 1:
            start, length = 0, -1
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    start = v2 if v2 < (1 << 63) else v2 - (1 << 64)
                elif f2 == 2:
                    length = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            ext.append((start, length))
    return ext


def _slice_proto(extents):
    out = b""
    for start, length in extents:
        e = b""
        if start:
            e += b"\x08" + _varint(start)
        if length >= 0:
            e += b"\x10" + _varint(length)
        out += b"\x0a" + _varint(len(e)) + e
    return out


def _oc_num_increasing(v):
    """OrderedCode::WriteNumIncreasing: one length byte, then the value big-endian without leading zero bytes."""
    b = b""
    while v > 0:
        b = bytes([v & 0xff]) + b
        v >>= 8
    return bytes([len(b)]) + b


def _oc_string(s):
    """OrderedCode::WriteString: \\x00 -> \\x00\\xff, \\xff -> \\xff\\x00, terminated by \\x00\\x01."""
    return _oc_escape(s) + b"\x00\x01"


def _oc_escape(s):
    out = bytearray()
    for c in s:
        if c == 0:
            out += b"\x00\xff"
        elif c == 0xff:
            out += b"\xff\x00"
        else:
            out.append(c)
    return bytes(out)


_OC_HEADER = {1: (0x80, 0), 2: (0xc0, 0), 3: (0xe0, 0), 4: (0xf0, 0), 5: (0xf8, 0), 6: (0xfc, 0), 7: (0xfe, 0), 8: (0xff, 0),
              9: (0xff, 0x80), 10: (0xff, 0xc0)}


def _oc_signed_increasing(val):
    """OrderedCode::WriteSignedNumIncreasing: 7 payload bits per byte, a unary length prefix XORed into the sign-extended big-endian
    value (|val| < 64 -> one byte 0x80 ^ val)."""
    x = ~val if val < 0 else val
    if x < 64:
        return bytes([(0x80 ^ val) & 0xff])
    bits = x.bit_length() + 1                               # value bits + the sign bit
    n = -(-bits // 7)
    raw = (val & ((1 << 80) - 1)).to_bytes(10, "big")       # sign-extended to 10 bytes
    b = bytearray(raw[10 - n:])
    h0, h1 = _OC_HEADER[n]
    b[0] ^= h0
    if n >= 2:
        b[1] ^= h1
    return bytes(b)


def encode_tensor_name_slice(name, extents):
    """checkpoint::EncodeTensorNameSlice: the index key under which one slice of a partitioned variable is stored."""
    out = _oc_num_increasing(0) + _oc_string(name.encode("utf-8")) + _oc_num_increasing(len(extents))
    for start, length in extents:
        out += _oc_signed_increasing(start) + _oc_signed_increasing(length)
    return out


# ---- the sorted table ------------------------------------------------------------------------------------------------------------------
class _BlockBuilder:
    def __init__(self):
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last = b""

    def add(self, key, value):
        shared = 0
        if self.count % _RESTART == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def finish(self):
        out = bytes(self.buf)
        for r in self.restarts:
            out += struct.pack("<I", r)
        return out + struct.pack("<I", len(self.restarts))

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4


def _write_block(f, contents):
    """block + trailer (type 0 = uncompressed, masked crc32c over contents + type) -> (offset, size) handle."""
    off = f.tell()
    f.write(contents)
    crc = _crc32c(b"\x00", _crc32c(contents))
    f.write(b"\x00" + struct.pack("<I", mask_crc(crc)))
    return off, len(contents)


def _handle(off, size):
    return _varint(off) + _varint(size)


def write_table(path, items):
    """items: sorted [(key bytes, value bytes)] -> a leveldb-format table file."""
    with open(path, "wb") as f:
        index = _BlockBuilder()
        blk = _BlockBuilder()
        pending = None                      # (last key of the finished block, its handle)

        def flush():
            nonlocal blk, pending
            if blk.count:
                pending = (blk.last, _write_block(f, blk.finish()))
                blk = _BlockBuilder()

        for key, value in items:
            if pending is not None:         # the index key only has to separate the two blocks: use the finished block's last key
                index.add(pending[0], _handle(*pending[1]))
                pending = None
            blk.add(key, value)
            if blk.size() >= _BLOCK_SIZE:
                flush()
        flush()
        if pending is not None:
            index.add(pending[0], _handle(*pending[1]))
        meta = _write_block(f, _BlockBuilder().finish())
        idx = _write_block(f, index.finish())
        footer = _handle(*meta) + _handle(*idx)
        footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC)
        f.write(footer)


def _read_block(buf, off, size, verify=True):
    contents = buf[off:off + size]
    typ = buf[off + size]
    crc = struct.unpack_from("<I", buf, off + size + 1)[0]
    if typ != 0:
        raise ValueError("compressed table blocks are not supported (type %d)" % typ)
    if verify and unmask_crc(crc) != _crc32c(bytes([typ]), _crc32c(contents)):
        raise ValueError("table block at offset %d fails its crc32c" % off)
    nrestart = struct.unpack_from("<I", contents, len(contents) - 4)[0]
    end = len(contents) - 4 - 4 * nrestart
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _read_varint(contents, pos)
        non_shared, pos = _read_varint(contents, pos)
        vlen, pos = _read_varint(contents, pos)
        key = key[:shared] + bytes(contents[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(contents[pos:pos + vlen])))
        pos += vlen
    return out


def read_table(path, verify=True):
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != _MAGIC:
        raise ValueError("%s is not a TensorFlow / leveldb table (bad magic)" % path)
    foot = buf[len(buf) - 48:]
    _, p = _read_varint(foot, 0)
    _, p = _read_varint(foot, p)
    ioff, p = _read_varint(foot, p)
    isize, p = _read_varint(foot, p)
    items = []
    for _, handle in _read_block(buf, ioff, isize, verify):
        off, q = _read_varint(handle, 0)
        size, _ = _read_varint(handle, q)
        items.extend(_read_block(buf, off, size, verify))
    return items


# ---- bundles -----------------------------------------------------------------------------------------------------------------------------
def write_bundle(prefix, tensors, partitions=None):
    """tensors: {name: ndarray} -> `<prefix>.index` + `<prefix>.data-00000-of-00001` (one shard, little endian).
    partitions: {name: n} -- write that tensor as a partitioned variable of n axis-0 slices ('div' row ranges, what
    min_max_variable_partitioner + partition_strategy='div' produce: models/DeepFM/deepFM.py:163-167)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    partitions = dict(partitions or {})
    items = {b"": _header_proto()}
    with open(prefix + ".data-00000-of-00001", "wb") as data:
        def put(key, a):
            off = data.tell()
            raw = a.reshape(-1).view(np.uint8)
            data.write(raw.tobytes() if raw.size < (1 << 20) else memoryview(raw))
            items[key] = _entry_proto(_NP2DT[a.dtype], a.shape, off, raw.size, mask_crc(_crc32c(raw)))

        for name in sorted(tensors, key=lambda s: s.encode("utf-8")):
            a = np.asarray(tensors[name])
            if not a.flags.c_contiguous:
                a = a.copy(order="C")                         # (np.ascontiguousarray would turn a scalar into shape (1,))
            if a.dtype not in _NP2DT:
                raise TypeError("%s: dtype %s has no checkpoint DataType here" % (name, a.dtype))
            a = a.astype(a.dtype.newbyteorder("<"), copy=False)
            n = int(partitions.get(name, 1))
            if n <= 1:
                put(name.encode("utf-8"), a)
                continue
            if a.ndim < 1 or n > a.shape[0]:
                raise ValueError("%s: cannot cut %s into %d axis-0 slices" % (name, a.shape, n))
            q, r = divmod(a.shape[0], n)
            exts, start = [], 0
            for j in range(n):
                rows = q + 1 if j < r else q
                # [TF-upstream] a Saver writes what SaveSliceInfo.spec says, and that spec is "offset,length" for EVERY dimension
                # ("0,16" for the unpartitioned axis of a [V, 16] variable), never the full-extent marker: explicit lengths in the
                # TensorSliceProto and in the OrderedCode key
                ext = [(start, rows)] + [(0, int(d)) for d in a.shape[1:]]
                exts.append(ext)
                put(encode_tensor_name_slice(name, ext), np.ascontiguousarray(a[start:start + rows]))
                start += rows
            items[name.encode("utf-8")] = _entry_proto(_NP2DT[a.dtype], a.shape, 0, 0, 0, slices=exts)
    write_table(prefix + ".index", sorted(items.items()))


def read_bundle(prefix, names=None, verify=True):
    """-> {name: ndarray} of `<prefix>.index` + its data shards (all tensors, or just `names`).  A partitioned variable (an entry
    with `slices`) comes back whole, assembled from its slices; the slices' own keys are not listed."""
    items = read_table(prefix + ".index", verify)
    if not items or items[0][0] != b"":
        raise ValueError("%s.index has no bundle header" % prefix)
    num_shards = 1
    for f, _, v in _fields(items[0][1]):
        if f == 1:
            num_shards = v
        elif f == 2 and v != 0:
            raise ValueError("big-endian bundles are not supported")
    table = dict(items[1:])
    shards = {}

    def load(name, e):
        if e["dtype"] not in _DT:
            raise TypeError("%s: checkpoint DataType %d is not supported" % (name, e["dtype"]))
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = np.memmap("%s.data-%05d-of-%05d" % (prefix, sid, num_shards), dtype=np.uint8, mode="r")
        raw = np.asarray(shards[sid][e["offset"]:e["offset"] + e["size"]])
        if verify and e["crc32c"] is not None and unmask_crc(e["crc32c"]) != _crc32c(raw):
            raise ValueError("%s fails its crc32c" % name)
        dt = np.dtype(_DT[e["dtype"]]).newbyteorder("<")
        return raw.view(dt).reshape(e["shape"]).astype(_DT[e["dtype"]], copy=True)

    out = {}
    for key, val in items[1:]:
        if key[:1] == b"\x00":                              # a slice's own key (EncodeTensorNameSlice starts with NumIncreasing(0))
            continue
        name = key.decode("utf-8")
        if names is not None and name not in names:
            continue
        e = _parse_entry(val)
        if not e["slices"]:
            out[name] = load(name, e)
            continue
        if e["dtype"] not in _DT:
            raise TypeError("%s: checkpoint DataType %d is not supported" % (name, e["dtype"]))
        full = np.empty(e["shape"], _DT[e["dtype"]])
        covered = 0
        for ext in e["slices"]:
            skey = encode_tensor_name_slice(name, ext)
            if skey not in table:
                raise ValueError("%s: the index lists a slice %s whose data entry is missing" % (name, ext))
            part = load("%s%s" % (name, ext), _parse_entry(table[skey]))
            idx = tuple(slice(st, None if ln < 0 else st + ln) for st, ln in ext)
            if full[idx].shape != part.shape:
                raise ValueError("%s: slice %s has shape %s, expected %s" % (name, ext, part.shape, full[idx].shape))
            full[idx] = part
            covered += part.size
        if covered != full.size:
            raise ValueError("%s: its %d slices cover %d of %d elements" % (name, len(e["slices"]), covered, full.size))
        out[name] = full
    return out


def write_checkpoint_state(model_dir, ckpt_name):
    """The `checkpoint` text proto an Estimator keeps in model_dir (CheckpointState)."""
    with open(os.path.join(model_dir, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (ckpt_name, ckpt_name))


def latest_checkpoint(model_dir):
    """[TF-upstream] tf.train.latest_checkpoint: the prefix named by model_dir/checkpoint (None if absent)."""
    p = os.path.join(model_dir, "checkpoint")
    if not os.path.exists(p):
        return None
    for line in open(p):
        if line.startswith("model_checkpoint_path:"):
            name = line.split(":", 1)[1].strip().strip('"')
            return name if os.path.isabs(name) else os.path.join(model_dir, name)
    return None

"""Reader for TensorFlow V2 checkpoint bundles (`<prefix>.index` + `<prefix>.data-00000-of-00001`)
without TensorFlow.

The reference restores `./trained_model/81448_0-1000000` (denoiser, SN/apply.py:430-432) and
`./trained_model/81457_2-545000` (separator, SS/apply.py:369-372) through tf.train.Saver.  The
bundle format is TensorFlow's (tensor_bundle + an LevelDB-style table); this file restates the
published on-disk layout:

  .index  = table of prefix-compressed (key -> value) blocks, 48-byte footer holding the
            metaindex and index BlockHandles and the magic 0xdb4775248b80fb57.
            key ""   -> BundleHeaderProto, key <tensor name> -> BundleEntryProto
            {1: dtype, 2: shape{2: dim{1: size}}, 3: shard_id, 4: offset, 5: size, 6: crc32c}
  .data-* = raw little-endian tensors at `offset`.

Integrity: every table block carries a 5-byte trailer (compression type + masked CRC-32C of block
and type byte) and every BundleEntryProto the masked CRC-32C of its tensor bytes
(mask(c) = rotr15(c) + 0xa282ead8).  Both are verified here -- the block trailers always, the
tensors by load_checkpoint(verify_crc=True).  The implementation is pinned to TensorFlow-written
values by the shipped `.index` files (tests/test_host.py: all their block trailers, and the one
tensor whose bytes are known, the separator's int32 scalar `Variable` == 0).
"""
import os
import struct
from collections import OrderedDict

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
DT_FLOAT, DT_INT32 = 1, 3
_DTYPES = {DT_FLOAT: np.dtype("<f4"), DT_INT32: np.dtype("<i4")}


def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _block_entries(raw):
    """Decode one table block (without its 5-byte trailer) into [(key, value)]."""
    n_restarts = struct.unpack_from("<I", raw, len(raw) - 4)[0]
    end = len(raw) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _varint(raw, pos)
        non_shared, pos = _varint(raw, pos)
        vlen, pos = _varint(raw, pos)
        key = key[:shared] + raw[pos:pos + non_shared]
        pos += non_shared
        out.append((key, raw[pos:pos + vlen]))
        pos += vlen
    return out


_CRC_TABLE = None


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli, reflected 0x82F63B78) of a bytes-like object.  Large buffers go through
    the library's host helper `nhans_crc32c` when it is built; the table loop below is the same
    function in plain Python."""
    global _CRC_TABLE
    mv = memoryview(data).cast("B")
    if len(mv) >= 4096:
        try:
            from . import hip
            lib = hip.load()
            buf = np.frombuffer(mv, dtype=np.uint8)
            return int(lib.nhans_crc32c(crc, buf.ctypes.data, len(buf)))
        except (ImportError, OSError, RuntimeError):
            pass
    if _CRC_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = tab
    c = crc ^ 0xFFFFFFFF
    for b in mv:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    """What TensorFlow stores: crc32c::Mask(crc32c::Value(data))."""
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _read_block(data, offset, size):
    if data[offset + size] != 0:
        raise ValueError("compressed table blocks are not supported")
    stored = struct.unpack_from("<I", data, offset + size + 1)[0]
    if masked_crc32c(data[offset:offset + size + 1]) != stored:
        raise ValueError("table block at offset %d: crc32c mismatch (corrupt .index)" % offset)
    return _block_entries(data[offset:offset + size])


def _parse_proto(buf):
    """Minimal protobuf wire decoder -> {field: [values]} (varint, fixed32, length-delimited)."""
    pos, out = 0, {}
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 5:
            val = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wt == 1:
            val = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.setdefault(field, []).append(val)
    return out


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def read_index(index_path):
    """-> OrderedDict name -> dict(dtype, shape, shard, offset, size, crc32c), in key order."""
    with open(index_path, "rb") as f:
        data = f.read()
    if len(data) < 48:
        raise ValueError("index file too short")
    footer = data[-48:]
    if struct.unpack_from("<Q", footer, 40)[0] != TABLE_MAGIC:
        raise ValueError("bad table magic in %s" % index_path)
    pos = 0
    _, pos = _varint(footer, pos)          # metaindex handle (unused)
    _, pos = _varint(footer, pos)
    ioff, pos = _varint(footer, pos)
    isize, pos = _varint(footer, pos)
    entries = OrderedDict()
    for _, handle in _read_block(data, ioff, isize):
        boff, p = _varint(handle, 0)
        bsize, p = _varint(handle, p)
        for key, val in _read_block(data, boff, bsize):
            if key == b"":
                continue                     # BundleHeaderProto
            msg = _parse_proto(val)
            shape = []
            for sh in msg.get(2, []):
                for dim in _parse_proto(sh).get(2, []):
                    shape.append(_signed(_parse_proto(dim).get(1, [0])[0]))
            entries[key.decode("utf-8")] = dict(
                dtype=msg.get(1, [0])[0], shape=tuple(shape), shard=msg.get(3, [0])[0],
                offset=msg.get(4, [0])[0], size=msg.get(5, [0])[0], crc32c=msg.get(6, [0])[0])
    return entries


def data_path(prefix):
    return prefix + ".data-00000-of-00001"


def is_lfs_pointer(path):
    try:
        if os.path.getsize(path) > 1024:
            return False
        with open(path, "rb") as f:
            return f.read(40).startswith(b"version https://git-lfs")
    except OSError:
        return False


def load_checkpoint(prefix, verify_crc=True):
    """Read every float/int tensor of a bundle -> OrderedDict name -> ndarray.  Raises
    FileNotFoundError when the data shard is missing or is a git-LFS pointer (as shipped in
    the reference tree) and ValueError when a tensor's bytes do not match the masked CRC-32C the
    index records for it (verify_crc=False skips that check)."""
    entries = read_index(prefix + ".index")
    dpath = data_path(prefix)
    if not os.path.exists(dpath) or is_lfs_pointer(dpath):
        raise FileNotFoundError(
            "%s is missing or a git-LFS pointer; fetch the real checkpoint blob" % dpath)
    total = max(e["offset"] + e["size"] for e in entries.values())
    if os.path.getsize(dpath) < total:
        raise ValueError("%s is shorter (%d B) than the index requires (%d B)"
                         % (dpath, os.path.getsize(dpath), total))
    out = OrderedDict()
    with open(dpath, "rb") as f:
        for name, e in entries.items():
            dt = _DTYPES.get(e["dtype"])
            if dt is None:
                continue
            f.seek(e["offset"])
            raw = f.read(e["size"])
            if verify_crc and masked_crc32c(raw) != e["crc32c"]:
                raise ValueError("%s: tensor %s does not match its crc32c in the index (corrupt or wrong data shard)"
                                 % (dpath, name))
            arr = np.frombuffer(raw, dtype=dt)
            out[name] = arr.reshape(e["shape"]).copy()
    return out

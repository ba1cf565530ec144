"""Reader and writer for TensorFlow V2 checkpoints ("tensor bundles": `<prefix>.index` +
`<prefix>.data-NNNNN-of-MMMMM`) without TensorFlow, so that the snapshots the reference writes
(`tf.train.Saver`, exp_clevr/train_clevr_gt_layout.py:147,221-223; README.md:75-79) can be loaded by
variable name, and the drop-in's own `Saver.save` (n2nmn_amd.runtime) leaves files of the same format:

    from n2nmn_amd.tf_checkpoint import read_checkpoint
    engine.load_weights(read_checkpoint('exp_clevr/tfmodel/clevr_gt_layout/00050000'),
                        strict=False)          # optimiser slots etc. are ignored by name

Format, restated from the TensorFlow 1.x sources (tensorflow/core/util/tensor_bundle/tensor_bundle.cc,
tensorflow/core/lib/io/{table,format,block}.cc -- the LevelDB table format -- and
tensorflow/core/protobuf/tensor_bundle.proto); no TensorFlow is available in this environment, so the
reader is checked against files produced by an independent writer of the same specification
(tests/test_tf_checkpoint.py) and against the published constants of the format (table magic
number, CRC32C test vector, CRC mask), NOT against a file written by TensorFlow itself:

  .index   LevelDB table.  Footer (last 48 bytes): metaindex BlockHandle, index BlockHandle (each
           two varint64: offset, size), padding, 8-byte magic 0xdb4775248b80fb57 (little endian).
           A block = entries [shared varint32][non_shared varint32][value_len varint32]
           [key suffix][value] ..., then uint32 restart offsets and uint32 restart count; on disk
           it is followed by a 1-byte compression type (0 = none, 1 = snappy) and a masked CRC32C.
           The index block maps separator keys to data-block handles.  Key "" holds a
           BundleHeaderProto (fields num_shards = 1, endianness = 2 [0 = little], version = 3); every other key is a
           tensor name with a BundleEntryProto (dtype = 1, shape = 2, shard_id = 3, offset = 4,
           size = 5, crc32c = 6 fixed32, slices = 7).
  .data-*  raw little-endian tensor bytes at [offset, offset + size) of shard shard_id.
"""
from __future__ import annotations

import os
import struct
from typing import Dict, List, Optional, Tuple

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_CRC_MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto (the dtypes a Saver writes for this model + a few more)
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8,
          9: np.int64, 10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}


# ---- CRC32C (Castagnoli), as used by LevelDB / TensorFlow ----------------------------------------
def _make_crc_table():
    poly = 0x82f63b78
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC_TABLE = _make_crc_table()


_NATIVE = [None]        # n2nmn_crc32c of the shared library, once it is loaded (False: not available)


def _native_crc():
    if _NATIVE[0] is None:
        _NATIVE[0] = False
        try:
            from . import _lib
            fn = _lib.lib().n2nmn_crc32c
            _NATIVE[0] = fn
        except Exception:
            pass
    return _NATIVE[0]


def crc32c_py(data: bytes, crc: int = 0) -> int:
    """table-driven byte loop (the definition; small inputs and boxes without the library)"""
    c = crc ^ 0xffffffff
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


def crc32c(data: bytes, crc: int = 0) -> int:
    """CRC-32C (Castagnoli) of `data`, continuing from `crc`.  Tensors (megabytes) go through the library's
    n2nmn_crc32c (SSE4.2 crc32 instruction / slicing tables, include/n2nmn.h section 8) when it is loaded."""
    if len(data) >= 4096:
        fn = _native_crc()
        if fn:
            return int(fn(crc, bytes(data) if not isinstance(data, bytes) else data, len(data)))
    return crc32c_py(data, crc)


def mask_crc(crc: int) -> int:
    """crc32c::Mask: rotate right by 15 bits and add a constant."""
    return ((((crc >> 15) | (crc << 17)) & 0xffffffff) + _CRC_MASK_DELTA) & 0xffffffff


def unmask_crc(masked: int) -> int:
    rot = (masked - _CRC_MASK_DELTA) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ---- varints / minimal protobuf ----------------------------------------------------------------
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7f) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 63:
            raise ValueError('tf_checkpoint: malformed varint')


def _proto_fields(buf: bytes):
    """yields (field number, wire type, value) of one protobuf message (varint, 64-bit, bytes,
    32-bit wire types)."""
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            val = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            val = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError('tf_checkpoint: unsupported protobuf wire type %d' % wt)
        yield field, wt, val


def _signed(v: int) -> int:
    return v - (1 << 64) if v >= 1 << 63 else v


def _parse_shape(buf: bytes) -> Tuple[int, ...]:
    dims = []
    for field, _, val in _proto_fields(buf):
        if field == 2:                                   # repeated Dim dim = 2
            size = 0
            for f2, _, v2 in _proto_fields(val):
                if f2 == 1:                              # int64 size = 1
                    size = _signed(v2)
            dims.append(size)
        elif field == 3 and val:                         # unknown_rank
            raise ValueError('tf_checkpoint: tensor of unknown rank')
    return tuple(dims)


def _parse_entry(buf: bytes) -> dict:
    e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for field, _, val in _proto_fields(buf):
        if field == 1:
            e['dtype'] = val
        elif field == 2:
            e['shape'] = _parse_shape(val)
        elif field == 3:
            e['shard_id'] = val
        elif field == 4:
            e['offset'] = val
        elif field == 5:
            e['size'] = val
        elif field == 6:
            e['crc32c'] = val
        elif field == 7:
            e['sliced'] = True
    return e


def _parse_header(buf: bytes) -> dict:
    h = dict(num_shards=1, endianness=0, version=None)
    for field, _, val in _proto_fields(buf):
        if field == 1:
            h['num_shards'] = val
        elif field == 2:
            h['endianness'] = val
        elif field == 3:
            h['version'] = val
    return h


# ---- LevelDB table -----------------------------------------------------------------------------
def _read_block(data: bytes, offset: int, size: int, verify: bool) -> bytes:
    raw = data[offset:offset + size]
    if len(raw) != size or offset + size + 5 > len(data):
        raise ValueError('tf_checkpoint: truncated table block')
    ctype = data[offset + size]
    if verify:
        stored = struct.unpack_from('<I', data, offset + size + 1)[0]
        if unmask_crc(stored) != crc32c(data[offset:offset + size + 1]):
            raise ValueError('tf_checkpoint: table block checksum mismatch')
    if ctype == 0:
        return raw
    if ctype == 1:
        raise NotImplementedError('tf_checkpoint: snappy-compressed table block (the bundle writer of '
                                  'TensorFlow 1.x writes its index uncompressed)')
    raise ValueError('tf_checkpoint: unknown block compression type %d' % ctype)


def _block_entries(block: bytes) -> List[Tuple[bytes, bytes]]:
    if len(block) < 4:
        raise ValueError('tf_checkpoint: malformed table block')
    nrestarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestarts
    if end < 0:
        raise ValueError('tf_checkpoint: malformed table block')
    out = []
    pos = 0
    key = b''
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


def read_index(path: str, verify: bool = True) -> Tuple[dict, Dict[str, dict]]:
    """(header, {tensor name: entry}) of a `<prefix>.index` file."""
    data = open(path, 'rb').read()
    if len(data) < 48:
        raise ValueError('tf_checkpoint: %s is too short to be a table' % path)
    footer = data[-48:]
    if struct.unpack_from('<Q', footer, 40)[0] != TABLE_MAGIC:
        raise ValueError('tf_checkpoint: %s is not a LevelDB table (bad magic number)' % path)
    pos = 0
    _, pos = _varint(footer, pos)                        # metaindex handle (unused)
    _, pos = _varint(footer, pos)
    ioff, pos = _varint(footer, pos)
    isize, pos = _varint(footer, pos)
    header: Optional[dict] = None
    entries: Dict[str, dict] = {}
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, p2 = _varint(handle, 0)
        bsize, _ = _varint(handle, p2)
        for key, val in _block_entries(_read_block(data, boff, bsize, verify)):
            if key == b'':
                header = _parse_header(val)
            elif key[:1] == b'\x00':
                continue        # data of one slice of a partitioned variable (checkpoint::EncodeTensorNameSlice
                #                 keys start with the ordered code of 0); the full tensor's own entry says `sliced`
            else:
                entries[key.decode('utf-8')] = _parse_entry(val)
    if header is None:
        raise ValueError('tf_checkpoint: %s has no bundle header entry' % path)
    if header['endianness'] != 0:
        raise NotImplementedError('tf_checkpoint: big-endian bundle')
    return header, entries


def read_checkpoint(prefix: str, names=None, verify: bool = True,
                    skip_missing: bool = False) -> Dict[str, np.ndarray]:
    """All (or the named) tensors of the checkpoint `<prefix>.index` / `<prefix>.data-*`, by the
    variable names the graph used.  verify: check the table-block and per-tensor CRC32Cs (the CRC is
    a pure-Python byte loop: pass `names` -- e.g. `Engine.variable_names()` -- so the Adam slots of a
    training snapshot, two thirds of its bytes, are neither read nor checksummed).
    skip_missing: names absent from the checkpoint are left out instead of raising KeyError."""
    header, entries = read_index(prefix + '.index', verify)
    shards: Dict[int, bytes] = {}
    out: Dict[str, np.ndarray] = {}
    for name in (sorted(entries) if names is None else names):
        if skip_missing and name not in entries:
            continue
        if name not in entries:
            raise KeyError('tf_checkpoint: no tensor named %r in %s' % (name, prefix))
        e = entries[name]
        if e['sliced']:
            raise NotImplementedError('tf_checkpoint: partitioned variable %r' % name)
        if e['dtype'] not in DTYPES:
            raise NotImplementedError('tf_checkpoint: dtype %d of %r' % (e['dtype'], name))
        sid = e['shard_id']
        if sid not in shards:
            path = '%s.data-%05d-of-%05d' % (prefix, sid, header['num_shards'])
            if not os.path.exists(path):
                raise FileNotFoundError(path)
            shards[sid] = open(path, 'rb').read()
        raw = shards[sid][e['offset']:e['offset'] + e['size']]
        dt = np.dtype(DTYPES[e['dtype']])
        count = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
        if len(raw) != e['size'] or count * dt.itemsize != e['size']:
            raise ValueError('tf_checkpoint: size of %r does not match its shape' % name)
        if verify and e['crc32c'] is not None and unmask_crc(e['crc32c']) != crc32c(raw):
            raise ValueError('tf_checkpoint: data checksum mismatch for %r' % name)
        out[name] = np.frombuffer(raw, dtype=dt.newbyteorder('<')).astype(dt).reshape(e['shape'])
    return out


# ---- writer ------------------------------------------------------------------------------------
_DTYPE_CODES = {np.dtype(v): k for k, v in DTYPES.items()}


def _enc_varint(v: int) -> bytes:
    out = bytearray()
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _enc_field(num: int, wire: int, payload) -> bytes:
    tag = _enc_varint((num << 3) | wire)
    if wire == 0:
        return tag + _enc_varint(payload)
    if wire == 2:
        return tag + _enc_varint(len(payload)) + payload
    if wire == 5:
        return tag + struct.pack('<I', payload)
    raise ValueError(wire)


class _TableBuilder:
    """LevelDB table (tensorflow/core/lib/io/table_builder.cc): data blocks of about `block_size` bytes with
    prefix-compressed keys and a restart point every 16 entries, one index block, an empty metaindex block,
    every block followed by its compression type byte (0) and masked CRC32C, then the 48-byte footer."""

    def __init__(self, block_size: int = 4096, restart_interval: int = 16):
        self.block_size, self.restart_interval = block_size, restart_interval
        self.out = bytearray()
        self.index = []            # (last key of the block, handle)
        self._reset()

    def _reset(self):
        self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b''

    def add(self, key: bytes, value: bytes):
        if self.count and key <= self.last:
            raise ValueError('table keys must be added in increasing order')
        shared = 0
        if self.count % self.restart_interval == 0:
            if self.count:
                self.restarts.append(len(self.buf))
        else:
            n = min(len(key), len(self.last))
            while shared < n and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _enc_varint(shared) + _enc_varint(len(key) - shared) + _enc_varint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1
        if len(self.buf) >= self.block_size:
            self._flush()

    def _emit(self, block: bytes) -> bytes:
        off = len(self.out)
        body = block + b'\x00'
        self.out += body + struct.pack('<I', mask_crc(crc32c_py(body)))
        return _enc_varint(off) + _enc_varint(len(block))

    @staticmethod
    def _finish_block(buf, restarts) -> bytes:
        return bytes(buf) + b''.join(struct.pack('<I', r) for r in restarts) + struct.pack('<I', len(restarts))

    def _flush(self):
        if not self.count:
            return
        self.index.append((self.last, self._emit(self._finish_block(self.buf, self.restarts))))
        self._reset()

    def finish(self) -> bytes:
        self._flush()
        meta = self._emit(self._finish_block(b'', [0]))
        ib = bytearray()
        rs = []
        for key, handle in self.index:           # the index block restarts at every entry
            rs.append(len(ib))
            ib += _enc_varint(0) + _enc_varint(len(key)) + _enc_varint(len(handle)) + key + handle
        idx = self._emit(self._finish_block(ib, rs or [0]))
        footer = meta + idx
        footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
        return bytes(self.out) + footer


def write_checkpoint(prefix: str, tensors: Dict[str, np.ndarray]) -> None:
    """`<prefix>.index` + `<prefix>.data-00000-of-00001` holding `tensors` (name -> array) as a one-shard
    tensor bundle: header entry under the empty key (num_shards 1, little endian, version producer 1), one
    BundleEntryProto per tensor in key order (dtype, shape, offset, size, masked crc32c of the bytes)."""
    data = bytearray()
    tb = _TableBuilder()
    tb.add(b'', _enc_field(1, 0, 1) + _enc_field(3, 2, _enc_field(1, 0, 1)))
    for name in sorted(tensors, key=lambda k: k.encode('utf-8')):
        a = np.asarray(tensors[name])
        if a.dtype not in _DTYPE_CODES:
            raise NotImplementedError('tf_checkpoint: cannot store dtype %s of %r' % (a.dtype, name))
        raw = np.ascontiguousarray(a).astype(a.dtype.newbyteorder('<'), copy=False).tobytes()
        shape = b''.join(_enc_field(2, 2, _enc_field(1, 0, int(d))) for d in a.shape)
        entry = _enc_field(1, 0, _DTYPE_CODES[a.dtype]) + _enc_field(2, 2, shape)
        if len(data):
            entry += _enc_field(4, 0, len(data))
        entry += _enc_field(5, 0, len(raw)) + _enc_field(6, 5, mask_crc(crc32c(raw)))
        tb.add(name.encode('utf-8'), entry)
        data += raw
    index = tb.finish()
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(bytes(data))
    with open(prefix + '.index', 'wb') as f:
        f.write(index)

"""n2nmn_amd.tf_checkpoint against (a) the published constants of the formats it restates and (b)
checkpoints produced by an INDEPENDENT writer of the same specification kept in this file
(TensorFlow itself is not available here, so no TF-written file can be used -- the module's
docstring says so).  The writer emits what tensorflow/core/util/tensor_bundle/tensor_bundle.cc does:
prefix-compressed keys with restart points every 16 entries, several data blocks, an index block,
an empty metaindex block, masked CRC32C trailers, one data shard with alignment padding."""
import struct

import numpy as np
import pytest

from n2nmn_amd import tf_checkpoint as T


def _vi(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field(num, wt, payload):
    key = _vi((num << 3) | wt)
    if wt == 0:
        return key + _vi(payload)
    if wt == 2:
        return key + _vi(len(payload)) + payload
    if wt == 5:
        return key + struct.pack('<I', payload)
    raise ValueError


def _shape_proto(shape):
    return b''.join(_field(2, 2, _field(1, 0, d)) for d in shape)


def _entry_proto(dtype, shape, offset, size, crc):
    return (_field(1, 0, dtype) + _field(2, 2, _shape_proto(shape)) + _field(4, 0, offset) +
            _field(5, 0, size) + _field(6, 5, crc))        # shard_id 0 is the proto default: omitted


def _block(items, restart_interval=16):
    out = bytearray()
    restarts = []
    prev = b''
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _vi(shared) + _vi(len(k) - shared) + _vi(len(v)) + k[shared:] + v
        prev = k
    for r in restarts or [0]:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts) or 1)
    return bytes(out)


def _write_table(path, items, per_block=5):
    body = bytearray()
    handles = []

    def emit(block):
        off = len(body)
        body.extend(block + b'\x00')                                      # compression type: none
        body.extend(struct.pack('<I', T.mask_crc(T.crc32c(block + b'\x00'))))
        return _vi(off) + _vi(len(block))

    for i in range(0, len(items), per_block):
        chunk = items[i:i + per_block]
        handles.append((chunk[-1][0] + b'\x00', emit(_block(chunk))))     # separator >= last key
    meta = emit(_block([]))
    index = emit(_block(handles, restart_interval=1))
    footer = meta + index
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', T.TABLE_MAGIC)
    open(path, 'wb').write(bytes(body) + footer)


DT = {np.dtype(np.float32): 1, np.dtype(np.int32): 3, np.dtype(np.int64): 9, np.dtype(np.float64): 2}


def write_checkpoint(prefix, tensors):
    data = bytearray()
    items = [(b'', _field(1, 0, 1) + _field(3, 2, _field(1, 0, 1)))]      # header: 1 shard, producer 1
    for name in sorted(tensors):
        a = np.asarray(tensors[name])          # (ascontiguousarray would turn a scalar into [1])
        raw = a.astype(a.dtype.newbyteorder('<')).tobytes()
        off = len(data)
        data.extend(raw)
        items.append((name.encode(), _entry_proto(DT[a.dtype], a.shape, off, len(raw),
                                                  T.mask_crc(T.crc32c(raw)))))
    _write_table(prefix + '.index', items)
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))


def test_published_constants():
    assert T.crc32c(b'123456789') == 0xE3069283              # CRC-32C check value (RFC 3720)
    assert T.crc32c(b'\x00' * 32) == 0x8A9136AA              # RFC 3720 B.4, 32 bytes of zeros
    assert T.unmask_crc(T.mask_crc(0x12345678)) == 0x12345678
    assert T.mask_crc(0) == 0xa282ead8
    assert T.TABLE_MAGIC == 0xdb4775248b80fb57               # leveldb table/format.h kTableMagicNumber


def test_round_trip_of_model_sized_checkpoint(tmp_path):
    from n2nmn_amd.spec import Dims, variable_shapes
    d = Dims(H=4, W=5, D=32, map_dim=18, embed_dim_txt=12, embed_dim_nmn=12, lstm_dim=16,
             num_vocab_txt=11, num_choices=7, T_encoder=6, T_decoder=8, N=6)
    rng = np.random.default_rng(0)
    tensors = {k: rng.standard_normal(s).astype(np.float32) for k, s in variable_shapes(d).items()}
    # what else a Saver of the training scripts stores: Adam slots, a scalar, an int64 step
    some = next(iter(tensors))
    tensors[some + '/Adam'] = np.zeros_like(tensors[some])
    tensors['beta1_power'] = np.float32(0.9).reshape(())
    tensors['global_step'] = np.array(50000, np.int64)
    prefix = str(tmp_path / '00050000')
    write_checkpoint(prefix, tensors)
    header, entries = T.read_index(prefix + '.index')
    assert header['num_shards'] == 1 and set(entries) == set(tensors)
    got = T.read_checkpoint(prefix)
    assert set(got) == set(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    sub = T.read_checkpoint(prefix, names=[some])
    assert list(sub) == [some]
    with pytest.raises(KeyError):
        T.read_checkpoint(prefix, names=['no/such/variable'])


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / 'ckpt')
    write_checkpoint(prefix, {'a/weights': np.arange(12, dtype=np.float32).reshape(3, 4),
                              'b/biases': np.ones(5, np.float32)})
    raw = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    raw[7] ^= 0x40
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(raw))
    with pytest.raises(ValueError, match='checksum'):
        T.read_checkpoint(prefix)
    assert T.read_checkpoint(prefix, verify=False)['b/biases'].sum() == 5
    idx = bytearray(open(prefix + '.index', 'rb').read())
    idx[3] ^= 0x01
    open(prefix + '.index', 'wb').write(bytes(idx))
    with pytest.raises(ValueError):
        T.read_index(prefix + '.index')
    idx[-1] ^= 0xff
    open(prefix + '.index', 'wb').write(bytes(idx))
    with pytest.raises(ValueError, match='magic'):
        T.read_index(prefix + '.index')

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


# ---- round 6: more foreign bytes, the product's own writer, and the reference's variable names ----
def _slices_proto(full_shape, starts, lengths):
    """TensorSliceProto (tensorflow/core/framework/tensor_slice.proto): extent = 1 {start = 1, length = 2}"""
    out = b''
    for s, n in zip(starts, lengths):
        ext = (_field(1, 0, s) if s else b'') + _field(2, 0, n)
        out += _field(1, 2, ext)
    return out


def test_partitioned_variable_is_refused_and_its_slice_keys_are_skipped(tmp_path):
    """A partitioned variable is stored as one entry of the FULL tensor that lists its slices (BundleEntryProto.slices = 7)
    plus one entry per slice under a key that starts with the ordered code of 0 (checkpoint::EncodeTensorNameSlice).  None
    of the reference's variables is partitioned; the reader must say so for such an entry instead of returning the
    empty full-tensor entry, must not trip over the binary slice keys, and must still serve the ordinary tensors of the
    same bundle."""
    data = bytearray()
    plain = np.arange(6, dtype=np.float32).reshape(2, 3)
    raw = plain.tobytes()
    items = [(b'', _field(1, 0, 1) + _field(3, 2, _field(1, 0, 1)))]
    # slice data entries: keys sort in front of every name (first byte 0x00)
    for part in range(2):
        sl = np.full((2, 4), part, np.float32).tobytes()
        key = b'\x00\x01' + b'emb/weights' + b'\x00\x01' + bytes([2, part * 2, 2, 0, 4])      # (opaque to the reader)
        items.append((key, _entry_proto(1, (2, 4), len(data), len(sl), T.mask_crc(T.crc32c(sl)))))
        data.extend(sl)
    off = len(data)
    data.extend(raw)
    items.append((b'a/plain', _entry_proto(1, plain.shape, off, len(raw), T.mask_crc(T.crc32c(raw)))))
    full = _field(1, 0, 1) + _field(2, 2, _shape_proto((4, 4))) + \
        _field(7, 2, _slices_proto((4, 4), (0, 0), (2, 4))) + _field(7, 2, _slices_proto((4, 4), (2, 0), (2, 4)))
    items.append((b'emb/weights', full))
    items.sort(key=lambda kv: kv[0])
    prefix = str(tmp_path / 'part')
    _write_table(prefix + '.index', items, per_block=2)
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
    header, entries = T.read_index(prefix + '.index')
    assert set(entries) == {'a/plain', 'emb/weights'} and entries['emb/weights']['sliced']
    assert np.array_equal(T.read_checkpoint(prefix, names=['a/plain'])['a/plain'], plain)
    with pytest.raises(NotImplementedError, match='partitioned'):
        T.read_checkpoint(prefix)


def test_bad_checksum_in_a_later_index_block_and_in_one_tensor(tmp_path):
    """corruption that sits neither in the first block nor in the first tensor"""
    rng = np.random.default_rng(1)
    tensors = {'v%02d/weights' % i: rng.standard_normal((5, 7)).astype(np.float32) for i in range(23)}
    prefix = str(tmp_path / 'many')
    write_checkpoint(prefix, tensors)               # 24 entries, 5 per block: 5 data blocks
    idx = bytearray(open(prefix + '.index', 'rb').read())
    idx[len(idx) // 2] ^= 0x10                      # somewhere in a middle block
    open(prefix + '.index', 'wb').write(bytes(idx))
    with pytest.raises(ValueError):
        T.read_index(prefix + '.index')
    write_checkpoint(prefix, tensors)
    raw = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    raw[17 * 5 * 7 * 4 + 9] ^= 0x01                 # inside the 18th tensor
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(raw))
    good = sorted(tensors)[:17]
    assert set(T.read_checkpoint(prefix, names=good)) == set(good)       # (only the named tensors are checksummed)
    with pytest.raises(ValueError, match="checksum mismatch for 'v17/weights'"):
        T.read_checkpoint(prefix)


def test_product_writer_against_the_independent_format_statement(tmp_path):
    """n2nmn_amd.tf_checkpoint.write_checkpoint (what the drop-in's Saver.save leaves, runtime.py) parsed with THIS file's
    own statement of the format: footer / magic, block trailers (compression byte + masked CRC32C), restart arrays,
    prefix-compressed keys, header and entry protos, data shard -- no function of the module under test on the read
    side except crc32c, which test_published_constants pins."""
    rng = np.random.default_rng(2)
    tensors = {'neural_module_network/m%03d/weights' % i: rng.standard_normal((3, 4)).astype(np.float32)
               for i in range(150)}                  # > one 4 KiB block of entries
    tensors['Variable'] = np.float32(0.5).reshape(())
    tensors['global_step'] = np.array(7, np.int64)
    prefix = str(tmp_path / 'w')
    T.write_checkpoint(prefix, tensors)
    idx = open(prefix + '.index', 'rb').read()
    data = open(prefix + '.data-00000-of-00001', 'rb').read()
    assert struct.unpack('<Q', idx[-8:])[0] == 0xdb4775248b80fb57

    def varint(buf, pos):
        out = shift = 0
        while True:
            b = buf[pos]
            pos += 1
            out |= (b & 0x7f) << shift
            if not b & 0x80:
                return out, pos
            shift += 7

    def block(off, size):
        body = idx[off:off + size + 1]
        assert body[-1] == 0                                                      # no compression
        assert struct.unpack('<I', idx[off + size + 1:off + size + 5])[0] == T.mask_crc(T.crc32c(body))
        blk = body[:-1]
        nrest = struct.unpack('<I', blk[-4:])[0]
        restarts = struct.unpack('<%dI' % nrest, blk[-4 - 4 * nrest:-4])
        end = len(blk) - 4 - 4 * nrest
        pos, key, out, starts = 0, b'', [], []
        while pos < end:
            starts.append(pos)
            shared, pos = varint(blk, pos)
            non, pos = varint(blk, pos)
            vlen, pos = varint(blk, pos)
            key = key[:shared] + blk[pos:pos + non]
            pos += non
            out.append((key, blk[pos:pos + vlen]))
            pos += vlen
        assert all(r in starts for r in restarts) and restarts[0] == 0
        assert all(starts.index(r) % 16 == 0 for r in restarts) or len(restarts) == len(out)   # data: every 16; index: every entry
        return out

    foot = idx[-48:]
    _, p = varint(foot, 0)
    _, p = varint(foot, p)
    ioff, p = varint(foot, p)
    isize, p = varint(foot, p)
    entries = []
    handles = block(ioff, isize)
    assert len(handles) >= 2                                                      # several data blocks
    for sep, h in handles:
        boff, q = varint(h, 0)
        bsize, _ = varint(h, q)
        blk = block(boff, bsize)
        assert blk[-1][0] <= sep
        entries += blk
    keys = [k for k, _ in entries]
    assert keys == sorted(keys) and keys[0] == b'' and len(keys) == len(tensors) + 1
    dt_of = {1: np.float32, 9: np.int64}
    for k, v in entries[1:]:
        f = {}
        pos = 0
        while pos < len(v):
            tag, pos = varint(v, pos)
            if tag & 7 == 0:
                f[tag >> 3], pos = varint(v, pos)
            elif tag & 7 == 2:
                n, pos = varint(v, pos)
                f[tag >> 3] = v[pos:pos + n]
                pos += n
            else:
                f[tag >> 3] = struct.unpack('<I', v[pos:pos + 4])[0]
                pos += 4
        want = np.asarray(tensors[k.decode()])
        raw = data[f.get(4, 0):f.get(4, 0) + f[5]]
        assert f[6] == T.mask_crc(T.crc32c(raw))
        got = np.frombuffer(raw, dt_of[f[1]]).reshape(want.shape)
        assert np.array_equal(got, want), k


def test_reference_variable_names_are_the_keys_a_snapshot_would_hold():
    """the TF 1.0.0 names of the CLEVR model's variables, spelled out from the reference's scopes
    (models_clevr/nmn3_model.py:22-49, nmn3_netgen_att.py:69-160, nmn3_modules.py scope= arguments, util/cnn.py:19-25,
    util/empty_safe_conv.py:24-27) -- against n2nmn_amd.spec.variable_shapes, the names Saver.restore looks up."""
    from n2nmn_amd.spec import Dims, variable_shapes
    got = variable_shapes(Dims())
    P = 'neural_module_network/'
    enc, dec = P + 'layout_generation/encoder_decoder/encoder/', P + 'layout_generation/encoder_decoder/decoder/'
    want = {enc + 'embedding_mat': (82, 300), enc + 'encoder_h_transform/weights': (512, 512),
            enc + 'encoder_h_transform/biases': (512,),
            dec + 'embedding_mat': (15, 300), dec + 'go_embedding': (1, 300), dec + 'att_prediction/v': (512,),
            dec + 'att_prediction/weights': (512, 512), dec + 'att_prediction/biases': (512,),
            dec + 'token_prediction/weights': (1024, 15), dec + 'token_prediction/biases': (15,)}
    for base in (enc, dec):
        for cell, rows in ((0, 812), (1, 1024)):
            want[base + 'lstm/multi_rnn_cell/cell_%d/basic_lstm_cell/weights' % cell] = (rows, 2048)
            want[base + 'lstm/multi_rnn_cell/cell_%d/basic_lstm_cell/biases' % cell] = (2048,)
    for k, shp in want.items():
        assert got.get(k) == shp, (k, got.get(k))
    mod = {k for k in got if k.startswith(P + 'layout_execution/module_variables/')}
    assert set(got) == set(want) | mod
    scopes = {k.split('/')[3] for k in mod}
    assert scopes == {'FindModule', 'FindSamePropertyModule', 'TransformModule', 'ExistModule', 'CountModule',
                      'EqualNumModule', 'MoreNumModule', 'LessNumModule', 'SamePropertyModule', 'DescribeModule'}
    assert all(k.endswith('/weights') or k.endswith('/biases') for k in mod)

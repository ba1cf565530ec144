"""DeviceFeeder (n2nmn_amd/data_reader.py): batches of the data reader go through pinned staging
and a copy stream into alternating SuperBuckets; every question's logits equal those of a plain
Engine.forward on the reader's host batch, short last batch included."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import data_reader_cases as DC  # noqa: E402

from n2nmn_amd import synth
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES

pytestmark = pytest.mark.gpu


def test_feeder_fills_alternating_buckets(tmp_path):
    import torch
    from n2nmn_amd.data_reader import DataReader, DeviceFeeder
    from n2nmn_amd.engine import Engine
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.superbucket import SuperBucket
    d = Dims(N=8)
    asm = Assembler(list(CLEVR_MODULE_NAMES))
    imdb, params = DC.build(str(tmp_path / 'imdb'), 'clevr', n=37, H=d.H, W=d.W, D=d.D)
    params = dict(params, batch_size=d.N, T_encoder=d.T_encoder, T_decoder=d.T_decoder, assembler=asm)
    w = synth.make_weights(d, seed=0)

    def reader():
        return DataReader(None, imdb=imdb, shuffle=False, one_pass=True, **params)

    # the word / token ids of the tiny test vocabularies are valid ids of the CLEVR-sized model
    ref_eng = Engine(d, asm)
    ref_eng.load_weights(w)
    want = []
    for b in reader().batches():
        s, _, v = ref_eng.forward(b, use_gt_layout=True, gt_layout=b['gt_layout_batch'])
        want.append((np.asarray(torch.as_tensor(s).cpu()), np.asarray(v)))
    assert [x[0].shape[0] for x in want] == [8, 8, 8, 8, 5]

    buckets = [SuperBucket(d, asm, K=2), SuperBucket(d, asm, K=2)]
    for bk in buckets:
        bk.load_weights(w)
    feeder = DeviceFeeder(reader(), buckets, use_gt_layout=True)
    got, sizes = [], []
    for group in feeder.groups():
        feeder.wait(group)
        bk = group.bucket
        assert bk is buckets[group.index % 2]
        bk.run(use_gt_layout=True)
        sizes.append(len(group.batches))
        for k, b in enumerate(group.batches):
            nb = b['seq_length_batch'].shape[0]
            sc, tok, val = bk.result(k)
            got.append((sc[:nb].cpu().numpy(), val[:nb].cpu().numpy().astype(bool)))
            slot = bk.slot(k)
            assert np.array_equal(slot['input_seq_batch'][:, :nb].cpu().numpy(), b['input_seq_batch'])
            assert np.array_equal(slot['image_feat_batch'][:nb].cpu().numpy(), b['image_feat_batch'])
    assert sizes == [2, 2, 1]
    assert len(got) == len(want)
    for (gs, gv), (ws, wv) in zip(got, want):
        assert np.array_equal(gv, wv.astype(bool))
        assert np.abs(gs - ws).max() <= 1e-5

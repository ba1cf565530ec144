"""Super-bucketing (n2nmn_amd/superbucket.py): K in-flight batches of 64 share every launch; a
question's logits must not depend on the slot it travelled in."""
import numpy as np
import pytest

from oracle import n2nmn_oracle as O
from n2nmn_amd import synth
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
from util import assert_close, t2n

pytestmark = pytest.mark.gpu
NAMES = list(CLEVR_MODULE_NAMES)


@pytest.fixture(scope='module')
def bucket():
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.superbucket import SuperBucket
    d = Dims()
    sb = SuperBucket(d, Assembler(NAMES), K=3)
    w = synth.make_weights(d, seed=0)
    sb.load_weights(w)
    return sb, d, w


def test_slots_equal_single_batches_and_oracle(bucket):
    sb, d, w = bucket
    batches = [synth.make_inputs(d, seed=90 + k, min_len=1) for k in range(3)]
    gts = [synth.template_layout_batch(d, offset=k) for k in range(3)]
    for k in range(3):
        sb.fill(k, batches[k], gts[k])
    sb.run(use_gt_layout=True)
    for k in range(3):
        scores, tokens, validity = [t2n(x).copy() for x in sb.result(k)]
        assert np.array_equal(tokens, gts[k]) and validity.all()
        one, _, _ = sb.engine.forward(batches[k], use_gt_layout=True, gt_layout=gts[k])
        assert_close('slot %d vs single batch' % k, scores, t2n(one), 2e-6)
        if k == 1:
            ref = O.forward(w, NAMES, batches[k], d.T_decoder, d.num_choices, np.float64,
                            use_gt_layout=True, gt_layout=gts[k])
            assert_close('slot vs oracle', scores, ref['scores'], 1e-4)


def test_greedy_layouts_in_a_bucket(bucket):
    sb, d, w = bucket
    batches = [synth.make_inputs(d, seed=95 + k) for k in range(3)]
    for k in range(3):
        sb.fill(k, batches[k])
    sb.run(use_gt_layout=False)
    scores, tokens, validity = [t2n(x).copy() for x in sb.result(2)]
    ref = O.forward(w, NAMES, batches[2], d.T_decoder, d.num_choices, np.float64)
    same = np.all(tokens == ref['dec']['predicted_tokens'], axis=0)
    assert same.mean() > 0.9                     # near-ties may flip a free-running token
    assert validity.all()
    assert_close('scores', scores[same], ref['scores'][same], 1e-4)


def test_slot_bounds(bucket):
    sb, d, w = bucket
    with pytest.raises(ValueError):
        sb.slot(3)

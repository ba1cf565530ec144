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
        # (a 192-row pass runs the staged walker, a single batch the one-workgroup walker: 1e-5,
        # tests/test_gpu_walker.py STAGED_TOL)
        assert_close('slot %d vs single batch' % k, scores, t2n(one), 1e-5)
        if k == 1:
            ref = O.forward(w, NAMES, batches[k], d.T_decoder, d.num_choices, np.float64,
                            use_gt_layout=True, gt_layout=gts[k])
            assert_close('slot vs oracle', scores, ref['scores'], 1e-4)


def test_greedy_layouts_in_a_bucket(bucket):
    """free-running decoder: tokens equal the oracle's up to a question's first near-tie (top-2 margin
    < 1e-3); logits are compared given the GPU's tokens"""
    sb, d, w = bucket
    batches = [synth.make_inputs(d, seed=95 + k) for k in range(3)]
    for k in range(3):
        sb.fill(k, batches[k])
    sb.run(use_gt_layout=False)
    scores, tokens, validity = [t2n(x).copy() for x in sb.result(2)]
    ref = O.forward(w, NAMES, batches[2], d.T_decoder, d.num_choices, np.float64)
    dec = ref['dec']
    sc = np.where(dec['token_validity'], dec['token_scores'], -np.inf)
    top2 = np.sort(sc, axis=2)[:, :, -2:]
    margin = top2[:, :, 1] - top2[:, :, 0]
    for i in range(d.N):
        stop = (tokens[:, i] != dec['predicted_tokens'][:, i]) | (margin[:, i] < 1e-3)
        upto = int(np.argmax(stop)) if stop.any() else d.T_decoder
        assert np.array_equal(tokens[:upto, i], dec['predicted_tokens'][:upto, i])
        if upto < d.T_decoder:
            assert margin[upto, i] < 1e-3, 'token flip at a non-tie'
    assert validity.all()
    forced = O.forward(w, NAMES, batches[2], d.T_decoder, d.num_choices, np.float64,
                       forced_tokens=tokens)
    assert_close('scores', scores, forced['scores'], 1e-4)


def test_two_buckets_on_one_engine_keep_their_own_results(bucket):
    """a worker alternates two buckets on one engine (bench.py, DeviceFeeder): bucket a's results must
    survive bucket b's pass (they used to be views of the engine's reuse buffers)"""
    from n2nmn_amd.superbucket import SuperBucket
    sb, d, w = bucket
    other = SuperBucket(d, sb.engine.assembler, K=3, engine=sb.engine)
    ba, bb = synth.make_inputs(d, seed=70), synth.make_inputs(d, seed=71)
    ga, gb = synth.template_layout_batch(d, offset=1), synth.template_layout_batch(d, offset=2)
    for k in range(3):
        sb.fill(k, ba, ga)
        other.fill(k, bb, gb)
    sb.run(use_gt_layout=True)
    mine = t2n(sb.result(0)[0]).copy()
    other.run(use_gt_layout=True)
    assert np.array_equal(t2n(sb.result(0)[0]), mine)
    assert not np.array_equal(t2n(other.result(0)[0]), mine)
    sb.run(use_gt_layout=True, n_slots=2)          # a narrower pass has result tensors of its own
    assert np.array_equal(t2n(sb.result(1)[0]), mine)
    with pytest.raises(ValueError):
        sb.result(2)


def test_slot_bounds(bucket):
    sb, d, w = bucket
    with pytest.raises(ValueError):
        sb.slot(3)


def test_encoder_h_transform_over_listed_rows_never_reads_a_stale_row(bucket):
    """Passes of >= 8192 (t, question) rows compute encoder_h_transform only for the rows inside their
    question's length (a compacted row list, GemmArgs::m_dev); the rest of the buffer keeps whatever
    an EARLIER pass left there.  Nothing may read it: a pass over batch Y right after a pass over an
    unrelated batch X must give, bit for bit, what a fresh bucket gives for Y -- including the lengths
    at the edges of the attention kernels' 'row len is the bias vector' rule (T - 1 used to read the
    real row T - 1), with ground-truth and with greedy layouts."""
    from n2nmn_amd.superbucket import SuperBucket
    sb, d, w = bucket
    T = d.T_encoder
    xs = [synth.make_inputs(d, seed=300 + k) for k in range(3)]
    ys = []
    for k in range(3):
        y = synth.make_inputs(d, seed=310 + k, min_len=1)
        lens = y['seq_length_batch'].copy()
        lens[:8] = [T, T - 1, T - 2, 1, 2, T, T - 1, 1]
        seq = y['input_seq_batch'].copy()
        seq[np.arange(T)[:, None] >= lens[None, :]] = 0
        ys.append(dict(y, seq_length_batch=lens, input_seq_batch=seq))
    gts = [synth.template_layout_batch(d, offset=k) for k in range(3)]
    fresh = SuperBucket(d, sb.engine.assembler, K=3)
    fresh.load_weights(w)
    for use_gt in (True, False):
        for k in range(3):
            sb.fill(k, xs[k], gts[k] if use_gt else None)
        sb.run(use_gt_layout=use_gt)
        for k in range(3):
            sb.fill(k, ys[k], gts[k] if use_gt else None)
            fresh.fill(k, ys[k], gts[k] if use_gt else None)
        sb.run(use_gt_layout=use_gt)
        fresh.run(use_gt_layout=use_gt)
        for k in range(3):
            a, b = sb.result(k), fresh.result(k)
            assert np.array_equal(t2n(a[1]), t2n(b[1])), (use_gt, k, 'tokens')
            # (greedy layouts nest Transform / FindSameProperty: since round 6 every reachable nesting level has its
            # launch by default, so a question's route -- and with it the summation order of its answer head --
            # depends on its own layout only: bit for bit, whatever the bucket ran before.  VERDICT r5 item 5.)
            assert np.array_equal(t2n(a[0]), t2n(b[0])), (use_gt, k, 'logits')
            assert np.isfinite(t2n(a[0])).all()

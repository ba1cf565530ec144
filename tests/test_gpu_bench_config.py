"""The configuration bench.py times, checked against the oracle (VERDICT r2 "test what you bench").

bench.py's default run drives n2nmn_amd.pipeline.PassPipeline: 2 workers (host thread + stream +
forked context) x 2 super-buckets of KCAP = 16 slots of 64 questions, recurrent-step mode
'throughput' (lstm_tile_kernel at >= 128 rows per pass), the layout walker with deferred pooling,
passes 8 to 16 slots wide, both workers running concurrently.  These tests build exactly that object
and compare EVERY slot of every pass with the fp64 batched oracle (oracle/n2nmn_oracle_batched.py,
pinned to the numpy oracle at 1e-10, which is pinned to the reference's code): logits <= 1e-4 for
teacher-forced (BASELINE configs[1]) and greedy (configs[2]) layouts; greedy tokens under the
top-2-margin rule of SURVEY.md 8(c).  Reference loop: exp_clevr/eval_clevr.py:103-135."""
import numpy as np
import pytest
import torch

from oracle import n2nmn_oracle as O
from oracle import n2nmn_oracle_batched as OB
from n2nmn_amd import synth
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
from util import assert_close, t2n

pytestmark = pytest.mark.gpu
torch.set_num_threads(min(16, torch.get_num_threads()))   # the oracle's small ops crawl on 256 threads
NAMES = list(CLEVR_MODULE_NAMES)
KCAP, S = 16, 2
WIDTHS = [[8, 10], [16, 8]]        # worker 0: bucket 0 then bucket 1; worker 1 likewise, concurrently


# both modes bench.py reports run through the SAME checks inside the suite: the
# exact-fp32 'throughput' mode of `value` and the opt-in split-operand mode of the `bf16x3` key
@pytest.fixture(scope='module', params=['throughput', 'throughput_bf16x3'])
def pipe(request):
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.pipeline import PassPipeline
    d = Dims()
    w = synth.make_weights(d, seed=0)
    p = PassPipeline(d, Assembler(NAMES), w, streams=S, kcap=KCAP,
                     mode=None if request.param == 'throughput' else request.param)
    assert p.mode == request.param and all(wk['engine'].mode == request.param for wk in p.workers)
    host = {}

    def inputs(i):
        # slot 7 of worker 1 / bucket 1 repeats slot 0 of worker 0 / bucket 0 (slot independence)
        seed = 0 if i == (1 * 2 + 1) * KCAP + 7 else i
        host[i] = (synth.make_inputs(d, seed=500 + seed, min_len=1),
                   synth.template_layout_batch(d, offset=seed))
        return host[i][0]
    p.fill_all(inputs, lambda i: host[i][1])
    torch.cuda.synchronize()
    yield p, d, w, host
    p.close()


def _concat(host, si, j, n):
    ids = [(si * 2 + j) * KCAP + k for k in range(n)]
    bs = [host[i][0] for i in ids]
    batch = dict(input_seq_batch=np.concatenate([b['input_seq_batch'] for b in bs], 1),
                 seq_length_batch=np.concatenate([b['seq_length_batch'] for b in bs]),
                 image_feat_batch=np.concatenate([b['image_feat_batch'] for b in bs]))
    return batch, np.concatenate([host[i][1] for i in ids], 1)


def _passes():
    for si in range(S):
        for j, n in enumerate(WIDTHS[si]):
            yield si, j, n


def test_teacher_forced_passes_match_the_oracle_in_every_slot(pipe):
    p, d, w, host = pipe
    p.run(WIDTHS, gt=True)
    w64 = OB.to_torch(w, torch.float64)
    worst = 0.0
    for si, j, n in _passes():
        b = p.bucket(si, j)
        assert b.n_run == n
        scores, tokens, validity = [t2n(x) for x in (b.scores, b.tokens, b.validity)]
        batch, gt = _concat(host, si, j, n)
        ref = OB.forward(w64, NAMES, batch, d.T_decoder, d.num_choices, True, gt)
        assert np.array_equal(tokens, gt) and validity.all() and ref['validity'].all()
        for k in range(n):
            c = slice(k * d.N, (k + 1) * d.N)
            worst = max(worst, assert_close('worker %d bucket %d slot %d' % (si, j, k), scores[c],
                                            ref['scores'][c], 1e-4))
    # the same questions in another slot, bucket and worker: the same logits up to the summation
    # order of the recurrent step (the tile is chosen per step from the lengths of the whole pass)
    a = t2n(p.bucket(0, 0).result(0)[0])
    bb = t2n(p.bucket(1, 1).result(7)[0])
    assert_close('slot independence', a, bb, 2e-6 if p.mode == 'throughput' else 1e-5)
    print('worst |logit - oracle| over %d slots: %.2e' % (sum(map(sum, WIDTHS)), worst))


def test_greedy_passes_match_the_oracle_in_every_slot(pipe):
    """configs[2]: the decoder chooses the layouts.  Tokens must equal the oracle's up to the first
    near-tie (top-2 margin < 1e-3) of a question; logits are compared given the GPU's tokens."""
    p, d, w, host = pipe
    p.run(WIDTHS, gt=False)
    w64 = OB.to_torch(w, torch.float64)
    flips = 0
    for si, j, n in _passes():
        b = p.bucket(si, j)
        scores, tokens, validity = [t2n(x) for x in (b.scores, b.tokens, b.validity)]
        batch, _ = _concat(host, si, j, n)
        free = OB.forward(w64, NAMES, batch, d.T_decoder, d.num_choices)
        ref_tok = free['predicted_tokens']
        sc = np.where(free['s2s']['token_validity'].numpy(), free['s2s']['token_scores'].numpy(), -np.inf)
        top2 = np.sort(sc, axis=2)[:, :, -2:]
        margin = top2[:, :, 1] - top2[:, :, 0]
        for i in range(tokens.shape[1]):
            stop = (tokens[:, i] != ref_tok[:, i]) | (margin[:, i] < 1e-3)
            upto = int(np.argmax(stop)) if stop.any() else d.T_decoder
            assert np.array_equal(tokens[:upto, i], ref_tok[:upto, i])
            if upto < d.T_decoder:
                assert margin[upto, i] < 1e-3, 'token flip at a non-tie'
                flips += int((tokens[:, i] != ref_tok[:, i]).any())
        assert validity.all()               # the automaton only lets valid layouts through
        forced = OB.forward(w64, NAMES, batch, d.T_decoder, d.num_choices, True, tokens)
        for k in range(n):
            c = slice(k * d.N, (k + 1) * d.N)
            assert_close('greedy worker %d bucket %d slot %d' % (si, j, k), scores[c],
                         forced['scores'][c], 1e-4)
    print('questions whose layout differs from the oracle at a near-tie:', flips)


def test_driver_shape_two_streams_of_ten_slots(pipe):
    """`bench.py --steps 20 --warmup 5` as round 2's driver ran it: one 10-slot pass per stream."""
    p, d, w, host = pipe
    for wk in p.workers:
        wk['next'] = 0
    p.run([[10], [10]], gt=True)
    w64 = OB.to_torch(w, torch.float64)
    for si in range(S):
        b = p.bucket(si, 0)
        batch, gt = _concat(host, si, 0, 10)
        ref = OB.forward(w64, NAMES, batch, d.T_decoder, d.num_choices, True, gt)
        assert_close('worker %d' % si, t2n(b.scores), ref['scores'], 1e-4)


def test_plan_splits_batches_into_passes():
    from n2nmn_amd.pipeline import PassPipeline
    assert PassPipeline.split(10, 8, 16) == [10]
    assert PassPipeline.split(16, 8, 16) == [8, 8]
    assert PassPipeline.split(40, 8, 16) == [8, 8, 8, 8, 8]
    assert PassPipeline.split(17, 16, 16) == [9, 8]
    assert PassPipeline.split(0, 8, 16) == []

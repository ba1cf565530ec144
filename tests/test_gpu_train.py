"""Training step (BASELINE.json configs[3], exp_clevr/train_clevr_gt_layout.py:104-130) on the GPU
against the autograd oracle (oracle/n2nmn_oracle_grad.py, fp64): losses, answer logits, EVERY
variable's gradient of total_loss, the gradients of the intermediates, and the Adam update.

Tolerances (fp32 kernels vs an fp64 oracle):
  * forward values: 1e-4 absolute (the forward bar of the north star)
  * gradients: max|got - want| <= GRAD_RTOL * max|want| + GRAD_ATOL per tensor -- gradients are
    sums over up to T*N = 2880 fp32 products, so the bar is relative to the tensor's scale
  * discrete min/max selections: per-node gradients are compared where the oracle's selection gap
    is >= SELECTION_GAP (see _check)
  * Adam: the update applied to the SAME (GPU) gradient must match the fp64 formula to 2e-6
"""
import numpy as np
import pytest

from oracle import n2nmn_oracle_grad as G
from n2nmn_amd import synth
from n2nmn_amd.spec import CLEVR_MODULE_NAMES, Dims
from util import assert_close, t2n

pytestmark = pytest.mark.gpu
NAMES = list(CLEVR_MODULE_NAMES)
GRAD_RTOL = 2e-4
GRAD_ATOL = 1e-7
WD = 5e-6
SELECTION_GAP = 1e-5


def grad_report(got: dict, want: dict):
    """per tensor: (name, max|want|, max|diff|, ok)"""
    rows = []
    for k in sorted(want):
        w = np.asarray(want[k], np.float64)
        g = np.asarray(got[k], np.float64).reshape(w.shape)
        scale = float(np.max(np.abs(w))) if w.size else 0.0
        diff = float(np.max(np.abs(g - w))) if w.size else 0.0
        ok = np.isfinite(g).all() and diff <= GRAD_RTOL * scale + GRAD_ATOL
        rows.append((k, scale, diff, bool(ok)))
    return rows


def format_report(rows):
    return '\n'.join('%-100s scale %.3e diff %.3e %s' % (k[-100:], s, d, 'ok' if ok else 'FAIL')
                     for k, s, d, ok in rows)


@pytest.fixture(scope='module')
def trainer_setup():
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.engine import Engine
    from n2nmn_amd.train import Trainer
    d = Dims(T_decoder=10)                      # train_clevr_gt_layout.py:35
    asm = Assembler(NAMES)
    eng = Engine(d, asm)
    w = synth.make_weights(d, seed=0)
    eng.load_weights(w)
    tr = Trainer(eng, weight_decay=WD)
    return tr, eng, d, asm, w


def _run(tr, d, w, batch, gt):
    tr.engine.load_weights(w)
    tr.forward_backward(batch, gt, reduce=False)
    losses = t2n(tr.losses)
    grads = {k: t2n(v) for k, v in tr.gradients().items()}
    ref_l, ref_g, ex = G.loss_and_grads(w, NAMES, batch, gt.shape[0], d.num_choices, gt, WD)
    return losses, grads, ref_l, ref_g, ex


def _check(tr, d, w, batch, gt):
    losses, grads, ref_l, ref_g, ex = _run(tr, d, w, batch, gt)
    N = batch['input_seq_batch'].shape[1]
    assert_close('scores', t2n(tr.scores), ex['scores'], 1e-4)
    for i, k in enumerate(('avg_sample_loss', 'seq_likelihood_loss', 'l2_reg', 'total_loss')):
        assert abs(losses[i] - ref_l[k]) <= 1e-4 * max(1.0, abs(ref_l[k])), (k, losses[i], ref_l[k])
    Td, T, L, E = gt.shape[0], batch['input_seq_batch'].shape[0], d.lstm_dim, d.embed_dim_txt
    inter_got = {
        'd_word_vecs': t2n(tr.debug_tensor('d_word_vecs', (Td, N, E))),
        'd_token_scores': t2n(tr.debug_tensor('d_token_scores', (Td, N, 16)))[:, :, :d.num_vocab_nmn],
        'd_encoder_outputs': t2n(tr.debug_tensor('d_encoder_outputs', (T, N, L))),
        'd_encoder_h_transformed': t2n(tr.debug_tensor('d_encoder_h_transformed', (T, N, L))),
        'd_scores': t2n(tr.debug_tensor('d_scores', (N, d.num_choices))),
    }
    inter_want = {k: ex[k] for k in inter_got}
    # Discrete selections (tf.minimum/maximum branch, reduce_min/max pixel) are discontinuous: where
    # the oracle's two candidates are closer than fp32 can resolve, the fp32 kernels may route a
    # pixel's gradient to the other candidate (with the synthetic weights, And/Or inputs of one
    # question nearly coincide).  Same protocol as the decoder's argmax: the
    # per-node gradient d_word_vecs is compared on the examples whose smallest selection gap is
    # clear; the variable gradients (sums over all nodes) are compared in full.
    clear = ex['selection_gap'] >= SELECTION_GAP
    assert clear.mean() >= 0.5
    inter_got['d_word_vecs'] = inter_got['d_word_vecs'][:, clear]
    inter_want['d_word_vecs'] = inter_want['d_word_vecs'][:, clear]
    rows = grad_report(inter_got, inter_want) + grad_report(grads, ref_g)
    bad = [r for r in rows if not r[3]]
    assert not bad, 'gradient mismatch:\n' + format_report(rows)
    return rows


def test_gradients_template_layouts(trainer_setup):
    """config 4 inputs: the 10-template layout mix, T_dec = 10, N = 64."""
    tr, eng, d, asm, w = trainer_setup
    batch = synth.make_inputs(d, seed=0)
    gt = synth.template_layout_batch(d)
    _check(tr, d, w, batch, gt)


@pytest.mark.parametrize('seed', [1, 2])
def test_gradients_random_trees(trainer_setup, seed):
    """random valid layouts: every operator incl. And/Or/Scene/EqualNum..., deep trees, images
    shared by several Find-type nodes."""
    tr, eng, d, asm, w = trainer_setup
    batch = synth.make_inputs(d, seed=40 + seed, min_len=1)
    gt = synth.random_valid_layouts(d, asm.P, asm.W, asm.b, seed=seed, max_len=[3, 7][seed - 1])
    _check(tr, d, w, batch, gt)


def test_gradients_ragged_small_batch(trainer_setup):
    """N below the context capacity, shorter T_enc, length-1 questions."""
    tr, eng, d, asm, w = trainer_setup
    small = Dims(T_decoder=10, N=37, T_encoder=17)
    batch = synth.make_inputs(small, seed=7, min_len=1)
    gt = synth.template_layout_batch(small, offset=3)
    _check(tr, d, w, batch, gt)


def test_exact_ties_in_nested_filters(trainer_setup):
    """Length-1 questions give every decoder step the same word vector, so nested Filters compare
    bit-identical attention maps: tf.minimum's tie rule (first argument) decides where the gradient
    goes, and the backward pass has to recompute the forward's value bit for bit to see the tie."""
    tr, eng, d, asm, w = trainer_setup
    small = Dims(T_decoder=10, N=16, T_encoder=9)
    batch = synth.make_inputs(small, seed=9, min_len=1)
    batch['seq_length_batch'][:] = 1
    batch['input_seq_batch'][1:] = 0
    tpl = [['_Scene', '_Filter', '_Filter', '_Filter', '_Exist'],
           ['_Find', '_Filter', '_Filter', '_Count'],
           ['_Scene', '_Filter', '_Filter', '_Find', '_SameProperty']]
    gt = np.array([synth.module_list2tokens(tpl[i % 3], small.T_decoder) for i in range(small.N)],
                  np.int32).T
    _check(tr, d, w, batch, gt)


def test_flat_layout_and_buckets(trainer_setup):
    tr, eng, d, asm, w = trainer_setup
    from n2nmn_amd.spec import num_parameters
    assert tr.numel == num_parameters(d)
    offs = sorted((o, n, k) for k, (o, n, s) in tr.layout.items())
    pos = 0
    for o, n, k in offs:
        assert o == pos, k
        pos += n
    assert pos == tr.numel
    enc = [k for k in tr.layout if '/encoder/' in k]
    assert tr.split == max(tr.layout[k][0] + tr.layout[k][1] for k in enc)
    assert all(tr.layout[k][0] >= tr.split for k in tr.layout if '/encoder/' not in k)


def test_adam_steps_match_oracle_formula(trainer_setup):
    """three optimiser steps: each update equals clip_by_norm + Adam (fp64) applied to the gradient
    the GPU produced at that step; moments carry over."""
    tr, eng, d, asm, w = trainer_setup
    from n2nmn_amd.train import Trainer
    eng.load_weights(w)
    tr2 = Trainer(eng, weight_decay=WD)          # a new Trainer resets the Adam moments
    assert tr2.iteration == 0
    batch = synth.make_inputs(d, seed=3)
    gt = synth.template_layout_batch(d, offset=1)
    cur = {k: np.asarray(v, np.float64) for k, v in w.items()}
    m = {k: np.zeros_like(v) for k, v in cur.items()}
    v = {k: np.zeros_like(x) for k, x in cur.items()}
    first_loss = None
    for step in range(1, 4):
        scale = tr2.forward_backward(batch, gt, reduce=False)
        g = {k: t2n(t).astype(np.float64) for k, t in tr2.gradients().items()}
        loss = float(t2n(tr2.losses)[3])
        first_loss = loss if first_loss is None else first_loss
        tr2.apply(scale)
        cur, m, v = G.adam_step(cur, g, m, v, step)
        got = {k: t2n(t) for k, t in tr2.get_weights().items()}
        for k in cur:
            assert_close('adam[%d] %s' % (step, k), got[k], cur[k], 2e-6)
    assert loss < first_loss          # the objective goes down on a repeated batch


def test_gradient_is_invariant_to_question_order(trainer_setup):
    """size-independent property at the full config-4 size (no oracle involved): the losses are
    batch means, so permuting the questions of a batch must leave every gradient unchanged up to
    fp32 summation order (this also moves every question to a different row block / length rank)."""
    tr, eng, d, asm, w = trainer_setup
    eng.load_weights(w)
    batch = synth.make_inputs(d, seed=11, min_len=1)
    # templates without And / Or / Filter: with the synthetic weights their two candidates nearly
    # coincide, so fp32 summation order alone can flip the selected branch (see _check)
    from n2nmn_amd.spec import CLEVR_LAYOUT_TEMPLATES
    keep = [tpl for tpl in CLEVR_LAYOUT_TEMPLATES if not {'_And', '_Or', '_Filter'} & set(tpl)]
    gt = np.array([synth.module_list2tokens(keep[i % len(keep)], d.T_decoder) for i in range(d.N)],
                  np.int32).T
    tr.forward_backward(batch, gt, reduce=False)
    g0 = t2n(tr.grads).copy()
    l0 = t2n(tr.losses).copy()
    perm = np.random.default_rng(0).permutation(d.N)
    pb = {k: (v[:, perm] if k == 'input_seq_batch' else v[perm]) for k, v in batch.items()}
    tr.forward_backward(pb, gt[:, perm], reduce=False)
    g1 = t2n(tr.grads)
    assert np.abs(t2n(tr.losses) - l0).max() <= 1e-5 * np.abs(l0).max()
    for name, (off, n, shape) in tr.layout.items():
        a, b = g0[off:off + n], g1[off:off + n]
        assert np.abs(a - b).max() <= 2e-4 * np.abs(a).max() + 1e-7, name


def test_single_question_and_full_length_batches(trainer_setup):
    """N = 1 (one 16-row MFMA tile mostly empty) and a batch where no row is ever masked; going
    from a small batch to a larger one also checks that nothing kept from the previous call (the
    activation sequences are laid out with the call's own batch stride) leaks into the next."""
    tr, eng, d, asm, w = trainer_setup
    one = Dims(T_decoder=10, N=1)
    b1 = synth.make_inputs(one, seed=3, min_len=7)
    _check(tr, d, w, b1, synth.template_layout_batch(one, offset=4))   # _Find _FindSameProperty _Count
    full = Dims(T_decoder=10, N=20)
    b2 = synth.make_inputs(full, seed=5, min_len=full.T_encoder)        # every length == T_enc
    assert (b2['seq_length_batch'] == full.T_encoder).all()
    _check(tr, d, w, b2, synth.template_layout_batch(full, offset=2))


def test_training_api_errors(trainer_setup):
    import ctypes as C
    from n2nmn_amd import _lib
    from n2nmn_amd.engine import Engine
    tr, eng, d, asm, w = trainer_setup
    # a fork cannot train; a context without train_enable rejects the training entries
    with pytest.raises(ValueError):
        from n2nmn_amd.train import Trainer
        Trainer(eng.fork())
    e2 = Engine(Dims(T_decoder=10, N=4), asm)
    e2.load_weights(w)
    io = _lib.TrainIO()
    rc = e2._lib.n2nmn_train_forward(e2._ctx, C.byref(io), None, None)
    assert rc < 0 and b'train_forward' in e2._lib.n2nmn_last_error()
    assert e2._lib.n2nmn_grad_numel(e2._ctx) < 0
    # program assembled from another batch size
    batch = synth.make_inputs(d, seed=1)
    gt_small = synth.template_layout_batch(Dims(T_decoder=10, N=8))
    with pytest.raises(ValueError):
        tr.forward_backward(batch, gt_small, reduce=False)


SCHEDULES = {          # keys of n2nmn_debug_set (include/n2nmn.h section 7), set before n2nmn_train_enable
    # the round-2 schedule: every weight gradient after its recurrence, on the caller's stream
    'leaves_after_recurrence': {'train_schedule': '0'},
    # as many time chunks as the encoder's backward takes, unbounded background launches
    'four_chunks_unbounded': {'train_chunks': '60,40,20', 'train_bg_wgs': '0'},
    # no side stream at all
    'single_stream': {'train_overlap': '0'},
}


@pytest.mark.parametrize('name', sorted(SCHEDULES))
def test_every_backward_schedule_gives_the_same_gradients(monkeypatch, name):
    """The weight-gradient GEMMs are leaves of the backward graph; where they run (side stream under
    the encoder's reverse-time pass, chunk by chunk, or after it) is a schedule, not arithmetic.
    Each schedule the library can be switched to (n2nmn_debug_set, read at n2nmn_train_enable) against the oracle,
    on the configuration-4 batch and on a ragged one (short questions: chunks without rows)."""
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.engine import Engine
    from n2nmn_amd.train import Trainer
    d = Dims(T_decoder=10)
    asm = Assembler(NAMES)
    eng = Engine(d, asm)
    for k, v in SCHEDULES[name].items():
        eng.debug_set(k, v)
    with pytest.raises(KeyError):
        eng.debug_set('no_such_switch', '1')
    w = synth.make_weights(d, seed=0)
    eng.load_weights(w)
    tr = Trainer(eng, weight_decay=WD)
    _check(tr, d, w, synth.make_inputs(d, seed=0), synth.template_layout_batch(d))
    small = Dims(T_decoder=10, N=37, T_encoder=17)
    _check(tr, d, w, synth.make_inputs(small, seed=7, min_len=1), synth.template_layout_batch(small, offset=3))
    short = synth.make_inputs(d, seed=3, min_len=1)
    short['seq_length_batch'] = np.minimum(short['seq_length_batch'], 4).astype(np.int32)
    _check(tr, d, w, short, synth.template_layout_batch(d, offset=1))


def test_backward_is_stable_over_repeated_steps(trainer_setup):
    """The default schedule keeps the decoder's weight gradients in flight while the encoder's
    reverse-time pass runs and starts the next forward pass right behind the optimiser: twenty
    forward/backward passes over one batch, every one against the same oracle gradients (a buffer
    shared by two phases, or a missing wait between two steps, shows up as an occasional mismatch)."""
    tr, eng, d, asm, w = trainer_setup
    batch = synth.make_inputs(d, seed=11)
    gt = synth.template_layout_batch(d, offset=2)
    _, _, _, ref_g, _ = _run(tr, d, w, batch, gt)
    for it in range(20):
        tr.forward_backward(batch, gt, reduce=False)
        grads = {k: t2n(v) for k, v in tr.gradients().items()}
        bad = [r for r in grad_report(grads, ref_g) if not r[3]]
        assert not bad, 'iteration %d:\n%s' % (it, format_report(bad))


def test_inference_right_after_an_optimiser_step_sees_the_new_weights(trainer_setup):
    """n2nmn_adam_step updates and re-packs the decoder's and the module network's variables on the
    library's side stream while the caller's stream is already free for the next encoder pass.  An
    inference call issued straight after it (no synchronisation in between) must use the NEW weights
    everywhere: compared with a fresh engine loaded from the trainer's weights, in the default mode
    (walker) and through the level path."""
    from n2nmn_amd.engine import Engine
    tr, eng, d, asm, w = trainer_setup
    eng.load_weights(w)
    batch = synth.make_inputs(d, seed=21)
    gt = synth.template_layout_batch(d, offset=4)
    probe = synth.make_inputs(d, seed=22)
    gt_probe = synth.template_layout_batch(d, offset=5)
    for it in range(3):
        tr.step(batch, gt)
        got, _, _ = eng.forward(probe, use_gt_layout=True, gt_layout=gt_probe)     # no sync before it
        got = t2n(got).copy()
        packed, _ = asm.assemble_packed(gt_probe)
        s2s = eng.seq2seq(probe['input_seq_batch'], probe['seq_length_batch'], use_gt_layout=True,
                          gt_layout=gt_probe)
        lvl = t2n(eng.execute(packed, probe['image_feat_batch'], s2s['word_vecs'])).copy()
        fresh = Engine(d, asm)
        fresh.load_weights({k: t2n(v) for k, v in tr.get_weights().items()})
        want, _, _ = fresh.forward(probe, use_gt_layout=True, gt_layout=gt_probe)
        want = t2n(want)
        assert_close('walker path, iteration %d' % it, got, want, 2e-6)
        assert_close('level path, iteration %d' % it, lvl, want, 2e-5)
    eng.load_weights(w)


def test_gradients_at_128_rows_in_throughput_mode():
    """A training batch of 128 in 'throughput' mode takes the recurrent steps of its forward pass
    through lstm_tile_kernel (whose training epilogue keeps gates / cell / hidden sequences in the
    ORIGINAL row order and does not skip finished row blocks) and its backward through 8 row blocks per
    job: same parity bar as the reference's batch of 64."""
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.engine import Engine
    from n2nmn_amd.train import Trainer
    d = Dims(N=128, T_decoder=10)
    asm = Assembler(NAMES)
    eng = Engine(d, asm)
    w = synth.make_weights(d, seed=0)
    eng.load_weights(w)
    eng.set_mode('throughput')
    tr = Trainer(eng, weight_decay=WD)
    _check(tr, d, w, synth.make_inputs(d, seed=31, min_len=1), synth.template_layout_batch(d, offset=6))

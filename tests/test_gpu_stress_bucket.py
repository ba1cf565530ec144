"""Randomised cross-check (formerly tools/stress_bucket.py): super-bucketed passes (random slot
count and batch size, ragged question lengths incl. length 1, teacher-forced or greedy layouts, all three
recurrent-step modes -- 'throughput' runs lstm_tile_kernel from 128 rows up, 'throughput_bf16x3' the
split-operand kernels) against
one-batch-at-a-time passes of a separate engine in the default mode.  Logits must agree to 2e-5,
tokens / validity exactly (greedy: up to a question's first near-tie of two token logits, < 1e-5).  Reference loop: exp_clevr/eval_clevr.py:103-135."""
import numpy as np
import pytest

from n2nmn_amd import synth
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('trial', range(8))
def test_random_bucket_equals_single_batches(trial):
    import torch
    from n2nmn_amd.engine import Engine
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.superbucket import SuperBucket
    rng = np.random.default_rng(2026 + trial)
    asm = Assembler(list(CLEVR_MODULE_NAMES))
    Nb = int(rng.choice([8, 16, 40, 64]))
    K = int(rng.integers(1, 9))
    d = Dims(N=Nb)
    w = synth.make_weights(d, seed=trial)
    one = Engine(d, asm)
    one.load_weights(w)
    sb = SuperBucket(d, asm, K)
    sb.load_weights(w)
    # (trials 0, 3, 6: 'throughput'; 1, 4, 7: 'latency'; 2, 5: the opt-in split-operand mode)
    sb.engine.set_mode(('throughput', 'latency', 'throughput_bf16x3')[trial % 3])
    use_gt = bool(rng.integers(0, 2))
    batches, gts = [], []
    for k in range(K):
        b = synth.make_inputs(d, seed=1000 * trial + k, min_len=1)
        lens = b['seq_length_batch'].copy()
        lens[rng.integers(0, Nb, size=max(1, Nb // 8))] = 1          # very short questions
        seq = b['input_seq_batch'].copy()
        seq[np.arange(d.T_encoder)[:, None] >= lens[None, :]] = 0
        b = dict(b, seq_length_batch=lens, input_seq_batch=seq)
        batches.append(b)
        gts.append(synth.template_layout_batch(d, offset=int(rng.integers(0, 10))))
        sb.fill(k, b, gts[-1] if use_gt else None)
    sb.run(use_gt_layout=use_gt)
    worst = 0.0
    for k in range(K):
        s1, t1, v1 = one.forward(batches[k], use_gt_layout=use_gt, gt_layout=gts[k] if use_gt else None)
        s1 = torch.as_tensor(s1).cpu().numpy().copy()
        t1, v1 = np.asarray(t1), np.asarray(v1).astype(bool)
        s2, t2, v2 = sb.result(k)
        s2, t2, v2 = s2.cpu().numpy(), t2.cpu().numpy(), v2.cpu().numpy().astype(bool)
        same = np.ones(Nb, bool)
        if not use_gt and not np.array_equal(t1, t2):
            # The two paths run different recurrent-step kernels (summation order: ~1e-7 on a token
            # logit), so a greedy choice between two tokens closer than that may legitimately differ
            # (the decoder's argmax is compared up to a question's first near-tie).
            # A question whose layouts differ must differ FIRST at such a near-tie of the single-batch
            # engine's own token scores; its logits are then not comparable and are left out.
            dbg = one.seq2seq(batches[k]['input_seq_batch'], batches[k]['seq_length_batch'], debug=True)
            ts = dbg['token_scores'].cpu().numpy()
            for i in np.nonzero((t1 != t2).any(axis=0))[0]:
                step = int(np.argmax(t1[:, i] != t2[:, i]))
                gap = abs(float(ts[step, i, t1[step, i]] - ts[step, i, t2[step, i]]))
                assert gap < 1e-5, (trial, k, int(i), step, gap, 'token flip at a non-tie')
                same[i] = False
            assert same.mean() >= 0.9, (trial, k, 'too many near-ties to be ties')
        err = float(np.abs(s1[same] - s2[same]).max())
        worst = max(worst, err)
        assert err <= 2e-5, (trial, k, err)
        assert np.array_equal(v1[same], v2[same]), (trial, k)
    print('trial %d: batch %d x K=%d, gt=%s, worst |logit difference| %.2e' % (trial, Nb, K, use_gt, worst))

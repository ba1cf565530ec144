"""Randomised cross-check (formerly tools/stress_bucket.py): super-bucketed passes (random slot
count and batch size, ragged question lengths incl. length 1, teacher-forced or greedy layouts, both
recurrent-step modes -- 'throughput' runs lstm_tile_kernel from 128 rows up) against
one-batch-at-a-time passes of a separate engine in the default mode.  Logits must agree to 2e-5,
tokens / validity exactly.  Reference loop: exp_clevr/eval_clevr.py:103-135."""
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
    sb.engine.set_mode('throughput' if trial % 2 == 0 else 'latency')
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
        s2, t2, v2 = sb.result(k)
        err = float(np.abs(torch.as_tensor(s1).cpu().numpy() - s2.cpu().numpy()).max())
        worst = max(worst, err)
        assert err <= 2e-5, (trial, k, err)
        assert np.array_equal(np.asarray(t1), t2.cpu().numpy()), (trial, k, 'tokens')
        assert np.array_equal(np.asarray(v1).astype(bool), v2.cpu().numpy().astype(bool)), (trial, k)
    print('trial %d: batch %d x K=%d, gt=%s, worst |logit difference| %.2e' % (trial, Nb, K, use_gt, worst))

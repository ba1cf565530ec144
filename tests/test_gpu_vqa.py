"""BASELINE.json configs[4]: the models_vqa path (exp_vqa/eval_vqa2.py:103-137) on the GPU at the
reference's dimensions (14x14x2048 ResNet features + 2 coordinate channels, map_dim 1024,
lstm_dim 1000, 17742-word vocabulary, 3001 answers, T_enc 26, T_dec 13) against the fp64 oracle.
Answer logits (scores_nmn + question prior) within 1e-4; decoder protocol as for CLEVR: token
scores compared, the oracle's tokens forced for the module network."""
import numpy as np
import pytest

from oracle import n2nmn_oracle as O
from n2nmn_amd import synth, vqa
from util import assert_close, t2n

pytestmark = pytest.mark.gpu
TOL = 1e-4
NQ = 24          # questions per test batch (the oracle runs 2050x1024 1x1 convs per Find in fp64)


@pytest.fixture(scope='module')
def vqa_setup():
    d = vqa.VQADims(N=NQ)
    eng = vqa.VQAEngine(d)
    w = synth.make_weights_from_shapes(vqa.vqa_variable_shapes(d), seed=0)
    eng.load_weights(w)
    return eng, d, w


def _batch(d, seed):
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, d.T_encoder + 1, size=d.N).astype(np.int32)
    seq = rng.integers(0, d.num_vocab_txt, size=(d.T_encoder, d.N)).astype(np.int32)
    seq[np.arange(d.T_encoder)[:, None] >= lens[None, :]] = 0
    feat = np.maximum(rng.standard_normal((d.N, d.H, d.W, d.D)), 0).astype(np.float32)
    return dict(input_seq_batch=seq, seq_length_batch=lens, image_feat_batch=feat)


# the layouts of exp_vqa/data/v2_gt_layout_*.npy (24 unique; these cover every module and arity)
LAYOUTS = (['_Find', '_Describe'], ['_Find', '_Find', '_And', '_Describe'],
           ['_Find', '_Transform', '_Describe'], ['_Find', '_Transform', '_Find', '_And', '_Describe'],
           ['_Find', '_Find', '_Find', '_And', '_And', '_Describe'],
           ['_Find', '_Transform', '_Transform', '_Describe'])


def _gt(eng, d):
    asm = eng.assembler
    return np.array([asm.module_list2tokens(LAYOUTS[i % len(LAYOUTS)], d.T_decoder)
                     for i in range(d.N)], np.int32).T


def test_vqa_gt_layouts(vqa_setup):
    eng, d, w = vqa_setup
    batch = _batch(d, 0)
    gt = _gt(eng, d)
    ref = O.forward_vqa(w, batch, d.T_decoder, d.num_choices, np.float64, use_gt_layout=True,
                        gt_layout=gt)
    assert ref['validity'].all()
    scores, tokens, validity = eng.forward(batch, use_gt_layout=True, gt_layout=gt)
    assert np.array_equal(tokens, gt) and validity.all()
    got = t2n(scores).copy()                      # the engine reuses its output buffer
    assert_close('scores', got, ref['scores'], TOL)
    assert np.array_equal(np.argmax(got, 1), np.argmax(ref['scores'], 1))
    nmn, _, _ = eng.forward(batch, use_gt_layout=True, gt_layout=gt, use_qpn=False)
    assert_close('scores_nmn', t2n(nmn), ref['scores_nmn'], TOL)


def test_vqa_decoder_and_forced_greedy_layouts(vqa_setup):
    eng, d, w = vqa_setup
    batch = _batch(d, 5)
    ref = O.forward_vqa(w, batch, d.T_decoder, d.num_choices, np.float64)
    s2s = eng.engine.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], d.T_decoder,
                             forced_tokens=ref['dec']['predicted_tokens'], debug=True)
    L = d.lstm_dim
    assert_close('token_scores', t2n(s2s['token_scores']), ref['dec']['token_scores'], TOL)
    assert_close('encoder_outputs', t2n(s2s['encoder_outputs'])[:, :, :L], ref['enc']['outputs'], TOL)
    assert np.abs(t2n(s2s['encoder_outputs'])[:, :, L:]).max() == 0.0      # padded units stay 0
    assert_close('word_vecs', t2n(s2s['word_vecs']), ref['dec']['word_vecs'], TOL)
    scores, tokens, validity = eng.forward(batch, forced_tokens=ref['dec']['predicted_tokens'])
    assert validity.all()
    assert_close('scores', t2n(scores), ref['scores'], TOL)


def test_vqa_rejects_clevr_only_operators(vqa_setup):
    eng, d, w = vqa_setup
    import torch
    e = eng.engine
    feat = torch.zeros((1, d.H, d.W, eng.idims.D), device=e.device)
    wv = torch.zeros((d.T_decoder, 1, d.embed_dim_txt), device=e.device)
    att = torch.zeros((1, d.H, d.W, 1), device=e.device)
    with pytest.raises(KeyError):
        e.module_forward('_Count', [att], [0], [0], feat, wv)


def test_vqa_pass_of_several_batches_equals_single_batches(vqa_setup):
    """bench.py's config5.passes: several client batches in one launch of >= 128 rows, recurrent step
    in 'throughput' mode (lstm_tile_kernel at lstm_dim 1024, gemm_dma_kernel on the 2064 -> 1024
    conv_image).  Slot k of the pass against the same questions served alone by the (oracle-pinned)
    single-batch engine of this module, and slot 0 against the oracle itself."""
    eng, d, w = vqa_setup
    K = 6                                        # 144 rows: tile kernel on (>= 128), ragged last block
    big = vqa.VQADims(N=K * d.N)
    eng_big = vqa.VQAEngine(big)
    eng_big.load_weights(w)
    eng_big.engine.set_mode('throughput')
    parts = [_batch(d, 100 + k) for k in range(K)]
    gts = [np.roll(_gt(eng, d), k, axis=1) for k in range(K)]
    cat = dict(input_seq_batch=np.concatenate([p['input_seq_batch'] for p in parts], 1),
               seq_length_batch=np.concatenate([p['seq_length_batch'] for p in parts]),
               image_feat_batch=np.concatenate([p['image_feat_batch'] for p in parts], 0))
    gt_cat = np.ascontiguousarray(np.concatenate(gts, 1))
    scores, tokens, validity = eng_big.forward(cat, use_gt_layout=True, gt_layout=gt_cat)
    assert validity.all() and np.array_equal(tokens, gt_cat)
    got = t2n(scores).copy()
    for k in range(K):
        alone, _, _ = eng.forward(parts[k], use_gt_layout=True, gt_layout=np.ascontiguousarray(gts[k]))
        assert_close('slot %d' % k, got[k * d.N:(k + 1) * d.N], t2n(alone), 2e-5)
    ref = O.forward_vqa(w, parts[0], d.T_decoder, d.num_choices, np.float64, use_gt_layout=True,
                        gt_layout=gts[0])
    assert_close('slot 0 vs oracle', got[:d.N], ref['scores'], TOL)


def test_resident_feature_slab_equals_add_coords_per_pass(vqa_setup):
    """VQAEngine.feature_slab: the input slab [N, 14, 14, 2064] carries the two coordinate channels of
    add_spatial_coordinate_map (models_vqa/nmn3_modules.py:11-31; constants) and the zero padding, written once;
    a client fills the 2048 image channels.  A pass on the slab is bit-identical to a pass that appends the
    coordinates itself (n2nmn_add_coords per call), for two different batches written into the same slab, and a
    tensor of slab width that is not a slab is refused."""
    import torch
    eng, d, w = vqa_setup
    gt = _gt(eng, d)
    slab = eng.feature_slab(d.N)
    for seed in (3, 4):
        batch = _batch(d, seed)
        want, _, _ = eng.forward(batch, use_gt_layout=True, gt_layout=gt)
        want = t2n(want).copy()
        slab[..., :d.D].copy_(torch.as_tensor(batch['image_feat_batch']).to(slab.device))
        with_c = t2n(eng.features_with_coords(batch['image_feat_batch'])).copy()
        assert np.array_equal(t2n(slab), with_c), 'slab layout differs from n2nmn_add_coords'
        got, tokens, validity = eng.forward(dict(batch, image_feat_batch=slab), use_gt_layout=True, gt_layout=gt)
        assert validity.all() and np.array_equal(t2n(got), want)
    with pytest.raises(ValueError):
        eng.forward(dict(batch, image_feat_batch=torch.zeros_like(slab)), use_gt_layout=True, gt_layout=gt)


def test_feature_slab_is_recognised_by_identity_not_by_address(vqa_setup):
    """the slab registry used to be data_ptr -> shape with no reference to the slab.  A slab the caller has
    dropped must stop being accepted even if the allocator hands its address to the next tensor of the same shape;
    whole rows of a live slab (slab[k:]) are a slab too."""
    import gc
    import torch
    eng, d, w = vqa_setup
    gt = _gt(eng, d)
    batch = _batch(d, 5)
    slab = eng.feature_slab(d.N)
    slab[..., :d.D].copy_(torch.as_tensor(batch['image_feat_batch']).to(slab.device))
    full, _, _ = eng.forward(dict(batch, image_feat_batch=slab), use_gt_layout=True, gt_layout=gt)
    full = t2n(full).copy()
    k = 3                                            # the last N - k questions as a view of the slab
    part = {kk: (v[:, k:] if kk in ('input_seq_batch',) else v[k:]) for kk, v in batch.items() if kk != 'image_feat_batch'}
    got, _, _ = eng.forward(dict(part, image_feat_batch=slab[k:]), use_gt_layout=True, gt_layout=gt[:, k:])
    assert np.abs(t2n(got) - full[k:]).max() <= 1e-5
    ptr, shape = slab.data_ptr(), tuple(slab.shape)
    del slab, got
    gc.collect()
    torch.cuda.synchronize()
    other = torch.zeros(shape, dtype=torch.float32, device=eng.engine.device)      # (usually the very same address)
    print('the freed slab address was %s' % ('reused' if other.data_ptr() == ptr else 'not reused'))
    with pytest.raises(ValueError, match='still alive'):
        eng.forward(dict(batch, image_feat_batch=other), use_gt_layout=True, gt_layout=gt)

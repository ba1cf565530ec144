"""GPU parity of the individual kernels behind the C-ABI against the fp64 oracle.
Tolerance: 1e-4 absolute on every fp32 output (BASELINE.json north_star)."""
import numpy as np
import pytest

from oracle import n2nmn_oracle as O
from util import assert_close, t2n

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize('M,N,K', [(64, 64, 32), (82, 2048, 300), (150, 250, 512), (1, 15, 4),
                                   (257, 130, 100), (2880, 512, 512), (19250, 250, 512),
                                   (20011, 512, 300), (19200, 200, 64), (4100, 1024, 2064),
                                   (2048, 100, 36), (2049, 384, 512)])
def test_gemm_matches_fp64(clevr_engine, M, N, K):
    """(tall cases included: conv_image at super-bucket sizes has M = 38400 and more.  M >= 2048 with
    a packed width that is a multiple of 128 runs gemm_dma_kernel -- ragged last row tile, K tails
    (300, 2064, 36: not multiples of the 32-k stage), models_vqa's 2064 x 1024 -- the rest gemm_pk)"""
    eng = clevr_engine[0]
    rng = np.random.default_rng(M * 7 + N)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)   # asymmetric: catches transposes
    bias = rng.standard_normal(N).astype(np.float32)
    C = t2n(eng.gemm(A, B, bias))
    ref = A.astype(np.float64) @ B.astype(np.float64) + bias
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    assert np.max(np.abs(C - ref) / (scale + 1)) < 5e-6


def _rand_case(d, rng, nb, n_full=None):
    n_full = d.N if n_full is None else n_full
    feat = np.maximum(rng.standard_normal((n_full, d.H, d.W, d.D)), 0).astype(np.float32)
    wv = (rng.standard_normal((d.T_decoder, n_full, d.embed_dim_txt)) * 0.3).astype(np.float32)
    t_idx = rng.integers(0, d.T_decoder, nb).astype(np.int32)
    b_idx = rng.integers(0, n_full, nb).astype(np.int32)
    a0 = (rng.standard_normal((nb, d.H, d.W, 1)) * 2).astype(np.float32)
    a1 = (rng.standard_normal((nb, d.H, d.W, 1)) * 2).astype(np.float32)
    return feat, wv, t_idx, b_idx, a0, a1


def _oracle_module(w, name, feat, wv, t_idx, b_idx, a0, a1, d):
    w64 = O._cast(w, np.float64)
    f = feat.astype(np.float64)[b_idx]
    txt = wv.astype(np.float64).reshape(-1, wv.shape[-1])[t_idx * wv.shape[1] + b_idx]
    a0 = a0.astype(np.float64); a1 = a1.astype(np.float64)
    return {
        '_Scene': lambda: O.m_scene(w64, len(t_idx), d.H, d.W, np.float64),
        '_Find': lambda: O.m_find(w64, f, txt),
        '_Filter': lambda: O.m_filter(w64, a0, f, txt),
        '_FindSameProperty': lambda: O.m_find_same_property(w64, a0, f, txt),
        '_Transform': lambda: O.m_transform(w64, a0, txt),
        '_And': lambda: O.m_and(a0, a1),
        '_Or': lambda: O.m_or(a0, a1),
        '_Exist': lambda: O.m_exist(w64, a0),
        '_Count': lambda: O.m_count(w64, a0),
        '_EqualNum': lambda: O.m_equal_num(w64, a0, a1),
        '_MoreNum': lambda: O.m_more_num(w64, a0, a1),
        '_LessNum': lambda: O.m_less_num(w64, a0, a1),
        '_SameProperty': lambda: O.m_same_property(w64, a0, a1, f, txt),
        '_Describe': lambda: O.m_describe(w64, a0, f, txt),
    }[name]()


ALL_OPS = ['_Scene', '_Find', '_Filter', '_FindSameProperty', '_Transform', '_And', '_Or', '_Exist',
           '_Count', '_EqualNum', '_MoreNum', '_LessNum', '_SameProperty', '_Describe']


@pytest.mark.parametrize('name', ALL_OPS)
@pytest.mark.parametrize('nb', [1, 7, 64])
def test_module_operator(clevr_engine, name, nb):
    """Modules.<X>Module(...) through n2nmn_module_forward vs the oracle's restatement of
    models_clevr/nmn3_modules.py."""
    from n2nmn_amd.nmn3_modules import Modules
    from n2nmn_amd.spec import MODULE_INPUT_NUM
    eng, d, asm, w = clevr_engine
    rng = np.random.default_rng(ALL_OPS.index(name) * 100 + nb)
    feat, wv, t_idx, b_idx, a0, a1 = _rand_case(d, rng, nb)
    import torch
    mods = Modules(torch.as_tensor(feat).cuda(), torch.as_tensor(wv).cuda(), d.num_choices,
                   engine=eng)
    method = getattr(mods, name[1:] + 'Module')
    ins = [a0, a1][:MODULE_INPUT_NUM[name]]
    got = t2n(method(*ins, t_idx, b_idx))
    want = _oracle_module(w, name, feat, wv, t_idx, b_idx, a0, a1, d)
    assert_close(name, got, want, TOL)


def test_modules_positional_constructor_of_the_reference():
    """`Modules(image_feat_grid, word_vecs, num_choices)` as exp_shapes/visualize_shapes.ipynb (cell 6) and
    models_clevr/nmn3_model.py:52 write it -- no engine argument: the operators use the most recently built model's
    weights, as the reference's use "the graph's" (VERDICT r5 weak #12)."""
    import torch
    from n2nmn_amd import synth
    from n2nmn_amd.engine import Engine
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.nmn3_modules import Modules
    from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
    d = Dims(N=8)
    eng = Engine(d, Assembler(list(CLEVR_MODULE_NAMES)))
    w = synth.make_weights(d, seed=4)
    eng.load_weights(w)
    assert Engine.latest() is eng
    rng = np.random.default_rng(17)
    feat, wv, t_idx, b_idx, a0, a1 = _rand_case(d, rng, 5)
    mods = Modules(torch.as_tensor(feat).cuda(), torch.as_tensor(wv).cuda(), d.num_choices)
    assert mods.engine is eng
    got = t2n(mods.FindModule(t_idx, b_idx))
    assert_close('_Find', got, _oracle_module(w, '_Find', feat, wv, t_idx, b_idx, a0, a1, d), TOL)
    got = t2n(mods.DescribeModule(a0, t_idx, b_idx))
    assert_close('_Describe', got, _oracle_module(w, '_Describe', feat, wv, t_idx, b_idx, a0, a1, d), TOL)


def test_module_small_feature_batch(clevr_engine):
    """N_full smaller than the context capacity, repeated batch indices."""
    eng, d, asm, w = clevr_engine
    rng = np.random.default_rng(5)
    feat, wv, t_idx, b_idx, a0, a1 = _rand_case(d, rng, 9, n_full=3)
    b_idx[:] = [0, 0, 0, 1, 2, 2, 1, 0, 2]
    for name in ('_Find', '_Describe', '_FindSameProperty', '_SameProperty'):
        from n2nmn_amd.spec import MODULE_INPUT_NUM
        ins = [a0, a1][:MODULE_INPUT_NUM[name]]
        got = t2n(eng.module_forward(name, ins, t_idx, b_idx, feat, wv))
        want = _oracle_module(w, name, feat, wv, t_idx, b_idx, a0, a1, d)
        assert_close(name, got, want, TOL)


def test_module_empty_batch_and_errors(clevr_engine):
    """Fold generates zero-size batches (util/empty_safe_conv.py); here they are a no-op."""
    eng, d, asm, w = clevr_engine
    rng = np.random.default_rng(6)
    feat, wv, t_idx, b_idx, a0, a1 = _rand_case(d, rng, 0)
    out = eng.module_forward('_Find', [], t_idx, b_idx, feat, wv)
    assert tuple(out.shape) == (0, d.H, d.W, 1)
    with pytest.raises(KeyError):
        eng.module_forward('_Nope', [], t_idx, b_idx, feat, wv)
    with pytest.raises(ValueError):
        eng.module_forward('_And', [a0], t_idx, b_idx, feat, wv)
    feat, wv, t_idx, b_idx, a0, a1 = _rand_case(d, rng, 2)
    b_idx[0] = d.N + 3
    with pytest.raises(ValueError):
        eng.module_forward('_Find', [], t_idx, b_idx, feat, wv)


def test_attention_extremes(clevr_engine):
    """Spatial softmax with large logits (max-subtraction path) and constant maps (Scene)."""
    eng, d, asm, w = clevr_engine
    rng = np.random.default_rng(7)
    feat, wv, t_idx, b_idx, a0, a1 = _rand_case(d, rng, 4)
    a0[0] = 80.0 * np.sign(a0[0]); a0[1] = 3.0; a0[2, 4, 7, 0] = 500.0; a0[3] = -1e4
    for name in ('_Describe', '_Exist', '_Count', '_FindSameProperty'):
        got = t2n(eng.module_forward(name, [a0], t_idx, b_idx, feat, wv))
        want = _oracle_module(w, name, feat, wv, t_idx, b_idx, a0, a1, d)
        scale = max(1.0, float(np.abs(want).max()))
        assert_close(name, got / scale, want / scale, TOL)

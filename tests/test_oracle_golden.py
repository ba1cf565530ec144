"""Pins the ORACLE's integer / host logic to golden vectors produced by the reference's own numpy
code (tests/golden/make_assembler_golden.py ran models_clevr/nmn3_assembler.py and
models_vqa/nmn3_assembler.py with TF stubbed)."""
import numpy as np

from oracle import n2nmn_oracle as O


def test_validity_mats_clevr(golden):
    g = golden['clevr']
    P, W, b = O.build_validity_mats(g['module_names'])
    assert np.array_equal(P, np.array(g['P']))
    assert np.array_equal(W, np.array(g['W']))
    assert np.array_equal(b, np.array(g['b']))


def test_validity_mats_vqa(golden):
    g = golden['vqa']
    P, W, b = O.build_validity_mats(g['module_names'])
    assert np.array_equal(P, np.array(g['P']))
    assert np.array_equal(W, np.array(g['W']))
    assert np.array_equal(b, np.array(g['b']))
    # SURVEY Appendix B spot values
    assert P.T.tolist() == [[1, 0, -1, -1, 0], [0, 0, 0, 1, 0], [-1, -1, -1, -1, -1]]


def test_assemble_all_golden_cases(golden):
    g = golden['clevr']
    names = g['module_names']
    total = 0
    for case in g['cases']:
        toks = np.array(case['tokens'], np.int32)
        exprs, validity = O.assemble(names, toks)
        assert validity.tolist() == case['validity'], case['tag']
        assert exprs == case['exprs'], case['tag']
        total += toks.shape[1]
    assert total > 1000


def test_vqa_gt_layouts(golden):
    g = golden['vqa']
    toks = np.array(g['gt_tokens_T20'], np.int32)
    exprs, validity = O.assemble(g['module_names'], toks)
    assert validity.all() and validity.tolist() == g['gt_validity']
    assert exprs == g['gt_exprs']
    assert len(g['gt_layouts']) == 24


def test_first_valid_tokens(golden):
    """from X=[0,0,T] only _Scene/_Find are allowed, for T in {6,10,20}."""
    names = golden['clevr']['module_names']
    P, W, b = O.build_validity_mats(names)
    for T in (6, 10, 20):
        ok = O.valid_tokens(np.array([[0, 0, T]]), W, b)[0]
        assert [names[i] for i in np.nonzero(ok)[0]] == ['_Scene', '_Find']

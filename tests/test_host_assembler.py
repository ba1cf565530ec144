"""Product host logic (no GPU): n2nmn_amd.nmn3_assembler.Assembler, whose RPN decoding runs in the
C++ scheduler behind the C-ABI, against the reference's golden vectors."""
import numpy as np
import pytest

from n2nmn_amd.nmn3_assembler import Assembler, PackedLayouts, build_validity_mats
from n2nmn_amd.spec import CLEVR_MODULE_NAMES, OP_CODE, MODULE_INPUT_NUM, INVALID_EXPR


@pytest.fixture(scope='module')
def asm():
    return Assembler(list(CLEVR_MODULE_NAMES))


def test_attributes_match_reference(golden, asm):
    g = golden['clevr']
    assert asm.module_names == g['module_names']
    assert asm.EOS_idx == g['EOS_idx'] == 14
    assert asm.num_vocab_nmn == 15
    assert asm.name2idx_dict['_Describe'] == 13
    assert np.array_equal(asm.P, np.array(g['P'])) and asm.P.dtype == np.int32
    assert np.array_equal(asm.W, np.array(g['W']))
    assert np.array_equal(asm.b, np.array(g['b']))


def test_vqa_vocabulary(golden):
    g = golden['vqa']
    P, W, b = build_validity_mats(g['module_names'])
    assert np.array_equal(P, np.array(g['P']))
    assert np.array_equal(W, np.array(g['W']))
    assert np.array_equal(b, np.array(g['b']))
    vasm = Assembler(g['module_names'])
    toks = np.array(g['gt_tokens_T20'], np.int32)
    exprs, validity = vasm.assemble(toks)
    assert validity.tolist() == g['gt_validity']
    assert list(exprs) == g['gt_exprs']


def test_assemble_matches_reference_on_every_case(golden, asm):
    for case in golden['clevr']['cases']:
        toks = np.array(case['tokens'], np.int32)
        exprs, validity = asm.assemble(toks)
        assert validity.dtype == bool
        assert validity.tolist() == case['validity'], case['tag']
        assert list(exprs) == case['exprs'], case['tag']   # incl. expr_str / error of INVALID_EXPR


def test_kats(asm):
    T = 10
    toks = np.array([asm.module_list2tokens(m, T) for m in (
        ['_Find', '_Transform', '_Filter', '_Describe'], ['_Find', '_And', '_Count'],
        ['_Scene', '_Describe', '_Find'])], np.int32).T
    exprs, validity = asm.assemble(toks)
    assert validity.tolist() == [True, False, False]
    e = exprs[0]
    assert (e['module'], e['time_idx'], e['batch_idx'], e['output_type']) == ('_Describe', 3, 0, 'ans')
    assert e['input_0']['module'] == '_Filter' and e['input_0']['time_idx'] == 2
    assert e['input_0']['input_0']['module'] == '_Transform'
    assert e['input_0']['input_0']['input_0'] == {
        'module': '_Find', 'output_type': 'att', 'time_idx': 0, 'batch_idx': 0}
    assert exprs[1] == {'module': INVALID_EXPR, 'error': 'not enough input for _And',
                        'expr_str': '_Find _And _Count' + ' <eos>' * 7}
    assert exprs[2]['error'] == 'final stack size not equal to 1 (2 remains)'
    no_eos = np.array([[1, 4, 4, 4]], np.int32).T
    exprs, validity = asm.assemble(no_eos)
    assert exprs[0]['error'] == 'cannot find <eos>' and not validity[0]


def test_input_order(asm):
    """input_1 is the most recently pushed attention (nmn3_assembler.py:194-199)."""
    toks = np.array([asm.module_list2tokens(['_Find', '_Scene', '_EqualNum'], 6)], np.int32).T
    exprs, _ = asm.assemble(toks)
    assert exprs[0]['input_0']['module'] == '_Find' and exprs[0]['input_1']['module'] == '_Scene'


def test_module_list2tokens_contract(golden, asm):
    assert asm.module_list2tokens(['_Find', '_Count'], 4) == [1, 8, 14, 14]
    assert asm.module_list2tokens(['_Find', '_Count']) == [1, 8]
    with pytest.raises(ValueError) as ei:
        asm.module_list2tokens(['_Find'] * 10, 10)
    assert str(ei.value) == golden['clevr']['list2tokens_error']
    with pytest.raises(KeyError):
        asm.module_list2tokens(['_Nope'], 4)


def test_token_out_of_range_is_an_error(asm):
    with pytest.raises(ValueError):
        asm.assemble_packed(np.array([[99, 14]], np.int32).T)


def test_packed_program_structure(asm):
    from n2nmn_amd.synth import template_layout_batch
    from n2nmn_amd.spec import Dims
    d = Dims()
    toks = template_layout_batch(d)
    packed, validity = asm.assemble_packed(toks)
    assert validity.all()
    nodes = packed.nodes()
    # one node per non-<eos> token
    assert len(nodes) == int((toks != asm.EOS_idx).sum()) == packed.num_nodes
    assert packed.num_rows == d.N
    for i, nd in enumerate(nodes):
        k = MODULE_INPUT_NUM[CLEVR_MODULE_NAMES[nd['op']]]
        ins = [nd['in0'], nd['in1']]
        for j in range(2):
            if j < k:
                assert 0 <= ins[j] < i                      # topological
                assert nodes[ins[j]]['batch_idx'] == nd['batch_idx']
                # a consumer never runs before its inputs are complete
                assert nodes[ins[j]]['level'] <= nd['level']
            else:
                assert ins[j] == -1
    roots = nodes[nodes['out_row'] >= 0]
    assert sorted(roots['out_row'].tolist()) == list(range(d.N))
    assert 1 <= packed.num_levels <= 8
    assert packed.num_launches <= 3 + 3 * packed.num_levels


def test_pack_expr_list_roundtrip(asm):
    """dict walk (build_feed_dict path) and token path produce the same trees."""
    rng = np.random.default_rng(0)
    from n2nmn_amd.synth import random_valid_layouts
    from n2nmn_amd.spec import Dims
    toks = random_valid_layouts(Dims(), asm.P, asm.W, asm.b, seed=3, n=40, T=12)
    exprs, validity = asm.assemble(toks)
    assert validity.all()
    plain = [dict(e) for e in exprs]                      # drops the cached .packed
    repacked = asm.pack_expr_list(plain)
    a, b = exprs.packed.nodes(), repacked.nodes()

    def canon(nodes):
        out = {}
        def sig(i):
            nd = nodes[i]
            return (int(nd['op']), int(nd['time_idx']), int(nd['batch_idx']),
                    sig(nd['in0']) if nd['in0'] >= 0 else None,
                    sig(nd['in1']) if nd['in1'] >= 0 else None)
        for i, nd in enumerate(nodes):
            if nd['out_row'] >= 0:
                out[int(nd['out_row'])] = sig(i)
        return out
    assert canon(a) == canon(b) and len(a) == len(b)


def test_from_nodes_rejects_bad_graphs():
    F, C = OP_CODE['_Find'], OP_CODE['_Count']
    with pytest.raises(ValueError):
        PackedLayouts.from_nodes([(C, 0, 0, 0, -1, 0)], 1)          # input refers to itself
    with pytest.raises(ValueError):
        PackedLayouts.from_nodes([(F, 0, 0, -1, -1, -1), (C, 1, 0, 0, -1, 5)], 1)   # bad out_row
    with pytest.raises(KeyError):
        PackedLayouts.from_nodes([(77, 0, 0, -1, -1, -1)], 1)
    p = PackedLayouts.from_nodes([(F, 0, 0, -1, -1, -1), (C, 1, 0, 0, -1, 0)], 1)
    assert p.num_nodes == 2 and p.num_levels == 2


def test_automaton_walks_always_assemble(asm):
    """Reference invariant (train_clevr_gt_layout.py:186): layouts decoded under the validity
    automaton are always valid."""
    from n2nmn_amd.synth import random_valid_layouts
    from n2nmn_amd.spec import Dims
    for T in (6, 10, 20):
        toks = random_valid_layouts(Dims(), asm.P, asm.W, asm.b, seed=T, n=200, T=T)
        _, validity = asm.assemble_packed(toks)
        assert validity.all()

"""Engine.layout_nesting (host side of n2nmn_walk_set_nesting_bound): the deepest nesting of _Transform /
_FindSameProperty nodes of a batch of HOST layouts, against a one-column-at-a-time restatement of the walker's
plan (kernels_walk.hip plan_layout; stack machine of models_clevr/nmn3_assembler.py:153-222)."""
import numpy as np

from n2nmn_amd import synth
from n2nmn_amd.engine import Engine
from n2nmn_amd.nmn3_assembler import Assembler
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES, MODULE_INPUT_NUM


class _Host:                      # layout_nesting needs the assembler only (no context, no GPU)
    def __init__(self, asm):
        self.assembler = asm


def _one_column(asm, col):
    stack, deepest = [], 0
    for tok in col:
        if tok == asm.EOS_idx:
            break
        name = asm.module_names[tok]
        k = MODULE_INPUT_NUM[name]
        if len(stack) < k:
            break
        ins = [stack.pop() for _ in range(k)]
        hd = max(ins, default=0) + (name in ('_Transform', '_FindSameProperty'))
        stack.append(hd)
        deepest = max(deepest, hd)
    return deepest


def test_layout_nesting_equals_the_column_by_column_stack_machine():
    d = Dims()
    asm = Assembler(list(CLEVR_MODULE_NAMES))
    h = _Host(asm)
    rng = np.random.default_rng(5)
    cases = [synth.template_layout_batch(d, offset=k) for k in range(3)]
    cases += [synth.clevr_like_layout_batch(d, seed=k) for k in range(3)]
    cases += [synth.random_valid_layouts(d, asm.P, asm.W, asm.b, seed=9 + k, max_len=14) for k in range(3)]
    cases.append(rng.integers(0, asm.num_vocab_nmn, size=(d.T_decoder, d.N)).astype(np.int32))   # mostly invalid
    for g in cases:
        want = max(_one_column(asm, g[:, i]) for i in range(g.shape[1]))
        assert Engine.layout_nesting(h, g) == want
    assert Engine.layout_nesting(h, cases[0]) == 1            # the template mix never nests
    deep = np.array([asm.module_list2tokens(['_Find'] + ['_FindSameProperty', '_Transform'] * 3 + ['_Count'],
                                            d.T_decoder)], np.int32).T
    assert Engine.layout_nesting(h, deep) == 6
    flat = np.array([asm.module_list2tokens(['_Find', '_Find', '_And', '_Exist'], d.T_decoder)], np.int32).T
    assert Engine.layout_nesting(h, flat) == 0


def test_behind_an_answer_operator_the_automaton_allows_only_eos():
    """What eos_retire's sequential form rests on (dec_compact_kernel): in every state the validity automaton
    of the layout decoder (models_clevr/nmn3_netgen_att.py:8-15 with the matrices of nmn3_assembler.py:94-117)
    can reach behind an answer operator -- or behind <eos> -- the only valid token is <eos>.  Random valid walks
    over the matrices that are pinned to the reference's own (tests/golden/assembler_golden.json)."""
    from oracle import n2nmn_oracle as O
    from n2nmn_amd.spec import MODULE_OUTPUT_TYPE
    d = Dims()
    names = list(CLEVR_MODULE_NAMES)
    eos = names.index('<eos>')
    P, Wv, bv = O.build_validity_mats(names)
    rng = np.random.default_rng(11)
    checked = 0
    for _ in range(1500):
        X = np.array([0, 0, d.T_decoder])
        done = False
        for t in range(d.T_decoder):
            valid = np.nonzero(np.all((X[:, None, None] * Wv).sum(0) - bv >= 0, axis=1))[0]
            assert len(valid) >= 1
            if done:
                checked += 1
                assert list(valid) == [eos], (t, X, valid)
            tok = int(rng.choice(valid))
            done = done or tok == eos or MODULE_OUTPUT_TYPE[names[tok]] == 'ans'
            X = X + P[tok]
    assert checked > 10000

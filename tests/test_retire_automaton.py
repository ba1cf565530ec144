"""n2nmn_automaton_forces_eos (host only): the proof the library runs when validity tables are installed, before it lets
the sequential decoder retire finished rows (N2NMN_S2S_EOS_RETIRE).  The reference's automaton
(models_clevr/nmn3_assembler.py:50-135; tables from tests/golden/assembler_golden.json, produced by the reference's own
code) has the property for the CLEVR and the VQA vocabularies; all-zero tables (models_shapes: every token always valid)
and a relaxed automaton do not."""
import json
import os

import numpy as np

from n2nmn_amd import _lib
from n2nmn_amd.nmn3_assembler import Assembler
from n2nmn_amd.spec import CLEVR_MODULE_NAMES

HERE = os.path.dirname(os.path.abspath(__file__))


def _forces(P, W, b, tok_op, T):
    P, W, b, t = (np.ascontiguousarray(x, np.int32) for x in (P, W, b, tok_op))
    return _lib.check(_lib.lib().n2nmn_automaton_forces_eos(P.ctypes.data, W.ctypes.data, b.ctypes.data, t.ctypes.data,
                                                            int(t.shape[0]), int(T)))


def test_reference_automaton_forces_eos_behind_eos_and_answer_operators():
    asm = Assembler(list(CLEVR_MODULE_NAMES))
    g = json.load(open(os.path.join(HERE, 'golden', 'assembler_golden.json')))
    ref = g['clevr'] if 'clevr' in g else None
    if ref is not None and 'P' in ref:          # the drop-in's tables ARE the reference's (test_host_assembler.py)
        assert np.array_equal(np.asarray(ref['P']), asm.P)
    for T in (1, 2, 10, 20):
        assert _forces(asm.P, asm.W, asm.b, asm._token_op, T) == 1, T


def test_tables_without_the_property_are_rejected():
    asm = Assembler(list(CLEVR_MODULE_NAMES))
    V = len(CLEVR_MODULE_NAMES)
    zeros = (np.zeros((V, 3), np.int32), np.zeros((3, V, 4), np.int32), np.zeros((V, 4), np.int32))
    assert _forces(*zeros, asm._token_op, 20) == 0                      # models_shapes: every token always valid
    # the reference's automaton with the "no token behind an answer" constraint of one operator removed: _Find stays
    # valid while time remains, whatever has been emitted
    W, b = asm.W.copy(), asm.b.copy()
    s = list(CLEVR_MODULE_NAMES).index('_Find')
    W[:, s, :] = 0
    b[s, :] = 0
    assert _forces(asm.P, W, b, asm._token_op, 20) == 0
    # no <eos> token in the table at all
    assert _forces(asm.P, asm.W, asm.b, np.abs(asm._token_op), 20) == 0

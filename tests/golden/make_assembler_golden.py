#!/usr/bin/env python3
"""Generate tests/golden/assembler_golden.json by RUNNING THE REFERENCE'S OWN numpy code.

Only works in the build container (needs /root/reference); the JSON it writes is committed and is
what travels to the GPU box.  The reference modules import tensorflow / tensorflow_fold at the
top, so both are stubbed in sys.modules (only the pure-numpy Assembler is exercised), and
`np.bool` (removed in numpy >= 1.24, used at models_clevr/nmn3_assembler.py:221) is aliased.
No bytecode is written into the read-only reference tree.

Usage:  python tests/golden/make_assembler_golden.py
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'assembler_golden.json')


def _import_reference():
    for name in ('tensorflow', 'tensorflow_fold'):
        m = types.ModuleType(name)
        m.convert_to_tensor = lambda *a, **k: None
        sys.modules[name] = m
    if not hasattr(np, 'bool'):
        np.bool = bool
    sys.path.insert(0, REF)
    from models_clevr import nmn3_assembler as clevr_asm
    from models_vqa import nmn3_assembler as vqa_asm
    return clevr_asm, vqa_asm


def _strip(expr):
    """expr dict -> JSON-able nested dict (ints instead of numpy ints)."""
    out = {}
    for k, v in expr.items():
        if isinstance(v, dict):
            out[k] = _strip(v)
        elif isinstance(v, (np.integer,)):
            out[k] = int(v)
        else:
            out[k] = v
    return out


def _automaton_walks(asm, rng, count, T):
    """random walks under the reference's P/W/b (numpy emulation of nmn3_netgen_att.py:8-15)."""
    seqs = []
    for _ in range(count):
        X = np.array([0, 0, T], np.int64)
        s = []
        for _t in range(T):
            ok = np.all(np.tensordot(X, asm.W, axes=1) - asm.b >= 0, axis=1)
            tok = int(rng.choice(np.nonzero(ok)[0]))
            s.append(tok)
            X = X + asm.P[tok]
        seqs.append(s)
    return seqs


def main():
    clevr_asm, vqa_asm = _import_reference()
    rng = np.random.default_rng(20260925)
    gold = {}

    # ---- CLEVR -------------------------------------------------------------------------
    asm = clevr_asm.Assembler(os.path.join(REF, 'exp_clevr/data/vocabulary_layout.txt'))
    gold['clevr'] = {
        'module_names': asm.module_names, 'EOS_idx': int(asm.EOS_idx),
        'P': asm.P.tolist(), 'W': asm.W.tolist(), 'b': asm.b.tolist()}
    V = asm.num_vocab_nmn
    cases = []

    def add(tokens_TN, tag):
        tokens_TN = np.asarray(tokens_TN, np.int32)
        exprs, validity = asm.assemble(tokens_TN)
        cases.append({'tag': tag, 'tokens': tokens_TN.tolist(),
                      'validity': [bool(v) for v in validity],
                      'exprs': [_strip(e) for e in exprs]})

    # KATs of SURVEY Appendix B + the 10 bench templates
    kats = [['_Find', '_Transform', '_Filter', '_Describe'], ['_Find', '_Find', '_EqualNum'],
            ['_Find', '_And', '_Count'], ['_Scene', '_Describe', '_Find'],
            ['_Find'], ['_Find', '_Count', '_Count'], ['_Count'], [],
            ['_Find', '_Find', '_Find', '_And', '_Or', '_Exist'],
            ['_Scene', '_Find', '_SameProperty'], ['_Find', '_Describe', '_Transform']]
    templates = [
        ['_Find', '_Count'], ['_Find', '_Exist'], ['_Find', '_Describe'],
        ['_Find', '_Transform', '_Filter', '_Describe'], ['_Find', '_FindSameProperty', '_Count'],
        ['_Find', '_Find', '_EqualNum'], ['_Find', '_Find', '_MoreNum'],
        ['_Find', '_Find', '_SameProperty'],
        ['_Find', '_Transform', '_Find', '_Transform', '_And', '_Filter', '_Count'],
        ['_Find', '_Find', '_Or', '_Exist']]
    for T in (10, 20):
        cols = [asm.module_list2tokens(m, T) for m in kats + templates]
        add(np.array(cols, np.int32).T, 'kats_templates_T%d' % T)
    # a layout with no <eos> at all
    add(np.array([[1, 4, 4, 4, 4, 4]], np.int32).T, 'no_eos')
    # uniformly random token soup (mostly invalid; every error branch)
    for T in (6, 10, 20):
        add(rng.integers(0, V, size=(T, 96)), 'random_T%d' % T)
    # short random prefixes followed by <eos> padding (hits the stack-size / type branches)
    for T in (10, 20):
        toks = np.full((T, 128), asm.EOS_idx, np.int32)
        for n in range(128):
            ln = int(rng.integers(0, 6))
            toks[:ln, n] = rng.integers(0, V - 1, size=ln)
        add(toks, 'short_prefix_T%d' % T)
    # automaton-constrained walks: the reference's invariant is that all of them are valid
    for T in (6, 10, 20):
        walks = np.array(_automaton_walks(asm, rng, 160, T), np.int32).T
        add(walks, 'automaton_T%d' % T)
        assert all(cases[-1]['validity']), 'reference automaton produced an invalid layout'
    gold['clevr']['cases'] = cases
    # module_list2tokens error contract (nmn3_assembler.py:140-141)
    try:
        asm.module_list2tokens(['_Find'] * 10, 10)
        raise AssertionError('expected ValueError')
    except ValueError as e:
        gold['clevr']['list2tokens_error'] = str(e)

    # ---- VQA (5-token vocabulary; fixtures exp_vqa/data/*gt_layout*.npy) ---------------
    vasm = vqa_asm.Assembler(os.path.join(REF, 'exp_vqa/data/vocabulary_layout.txt'))
    gold['vqa'] = {'module_names': vasm.module_names, 'EOS_idx': int(vasm.EOS_idx),
                   'P': vasm.P.tolist(), 'W': vasm.W.tolist(), 'b': vasm.b.tolist()}
    uniq = {}
    for fn in ('gt_layout_val2014_new_parse.npy', 'v2_gt_layout_val2014_new_parse.npy'):
        d = np.load(os.path.join(REF, 'exp_vqa/data', fn), allow_pickle=True,
                    encoding='latin1').item()
        for lay in d.values():
            uniq.setdefault(tuple(lay), 0)
            uniq[tuple(lay)] += 1
    layouts = sorted(uniq)
    toks = np.array([vasm.module_list2tokens(list(l), 20) for l in layouts], np.int32).T
    exprs, validity = vasm.assemble(toks)
    gold['vqa']['gt_layouts'] = [list(l) for l in layouts]
    gold['vqa']['gt_layout_counts'] = [int(uniq[l]) for l in layouts]
    gold['vqa']['gt_tokens_T20'] = toks.tolist()
    gold['vqa']['gt_validity'] = [bool(v) for v in validity]
    gold['vqa']['gt_exprs'] = [_strip(e) for e in exprs]

    with open(OUT, 'w') as f:
        json.dump(gold, f, separators=(',', ':'))
    print('wrote', OUT, os.path.getsize(OUT), 'bytes;', len(cases), 'clevr case groups,',
          len(layouts), 'unique vqa layouts')


if __name__ == '__main__':
    main()

"""The layout walker (n2nmn_execute_tokens / n2nmn_walk_layouts: on-device Assembler.assemble + Fold
batching + every Modules.* operator in one kernel) against
  * the reference's own assembler goldens: the DEVICE's validity bit of every one of the ~1070 CLEVR
    reference-generated token sequences (bit-exact, models_clevr/nmn3_assembler.py:153-222),
  * the fp64 oracle and the reference-code fixture (1e-4),
  * the level-scheduler path (n2nmn_assemble + n2nmn_execute_program), which stays the compatibility
    and training path.
"""
import json
import os
import sys

import numpy as np
import pytest

from oracle import n2nmn_oracle as O
from n2nmn_amd import synth
from n2nmn_amd.spec import CLEVR_MODULE_NAMES
from util import assert_close, t2n

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import float_cases as FC  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-4
NAMES = list(CLEVR_MODULE_NAMES)


def test_walker_is_the_default_path(clevr_engine):
    eng, d, asm, w = clevr_engine
    assert eng.walk_supported()


def test_device_validity_equals_reference_assembler_on_every_golden_case(clevr_engine, golden):
    """The five validity checks run on the GPU: every reference-generated sequence (KATs, templates,
    random soup, short prefixes, automaton walks) gets the reference's validity bit, and invalid
    layouts give exactly zero logits."""
    import torch
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=3)
    checked = invalid = 0
    for case in golden['clevr']['cases']:
        toks = np.array(case['tokens'], np.int32)          # [T, n]
        T, n = toks.shape
        if T > d.T_decoder:
            continue
        for c0 in range(0, n, d.N):
            tk = toks[:, c0:c0 + d.N]
            nb = tk.shape[1]
            wv = torch.zeros((T, nb, d.embed_dim_txt), device=eng.device)
            scores, validity = eng.execute_tokens(tk, batch['image_feat_batch'][:nb], wv)
            want = np.array(case['validity'][c0:c0 + nb], bool)
            got = t2n(validity).astype(bool)
            assert np.array_equal(got, want), case['tag']
            sc = t2n(scores)
            assert np.all(sc[~want] == 0.0), case['tag']
            assert np.isfinite(sc).all()
            checked += nb
            invalid += int((~want).sum())
    assert checked > 1000 and invalid > 500


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_walker_matches_oracle_and_level_path_on_random_trees(clevr_engine, seed):
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=50 + seed, min_len=1)
    toks = synth.random_valid_layouts(d, asm.P, asm.W, asm.b, seed=10 + seed,
                                      max_len=[4, 9, None][seed - 1])
    ref = O.forward(w, NAMES, batch, d.T_decoder, d.num_choices, np.float64, forced_tokens=toks)
    s2s = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], forced_tokens=toks)
    scores, validity = eng.execute_tokens(s2s['predicted_tokens'], batch['image_feat_batch'],
                                          s2s['word_vecs'])
    assert t2n(validity).all()
    got = t2n(scores).copy()
    assert_close('walker vs oracle', got, ref['scores'], TOL)
    packed, v2 = asm.assemble_packed(toks)
    lvl = t2n(eng.execute(packed, batch['image_feat_batch'], s2s['word_vecs']))
    assert_close('walker vs level scheduler', got, lvl, 2e-5)


def test_walker_mixed_valid_and_invalid_rows(clevr_engine):
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=61)
    toks = synth.template_layout_batch(d)
    eos = asm.EOS_idx
    toks[:, 3] = eos                                     # empty layout: stack size 0
    toks[:, 7] = asm.name2idx_dict['_Find']              # no <eos>
    toks[:3, 11] = [asm.name2idx_dict['_Find'], asm.name2idx_dict['_And'], asm.name2idx_dict['_Count']]
    toks[3:, 11] = eos                                   # not enough input for _And
    toks[:3, 12] = [asm.name2idx_dict['_Find'], asm.name2idx_dict['_Count'], asm.name2idx_dict['_Exist']]
    toks[3:, 12] = eos                                   # input incompatible (ans fed to Exist)
    toks[:2, 13] = [asm.name2idx_dict['_Find'], asm.name2idx_dict['_Transform']]
    toks[2:, 13] = eos                                   # result type must be ans
    ref = O.forward(w, NAMES, batch, d.T_decoder, d.num_choices, np.float64, forced_tokens=toks)
    s2s = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], forced_tokens=toks)
    scores, validity = eng.execute_tokens(s2s['predicted_tokens'], batch['image_feat_batch'],
                                          s2s['word_vecs'])
    assert np.array_equal(t2n(validity).astype(bool), ref['validity'])
    assert not ref['validity'][[3, 7, 11, 12, 13]].any()
    assert_close('scores', t2n(scores), ref['scores'], TOL)


def test_walker_small_ragged_batch(clevr_engine):
    eng, d, asm, w = clevr_engine
    from n2nmn_amd.spec import Dims
    small = Dims(N=5, T_encoder=9, T_decoder=12)
    batch = synth.make_inputs(small, seed=5, min_len=1)
    toks = synth.template_layout_batch(small, offset=3)
    ref = O.forward(w, NAMES, batch, small.T_decoder, d.num_choices, np.float64, forced_tokens=toks)
    scores, tokens, validity = eng.forward(batch, T_dec=small.T_decoder, use_gt_layout=True,
                                           gt_layout=toks)
    ref_gt = O.forward(w, NAMES, batch, small.T_decoder, d.num_choices, np.float64,
                       use_gt_layout=True, gt_layout=toks)
    assert validity.all() and np.array_equal(tokens, toks)
    assert_close('scores', t2n(scores), ref_gt['scores'], TOL)
    del ref


def test_super_bucket_of_forked_batches_equals_single_batches(clevr_engine):
    """K = 3 in-flight batches (forked contexts: own workspace, shared weights) in ONE walker
    launch give the logits of three single-batch launches."""
    import torch
    eng, d, asm, w = clevr_engine
    engines = [eng, eng.fork(), eng.fork()]
    jobs, singles = [], []
    for k, e in enumerate(engines):
        batch = synth.make_inputs(d, seed=70 + k)
        toks = synth.random_valid_layouts(d, asm.P, asm.W, asm.b, seed=20 + k, max_len=6)
        s2s = e.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], forced_tokens=toks,
                        reuse_buffers=False)
        feat = e._dev(batch['image_feat_batch'], torch.float32)
        sc1, _ = e.execute_tokens(s2s['predicted_tokens'], feat, s2s['word_vecs'],
                                  reuse_buffers=False)
        singles.append(t2n(sc1).copy())
        scores = torch.full((d.N, d.num_choices), float('nan'), device=eng.device)
        valid = torch.zeros((d.N,), dtype=torch.int32, device=eng.device)
        jobs.append((e, s2s['predicted_tokens'], feat, s2s['word_vecs'], scores, valid))
    torch.cuda.synchronize()
    eng.walk(jobs, d.N, d.T_decoder)           # 192 questions: pooling answers run deferred
    for k in range(3):
        assert_close('batch %d' % k, t2n(jobs[k][4]), singles[k], 2e-5)
        assert t2n(jobs[k][5]).all()
    try:
        eng.set_defer_pool(0)                  # same path as the single launches: bit-identical
        eng.walk(jobs, d.N, d.T_decoder)
    finally:
        eng.set_defer_pool(-1)
    for k in range(3):
        assert np.array_equal(t2n(jobs[k][4]), singles[k])


def test_walker_on_the_reference_code_fixture(clevr_engine):
    """the greedy case of tests/golden/float_golden.npz (numbers produced by the reference's own
    code): decoder tokens -> walker, nothing fetched in between."""
    eng, d0, asm, w = clevr_engine
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                             'float_golden.npz'))
    d, batch = FC.clevr_inputs('greedy')
    scores, tokens, validity = eng.forward(batch, T_dec=d.T_decoder, fetch=False)
    assert np.array_equal(t2n(tokens), z['greedy/predicted_tokens'])
    assert np.array_equal(t2n(validity).astype(bool), z['greedy/validity'])
    assert_close('scores', t2n(scores), z['greedy/scores'], TOL)


def test_attention_table_text_maps_equal_word_vec_text_maps(clevr_engine):
    """fc_text(sum_tau att * emb[word]) == b + sum_tau att * (emb . W_txt)[word]: the walker fed with
    the decoder's attention maps (no word_vecs / text-map launches) against the walker fed with
    word_vecs, and against the oracle."""
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=77, min_len=1)
    toks = synth.random_valid_layouts(d, asm.P, asm.W, asm.b, seed=31, max_len=8)
    ref = O.forward(w, NAMES, batch, d.T_decoder, d.num_choices, np.float64, forced_tokens=toks)
    s2s = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], forced_tokens=toks,
                      reuse_buffers=False)
    a, _ = eng.execute_tokens(s2s['predicted_tokens'], batch['image_feat_batch'], s2s['word_vecs'],
                              reuse_buffers=False)
    b, _ = eng.execute_tokens(s2s['predicted_tokens'], batch['image_feat_batch'], None,
                              reuse_buffers=False,
                              atts=(s2s['atts'], s2s['_input_seq'], s2s['_seq_length']))
    assert_close('table vs word_vecs', t2n(b), t2n(a), 2e-5)
    assert_close('table vs oracle', t2n(b), ref['scores'], TOL)


@pytest.mark.parametrize('seed', [1, 2])
def test_deferred_pooling_equals_inline_pooling(clevr_engine, seed):
    """throughput mode: root Describe / SameProperty leave the walker for walk_pool_kernel +
    walk_heads_kernel; same logits as the in-walker path and the oracle."""
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=80 + seed, min_len=1)
    toks = synth.random_valid_layouts(d, asm.P, asm.W, asm.b, seed=40 + seed, max_len=5) \
        if seed == 1 else synth.template_layout_batch(d, offset=2)
    ref = O.forward(w, NAMES, batch, d.T_decoder, d.num_choices, np.float64, forced_tokens=toks)
    s2s = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], forced_tokens=toks,
                      reuse_buffers=False)
    names = asm.module_names
    roots = [names[toks[(toks[:, n] != asm.EOS_idx).sum() - 1, n]] for n in range(d.N)]
    assert sum(r in ('_Describe', '_SameProperty') for r in roots) >= 8
    out = {}
    try:
        for mode in (0, 1):
            eng.set_defer_pool(mode)
            sc, val = eng.execute_tokens(s2s['predicted_tokens'], batch['image_feat_batch'],
                                         s2s['word_vecs'], reuse_buffers=False)
            out[mode] = t2n(sc).copy()
            assert t2n(val).all()
    finally:
        eng.set_defer_pool(-1)
    assert_close('deferred vs inline', out[1], out[0], 2e-5)
    assert_close('deferred vs oracle', out[1], ref['scores'], TOL)


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_chip_wide_front_end_equals_in_walker_front_end(clevr_engine, seed):
    """passes of many questions: walk_tmap_kernel (text maps from the attention tables) and
    walk_find_kernel (Find / Filter epilogues, 4 workgroups per question) run ahead of the walker
    (n2nmn_walk_set_front_end); same logits as the all-in-one walker and the oracle, on random deep
    layouts (up to 4+ Find-type nodes), the template mix, and a batch with invalid layouts."""
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=60 + seed, min_len=1)
    if seed == 2:
        toks = synth.template_layout_batch(d, offset=3)
    else:
        toks = synth.random_valid_layouts(d, asm.P, asm.W, asm.b, seed=70 + seed, max_len=9)
    if seed == 3:                                   # some invalid columns: zero logits, validity 0
        toks = toks.copy()
        toks[:, ::7] = asm.name2idx_dict['_Find']
    ref = O.forward(w, NAMES, batch, d.T_decoder, d.num_choices, np.float64, forced_tokens=toks)
    s2s = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], forced_tokens=toks,
                      reuse_buffers=False, word_vecs=False)
    out, val = {}, {}
    try:
        for mode in (0, 1):
            eng.set_front_end(mode)
            sc, v = eng.execute_tokens(s2s['predicted_tokens'], batch['image_feat_batch'], None,
                                       reuse_buffers=False,
                                       atts=(s2s['atts'], s2s['_input_seq'], s2s['_seq_length']))
            out[mode], val[mode] = t2n(sc).copy(), t2n(v).copy()
    finally:
        eng.set_front_end(-1)
    assert np.array_equal(val[0], val[1]) and np.array_equal(val[1].astype(bool), ref['validity'])
    assert_close('chip-wide vs in-walker front end', out[1], out[0], 2e-6)
    assert_close('chip-wide front end vs oracle', out[1], ref['scores'], TOL)


def test_walker_node_counters_equal_the_layouts_executed(clevr_engine):
    """The profiler's algorithmic bytes / flops of the walker come from node counters the profiled
    launches accumulate on the device (n2nmn_debug_walk_stats).  They must equal what the layouts
    contain -- round 2 shipped with the pooled-input counter commented out (always 0)."""
    import ctypes as C
    from n2nmn_amd import _lib
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=5)
    gt = synth.random_valid_layouts(d, asm.P, asm.W, asm.b, seed=4, max_len=7)
    eng.profile_begin()
    _, tokens, validity = eng.forward(batch, use_gt_layout=True, gt_layout=gt)
    eng.profile_end()
    assert validity.all()
    st = (C.c_uint64 * 10)()
    _lib.check(eng._lib.n2nmn_debug_walk_stats(eng._ctx, st))
    idx = {n: i for i, n in enumerate(NAMES)}
    t = np.asarray(tokens)
    cnt = lambda *names: sum(int((t == idx[k]).sum()) for k in names)
    pool = cnt('_FindSameProperty', '_SameProperty', '_Describe')
    text = cnt('_Find', '_Filter', '_FindSameProperty', '_SameProperty', '_Describe', '_Transform')
    find_passes = sum((int(np.isin(t[:, i], [idx['_Find'], idx['_Filter']]).sum()) + 3) // 4
                      for i in range(t.shape[1]))
    assert st[5] == t.shape[1]                                     # valid questions
    assert st[2] == pool                                           # pooling nodes
    assert st[1] == pool + cnt('_SameProperty')                    # pooled attention inputs
    assert st[3] == text and st[4] == cnt('_Transform')
    assert st[0] + st[8] == cnt('_FindSameProperty') + find_passes  # conv_image map passes


# staged walker vs the one-workgroup walker: the same operators with other summation orders (the answer
# heads reduce by waves, FindSameProperty pools / applies fc_att in 8 channel parts)
STAGED_TOL = 1e-5


@pytest.mark.parametrize('seed', [1, 2, 3, 4])
def test_staged_walker_equals_the_one_workgroup_walker(clevr_engine, seed):
    """passes of many questions (n2nmn_walk_set_staged): the plan step of walk_tmap_kernel decodes every
    layout on the device, walk_heavy_kernel runs the Transform / FindSameProperty nodes over light input
    subtrees as chip-wide jobs, walk_light_kernel finishes the tree, and questions with NESTED
    Transform / FindSameProperty nodes stay with walk_kernel.  Same logits and validity as the
    one-workgroup walker (same operators; FindSameProperty pools and applies fc_att in channel parts, so its
    sums run in another order: 1e-5) and the oracle, on the template mix (every staged
    operator, no nesting), random layouts (shallow: mostly staged; deep: mostly the fall-back list), and
    a batch with invalid columns."""
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=160 + seed, min_len=1)
    if seed == 2:
        toks = synth.template_layout_batch(d, offset=5)
    else:
        toks = synth.random_valid_layouts(d, asm.P, asm.W, asm.b, seed=170 + seed,
                                          max_len=5 if seed == 1 else 12)
    if seed == 3:                                   # some invalid columns: zero logits, validity 0
        toks = toks.copy()
        toks[:, ::5] = asm.name2idx_dict['_Find']
    names = asm.module_names
    heavy = np.isin(toks, [asm.name2idx_dict['_Transform'], asm.name2idx_dict['_FindSameProperty']])
    print('layouts with a Transform / FindSameProperty node: %d of %d (with two or more: %d)' %
          (int(heavy.any(0).sum()), d.N, int((heavy.sum(0) >= 2).sum())))
    ref = O.forward(w, NAMES, batch, d.T_decoder, d.num_choices, np.float64, forced_tokens=toks)
    s2s = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], forced_tokens=toks,
                      reuse_buffers=False, word_vecs=False)
    out, val = {}, {}
    try:
        eng.set_front_end(1)
        eng.set_defer_pool(1)
        for mode in (0, 1, 1):                      # twice staged: the job lists must start empty again
            eng.set_staged(mode)
            sc, v = eng.execute_tokens(s2s['predicted_tokens'], batch['image_feat_batch'], None,
                                       reuse_buffers=False,
                                       atts=(s2s['atts'], s2s['_input_seq'], s2s['_seq_length']))
            if mode in out:
                # (the second staged pass may list nested layouts level by level where the first sent them
                # to the fall-back walker -- the host has seen the first pass's nesting depth by now: the
                # answer heads then reduce in walk_light_kernel's order instead of walk_kernel's)
                assert_close('second staged pass vs the first', t2n(sc), out[mode], STAGED_TOL)
            out[mode], val[mode] = t2n(sc).copy(), t2n(v).copy()
    finally:
        eng.set_front_end(-1)
        eng.set_defer_pool(-1)
        eng.set_staged(-1)
    assert np.array_equal(val[0], val[1]) and np.array_equal(val[1].astype(bool), ref['validity'])
    assert_close('staged vs one-workgroup walker', out[1], out[0], STAGED_TOL)
    assert_close('staged walker vs oracle', out[1], ref['scores'], TOL)


def test_staged_walker_node_counters(clevr_engine):
    """the profiler's node counters (n2nmn_debug_walk_stats) under the staged walker: every question is
    counted once, by walk_light_kernel or -- nested layouts -- by walk_kernel."""
    import ctypes as C
    from n2nmn_amd import _lib
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=9)
    gt = synth.random_valid_layouts(d, asm.P, asm.W, asm.b, seed=8, max_len=9)
    try:
        eng.set_front_end(1)
        eng.set_defer_pool(1)
        eng.profile_begin()
        _, tokens, validity = eng.forward(batch, use_gt_layout=True, gt_layout=gt)
        eng.profile_end()
    finally:
        eng.set_front_end(-1)
        eng.set_defer_pool(-1)
    assert validity.all()
    st = (C.c_uint64 * 10)()
    _lib.check(eng._lib.n2nmn_debug_walk_stats(eng._ctx, st))
    idx = {n: i for i, n in enumerate(NAMES)}
    t = np.asarray(tokens)
    cnt = lambda *names: sum(int((t == idx[k]).sum()) for k in names)
    pool = cnt('_FindSameProperty', '_SameProperty', '_Describe')
    find_passes = sum((int(np.isin(t[:, i], [idx['_Find'], idx['_Filter']]).sum()) + 3) // 4
                      for i in range(t.shape[1]))
    assert st[5] == t.shape[1]
    assert st[2] == pool and st[1] == pool + cnt('_SameProperty')
    assert st[4] == cnt('_Transform')
    assert st[0] + st[8] == cnt('_FindSameProperty') + find_passes


def _heavy_depth(tokens, asm):
    """deepest nesting of Transform / FindSameProperty per layout (host restatement of the plan step)"""
    heavy = {asm.name2idx_dict['_Transform'], asm.name2idx_dict['_FindSameProperty']}
    arity = {i: asm._input_num[n] for i, n in enumerate(asm.module_names) if n != '<eos>'}
    out = []
    for col in np.asarray(tokens).T:
        stack, deepest = [], 0
        for tok in col:
            if tok == asm.EOS_idx:
                break
            k = arity[int(tok)]
            ins = [stack.pop() for _ in range(k)]
            hd = max(ins, default=0) + (1 if int(tok) in heavy else 0)
            deepest = max(deepest, hd)
            stack.append(hd)
        out.append(deepest)
    return np.array(out)


def test_staged_walker_lists_nested_layouts_level_by_level(clevr_engine):
    """Nested Transform / FindSameProperty nodes (real CLEVR relate chains, the greedy decoder's layouts):
    the first pass that meets them serves them through the fall-back walker; the deepest nesting of a pass
    reaches the host through a host-mapped word, and the passes behind it launch one walk_heavy level per
    nesting depth (up to WALK_HLEVELS = 24: every layout of T_dec = 20 tokens), so nothing stays on the fall-back
    list.  Same logits
    whichever way a question went; the profiler counts the fall-back questions (walk stats [9])."""
    import ctypes as C
    from n2nmn_amd import _lib
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=31)
    toks = synth.random_valid_layouts(d, asm.P, asm.W, asm.b, seed=611, max_len=14)
    # (automaton walks rarely nest: relate chains of every depth 1 .. 6 and mixed trees are written in)
    chains = [['_Find'] + ['_Transform'] * k + ['_Describe'] for k in range(1, 7)] + \
             [['_Find'] + ['_FindSameProperty', '_Transform'] * k + ['_Count'] for k in range(1, 4)] + \
             [['_Find', '_Transform', '_Find', '_FindSameProperty', '_Transform', '_And', '_Filter', '_Exist'],
              ['_Find', '_Transform', '_Transform', '_Find', '_Transform', '_Or', '_FindSameProperty', '_Describe'],
              ['_Find', '_Find', '_Transform', '_FindSameProperty', '_SameProperty']]
    for i, lay in enumerate(chains * 3):
        toks[:, (5 * i + 1) % d.N] = asm.module_list2tokens(lay, d.T_decoder)
    depth = _heavy_depth(toks, asm)
    assert (depth >= 2).sum() >= 20 and (depth > 4).sum() >= 3 and (depth == 3).any() and (depth == 6).any()
    print('nesting depth histogram:', np.bincount(depth))
    s2s = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], forced_tokens=toks,
                      reuse_buffers=False, word_vecs=False)
    st = (C.c_uint64 * 10)()

    def one_pass(profile):
        if profile:
            eng.profile_begin()
        sc, v = eng.execute_tokens(s2s['predicted_tokens'], batch['image_feat_batch'], None, reuse_buffers=False,
                                   atts=(s2s['atts'], s2s['_input_seq'], s2s['_seq_length']))
        out = t2n(sc).copy()
        assert t2n(v).all()
        if profile:
            eng.profile_end()
            _lib.check(eng._lib.n2nmn_debug_walk_stats(eng._ctx, st))
            return out, int(st[9])
        return out, None
    try:
        eng.set_front_end(1)
        eng.set_defer_pool(1)
        eng.set_staged(0)
        ref, _ = one_pass(False)                        # the one-workgroup walker
        eng.set_staged(1)
        # (round 6) the DEFAULT launches every reachable nesting level: no layout reaches the fall-back list, and
        # the logits do not depend on what the context ran before -- bit for bit
        d0, fb_d0 = one_pass(True)
        d1, _ = one_pass(False)
        assert fb_d0 == 0 and np.array_equal(d0, d1)
        assert_close('default route vs the one-workgroup walker', d0, ref, STAGED_TOL)
        # the ADAPTIVE form (n2nmn_walk_set_levels(ctx, -1)): as many level launches as the last two passes needed
        eng.set_walk_levels(-1)
        # template passes first: the hint of earlier tests in this process must have decayed
        tpl = synth.template_layout_batch(d)
        s2t = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], forced_tokens=tpl,
                          reuse_buffers=False, word_vecs=False)
        for _ in range(3):
            eng.execute_tokens(s2t['predicted_tokens'], batch['image_feat_batch'], None, reuse_buffers=False,
                               atts=(s2t['atts'], s2t['_input_seq'], s2t['_seq_length']))
        first, fb_first = one_pass(True)
        assert fb_first == int((depth >= 2).sum())      # one level launched: every nested layout falls back
        one_pass(False)
        later, fb_later = one_pass(True)
        assert fb_later == 0                            # one walk_heavy launch per nesting level seen
    finally:
        eng.set_walk_levels(0)
        eng.set_front_end(-1)
        eng.set_defer_pool(-1)
        eng.set_staged(-1)
    assert_close('nested layouts on the fall-back list vs the one-workgroup walker', first, ref, STAGED_TOL)
    assert_close('nested layouts level by level vs the one-workgroup walker', later, ref, STAGED_TOL)
    full = O.forward(w, NAMES, batch, d.T_decoder, d.num_choices, np.float64, forced_tokens=toks)
    assert_close('level by level vs oracle', later, full['scores'], TOL)


def test_fixed_level_count_makes_nested_layouts_reproducible_bit_for_bit(clevr_engine):
    """n2nmn_walk_set_levels: in the adaptive form (levels = -1) a nested layout goes through the
    fall-back walker in a context's first nested pass and through the level launches later (other summation
    order, 1e-5).  With a fixed level count -- and with the default, which launches every reachable level
    (round 6) -- the route depends on the layout alone: a pass right after template passes and a pass after
    nested ones return the same bits, and the default returns the bits of levels = T_dec - 1."""
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=77)
    toks = synth.random_valid_layouts(d, asm.P, asm.W, asm.b, seed=612, max_len=14)
    for i, k in enumerate(range(1, 7)):
        toks[:, 3 * i] = asm.module_list2tokens(['_Find'] + ['_FindSameProperty', '_Transform'] * k + ['_Count']
                                                if k <= 3 else ['_Find'] + ['_Transform'] * k + ['_Describe'],
                                                d.T_decoder)
    tpl = synth.template_layout_batch(d)
    s2n = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], forced_tokens=toks,
                      reuse_buffers=False, word_vecs=False)
    s2t = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], forced_tokens=tpl,
                      reuse_buffers=False, word_vecs=False)

    def run(s2s):
        sc, v = eng.execute_tokens(s2s['predicted_tokens'], batch['image_feat_batch'], None, reuse_buffers=False,
                                   atts=(s2s['atts'], s2s['_input_seq'], s2s['_seq_length']))
        return t2n(sc).copy()
    try:
        eng.set_front_end(1)
        eng.set_defer_pool(1)
        eng.set_staged(1)
        eng.set_walk_levels(d.T_decoder - 1)
        for _ in range(3):
            run(s2t)
        a = run(s2n)                 # first nested pass of the "history"
        run(s2n)
        b = run(s2n)
        eng.set_walk_levels(0)       # the default: every reachable level
        for _ in range(2):
            run(s2t)
        c = run(s2n)
        eng.set_staged(0)
        ref = run(s2n)
    finally:
        eng.set_walk_levels(0)
        eng.set_front_end(-1)
        eng.set_defer_pool(-1)
        eng.set_staged(-1)
    assert np.array_equal(a, b), 'fixed level count: the logits must not depend on what ran before'
    assert np.array_equal(a, c), 'the default route is the route of levels = T_dec - 1'
    assert_close('level by level vs the one-workgroup walker', a, ref, STAGED_TOL)



def test_nesting_bound_from_host_layouts_launches_exact_levels_and_no_fallback(clevr_engine):
    """n2nmn_walk_set_nesting_bound: a caller that holds the layouts on the host tells the walker how deep
    they nest -> exactly that many level launches, no fall-back launch.  Same bits as a pass with a fixed
    level count that covers every layout; the promise covers ONE pass; a layout that breaks it comes back
    INVALID (validity 0, zero logits), the others are unaffected."""
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=31)
    toks = synth.random_valid_layouts(d, asm.P, asm.W, asm.b, seed=77, max_len=14)
    toks[:, 1] = asm.module_list2tokens(['_Find'] + ['_FindSameProperty', '_Transform'] * 2 + ['_Count'], d.T_decoder)
    toks[:, 2] = asm.module_list2tokens(['_Find', '_Transform', '_Exist'], d.T_decoder)
    nest = eng.layout_nesting(toks)
    assert nest >= 4
    s2s = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], forced_tokens=toks,
                      reuse_buffers=False, word_vecs=False)

    def run(bound=None):
        if bound is not None:
            eng.set_nesting_bound(bound)
        sc, v = eng.execute_tokens(s2s['predicted_tokens'], batch['image_feat_batch'], None, reuse_buffers=False,
                                   atts=(s2s['atts'], s2s['_input_seq'], s2s['_seq_length']))
        return t2n(sc).copy(), t2n(v).copy()
    try:
        eng.set_front_end(1)
        eng.set_defer_pool(1)
        eng.set_staged(1)
        eng.set_walk_levels(d.T_decoder - 1)
        ref = run()
        got = run(nest)
        after = run()                                # the promise is spent: the fixed level count again
        broken = run(1)
    finally:
        eng.set_walk_levels(0)
        eng.set_front_end(-1)
        eng.set_defer_pool(-1)
        eng.set_staged(-1)
    assert ref[1].all()
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert np.array_equal(after[0], ref[0])
    # bound 1: every layout nested deeper than one level is reported invalid, nothing else changes
    deep = np.array([eng.layout_nesting(toks[:, i:i + 1]) > 1 for i in range(d.N)])
    assert deep[1] and not deep[2] and 0 < deep.sum() < d.N
    assert np.array_equal(broken[1].astype(bool), ~deep)
    assert np.all(broken[0][deep] == 0) and np.array_equal(broken[0][~deep], ref[0][~deep])


@pytest.mark.parametrize('front', [0, 1])
def test_conv_image_inside_the_walker_call_equals_the_separate_launch(clevr_engine, front):
    """n2nmn_walk_set_conv_inline: the walker call computes the hoisted conv_image maps itself (behind the text
    maps, FindSameProperty's first, Find's last -- right in front of their reader).  Same GEMM kernels on the
    same operands as n2nmn_conv_image: the logits must be the same BITS, with the chip-wide front end and
    without it, and the request covers one call only."""
    eng, d, asm, w = clevr_engine
    batch = synth.make_inputs(d, seed=58, min_len=1)
    toks = synth.template_layout_batch(d, offset=4)
    s2s = eng.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], forced_tokens=toks,
                      reuse_buffers=False, word_vecs=False)
    import torch
    feat = torch.as_tensor(batch['image_feat_batch']).cuda()
    other = torch.as_tensor(synth.make_inputs(d, seed=59)['image_feat_batch']).cuda()

    def run(conv_done):
        sc, v = eng.execute_tokens(s2s['predicted_tokens'], feat, None, reuse_buffers=False, conv_done=conv_done,
                                   atts=(s2s['atts'], s2s['_input_seq'], s2s['_seq_length']))
        return t2n(sc).copy(), t2n(v).copy()
    try:
        eng.set_front_end(front)
        eng.set_defer_pool(front)
        eng.conv_image(feat, s2s['predicted_tokens'])
        sep = run(True)
        eng.conv_image(other, s2s['predicted_tokens'])       # stale maps of other images in the workspace
        inl = run(False)
        eng.conv_image(other, s2s['predicted_tokens'])
        stale = run(True)                                    # the request is spent: this reads the maps as they lie
    finally:
        eng.set_front_end(-1)
        eng.set_defer_pool(-1)
    assert np.array_equal(inl[0], sep[0]) and np.array_equal(inl[1], sep[1])
    assert not np.array_equal(stale[0], sep[0]), 'a request for the maps must cover ONE walker call'
    ref = O.forward(w, NAMES, batch, d.T_decoder, d.num_choices, np.float64, forced_tokens=toks)
    assert_close('vs oracle', inl[0], ref['scores'], TOL)

"""Data plane (n2nmn_amd/data_reader.py) against the REFERENCE'S OWN loaders: the digests in
tests/golden/data_reader_golden.json were produced by util/clevr_train/data_reader.py and
util/vqa_train/data_reader.py (BatchLoader* + _run_prefetch, imported from the checkout) on the
synthetic imdbs of tests/golden/data_reader_cases.py; bit-exact arrays, same keys, same epoch
order and random answers under the same numpy seed."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import data_reader_cases as DC  # noqa: E402

from n2nmn_amd import data_reader as R
from n2nmn_amd.nmn3_assembler import Assembler
from n2nmn_amd.spec import CLEVR_MODULE_NAMES
from n2nmn_amd.vqa import VQA_MODULE_NAMES, VQA_OP_CODE


def _assembler(variant, extra):
    if extra.get('use_count_module'):
        return DC.TokenTable(DC.COUNT_NAMES)
    if variant == 'clevr':
        return Assembler(list(CLEVR_MODULE_NAMES))
    return Assembler(list(VQA_MODULE_NAMES), op_code=VQA_OP_CODE)


def _batches(name, variant, rk, extra, root):
    imdb, params = DC.build(os.path.join(root, name), variant)
    params = dict(params, assembler=_assembler(variant, extra), **extra)
    np.random.seed(11)
    rd = R.DataReader(None, imdb=imdb, variant=variant, prefetch_num=2, **rk, **params)
    out = []
    for b in rd.batches():
        out.append(b)
        if len(out) == DC.NUM_BATCHES:
            break
    rd.close()
    return out


@pytest.fixture(scope='module')
def golden():
    with open(os.path.join(HERE, 'golden', 'data_reader_golden.json')) as f:
        return json.load(f)


@pytest.mark.parametrize('case', DC.CASES, ids=[c[0] for c in DC.CASES])
def test_batches_equal_the_reference_loaders(case, golden, tmp_path):
    name, variant, rk, extra = case
    got = DC.digest(_batches(name, variant, rk, extra, str(tmp_path)))
    want = golden[name]
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        assert sorted(g) == sorted(w), (name, i, sorted(g), sorted(w))
        for k in w:
            assert g[k] == w[k], (name, 'batch %d' % i, k, g[k], w[k])


def test_one_pass_ends_and_last_batch_is_short(tmp_path):
    bs = _batches('clevr_plain', 'clevr', dict(shuffle=False, one_pass=True), {}, str(tmp_path))
    assert [b['seq_length_batch'].shape[0] for b in bs] == [5, 5, 5, 5, 3]
    assert bs[0]['input_seq_batch'].shape == (10, 5) and bs[0]['input_seq_batch'].dtype == np.int32
    assert bs[0]['input_seq_batch'][0, 0] == 0            # 'zebra' -> <unk> = index 0


def test_prune_filter_modules():
    f = R.prune_filter_modules
    assert f(['_Find', '_Filter', '_Filter', '_Exist']) == ['_Find', '_Exist']
    assert f(['_Find', '_Transform', '_Filter', '_Describe']) == ['_Find', '_Transform', '_Filter', '_Describe']
    assert f(['_Scene', '_Count']) == ['_Scene', '_Count']


def test_vocab_without_unk_raises(tmp_path):
    p = tmp_path / 'v.txt'
    p.write_text('a\nb\n')
    v = R.VocabDict(str(p))
    assert v.word2idx('b') == 1 and v.tokenize_and_index('A  b') == [0, 1]
    assert R.tokenize('A, b!') == ['a', ',', 'b', '!']      # separators are tokens (text_processing.py:3-7)
    with pytest.raises(ValueError):
        v.word2idx('c')


def test_loader_errors_reach_the_consumer(tmp_path):
    imdb, params = DC.build(str(tmp_path / 'x'), 'clevr')
    imdb[7]['feature_path'] = str(tmp_path / 'missing.npy')
    rd = R.DataReader(None, imdb=imdb, shuffle=False, one_pass=True, load_gt_layout=False, **params)
    with pytest.raises(FileNotFoundError):
        list(rd.batches())


@pytest.mark.skipif(not os.path.isdir('/root/reference/util/clevr_train'),
                    reason='reference checkout not present (GPU box)')
def test_golden_is_what_the_reference_loaders_produce_today(golden, tmp_path):
    import subprocess
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    before = json.dumps(golden, sort_keys=True)
    subprocess.run([sys.executable, os.path.join(HERE, 'golden', 'make_data_reader_golden.py')],
                   check=True, env=env, stdout=subprocess.DEVNULL)
    with open(os.path.join(HERE, 'golden', 'data_reader_golden.json')) as f:
        assert json.dumps(json.load(f), sort_keys=True) == before


def test_batches_built_inside_staging_sets_equal_plain_batches(tmp_path):
    """use_staging: the prefetch thread builds each batch inside a preallocated set (what
    DeviceFeeder pins) -- same arrays as the plain reader, every batch's arrays are views of the set
    it names, and the reader stalls (instead of overwriting) while every set is out."""
    plain = _batches('clevr_plain', 'clevr', dict(shuffle=False, one_pass=True), {}, str(tmp_path))
    imdb, params = DC.build(os.path.join(str(tmp_path), 'clevr_plain_staged'), 'clevr')
    params = dict(params, assembler=_assembler('clevr', {}))
    rd = R.DataReader(None, imdb=imdb, variant='clevr', prefetch_num=2, shuffle=False, one_pass=True,
                      **params)
    time_cap = 64
    bl = rd.batch_loader

    def staging_set():
        return dict(input_seq_batch=np.full((bl.T_encoder, time_cap), -7, np.int32),
                    seq_length_batch=np.full((time_cap,), -7, np.int32),
                    image_feat_batch=np.full((time_cap, bl.feat_H, bl.feat_W, bl.feat_D), np.nan, np.float32))
    sets = [staging_set() for _ in range(4)]
    free = rd.use_staging(sets)
    staged = []
    it = rd.batches()
    for b in it:
        st = b.pop('_staging', None)
        staged.append({k: (np.array(v) if isinstance(v, np.ndarray) else v) for k, v in b.items()})
        if st is not None:
            assert any(st is s for s in sets)
            for k in ('input_seq_batch', 'seq_length_batch', 'image_feat_batch'):
                assert np.shares_memory(b[k], st[k]), k
            free.put(st)
    rd.close()
    assert len(staged) == len(plain)
    for a, b in zip(plain, staged):
        for k in ('input_seq_batch', 'seq_length_batch', 'image_feat_batch'):
            assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k
        assert a['image_path_list'] == b['image_path_list']

#!/usr/bin/env python3
"""tests/golden/data_reader_golden.json: digests of the batches the REFERENCE'S OWN loaders
(util/clevr_train/data_reader.py, util/vqa_train/data_reader.py: BatchLoaderClevr / BatchLoaderVqa and
_run_prefetch, imported from the checkout) produce on the synthetic imdbs of data_reader_cases.py.
Only runs where /root/reference exists.  Two shims for today's numpy, neither touching the logic:
the imdb is handed over as a list (np.load of a pickled object array needs allow_pickle=True, which
the reference's DataReader.__init__ does not pass) and `np.bool` is aliased to `bool`."""
import json
import os
import queue
import sys
import tempfile
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('N2NMN_REFERENCE', '/root/reference')
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, 'oracle', 'tf1_stub'))   # the assembler modules import tensorflow
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import data_reader_cases as DC  # noqa: E402


def reference_batches(variant, imdb, params, shuffle, one_pass, seed):
    if not hasattr(np, 'bool'):
        np.bool = bool
    if variant == 'clevr':
        from util.clevr_train import data_reader as R
        loader = R.BatchLoaderClevr(imdb, params)
    else:
        from util.vqa_train import data_reader as R
        loader = R.BatchLoaderVqa(imdb, params)
    q = queue.Queue(maxsize=2)
    np.random.seed(seed)
    th = threading.Thread(target=R._run_prefetch, args=(q, loader, imdb, shuffle, one_pass, params),
                          daemon=True)
    th.start()
    out = []
    while len(out) < DC.NUM_BATCHES:
        b = q.get()
        if b is None:
            break
        out.append(b)
    # an endless reader's thread keeps drawing from numpy's GLOBAL generator until it blocks on the
    # full queue: wait for that, or it would disturb the next case's seeded sequence
    import time
    deadline = time.time() + 5.0
    while th.is_alive() and time.time() < deadline:
        if q.full():
            time.sleep(0.3)         # the batch it was building when the queue filled up
            break
        time.sleep(0.01)
    return out          # the thread stays blocked on the full queue (daemon)


def assembler_for(variant, extra):
    """the reference's own Assembler objects (the loaders call only module_list2tokens)"""
    if extra.get('use_count_module'):
        return DC.TokenTable(DC.COUNT_NAMES)
    if variant == 'clevr':
        from models_clevr.nmn3_assembler import Assembler
        return Assembler(os.path.join(REF, 'exp_clevr/data/vocabulary_layout.txt'))
    from models_vqa.nmn3_assembler import Assembler
    return Assembler(os.path.join(REF, 'exp_vqa/data/vocabulary_layout.txt'))


def main():
    gold = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, variant, rk, extra in DC.CASES:
            imdb, params = DC.build(os.path.join(tmp, name), variant)
            params = dict(params, assembler=assembler_for(variant, extra), **extra)
            bs = reference_batches(variant, imdb, params, rk['shuffle'], rk['one_pass'], seed=11)
            gold[name] = DC.digest(bs)
            print(name, len(bs), 'batches')
    path = os.path.join(HERE, 'data_reader_golden.json')
    with open(path, 'w') as f:
        json.dump(gold, f, indent=0, sort_keys=True)
    print('wrote', path)


if __name__ == '__main__':
    main()

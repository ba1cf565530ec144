#!/usr/bin/env python3
"""Generates tests/golden/shapes_golden.json from the reference's own SHAPES files
(/root/reference/exp_shapes/...): the first 12 questions of `train.tiny` after the seed-3 shuffle of
exp_shapes/eval_shapes.py:86-95 -- vocabulary, token arrays, ground-truth layouts, labels, uint8
images and the image mean -- plus the outputs of oracle/n2nmn_oracle_shapes.py on them with seed-0
synthetic weights (a regression pin of the restatement; the reference's TF path cannot be run).
The reference does not travel to the GPU box, this fixture does."""
import base64
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
from oracle import n2nmn_oracle_shapes as S      # noqa: E402
from n2nmn_amd import synth                      # noqa: E402

REF = os.environ.get('N2NMN_REFERENCE', '/root/reference')
NQ = 12


def b64(a):
    return base64.b64encode(np.ascontiguousarray(a).tobytes()).decode()


def main():
    sp = S.load_split(REF, 'train.tiny')
    sel = slice(0, NQ)
    batch = dict(image_batch=(sp['images_u8'][sel].astype(np.float32) - sp['image_mean']).astype(np.float32),
                 text_seq_batch=sp['text_seq'][:, sel], seq_length_batch=sp['seq_length'][sel])
    w = synth.make_weights_from_shapes(S.variable_shapes(len(sp['vocab']), len(sp['layout_vocab'])), seed=0,
                                       dtype=np.float64)
    r_gt = S.forward(w, batch, use_gt_layout=True, gt_layout=sp['gt_layout'][:, sel])
    r_free = S.forward(w, batch)
    out = dict(
        vocab=sp['vocab'], layout_vocab=sp['layout_vocab'], num_questions=int(len(sp['labels'])),
        order_head=sp['order'][:NQ].tolist(), text_seq=sp['text_seq'][:, sel].tolist(),
        seq_length=sp['seq_length'][sel].tolist(), gt_layout=sp['gt_layout'][:, sel].tolist(),
        labels=sp['labels'][sel].tolist(), images_u8_b64=b64(sp['images_u8'][sel]),
        images_shape=list(sp['images_u8'][sel].shape), image_mean_b64=b64(sp['image_mean'].astype(np.float32)),
        image_mean_shape=list(sp['image_mean'].shape),
        scores_gt=r_gt['scores'].tolist(), validity_gt=r_gt['validity'].tolist(),
        tokens_free=r_free['dec']['predicted_tokens'].tolist(), validity_free=r_free['validity'].tolist(),
        scores_free=r_free['scores'].tolist(), feat_sum=float(r_gt['feat'].sum()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'shapes_golden.json')
    with open(path, 'w') as f:
        json.dump(out, f)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()

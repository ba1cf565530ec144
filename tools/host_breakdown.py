#!/usr/bin/env python3
"""Host-side time per phase of one hot-path step (enqueue cost vs GPU time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from n2nmn_amd import synth
from n2nmn_amd.engine import Engine
from n2nmn_amd.nmn3_assembler import Assembler
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES

d = Dims(); asm = Assembler(list(CLEVR_MODULE_NAMES)); eng = Engine(d, asm)
eng.load_weights(synth.make_weights(d, seed=0))
b = {k: torch.as_tensor(v).cuda() for k, v in synth.make_inputs(d, seed=0).items()}
gt = torch.as_tensor(synth.template_layout_batch(d)).cuda()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(20):
        eng.forward(b, use_gt_layout=True, gt_layout=gt)
    torch.cuda.synchronize()
    acc = [0.0] * 6
    n = 200
    for _ in range(n):
        t0 = time.perf_counter()
        s2s = eng.seq2seq(b['input_seq_batch'], b['seq_length_batch'], None, True, gt)
        t1 = time.perf_counter()
        tok = s2s['predicted_tokens'].cpu().numpy()
        t2 = time.perf_counter()
        packed, val = asm.assemble_packed(tok)
        t3 = time.perf_counter()
        sc = eng.execute(packed, b['image_feat_batch'], s2s['word_vecs'])
        t4 = time.perf_counter()
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        for i, v in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0)):
            acc[i] += v
    names = ['seq2seq enqueue', 'tokens .cpu() (wait phase 1)', 'assemble_packed', 'execute enqueue',
             'final sync (wait phase 2)', 'total']
    for nm, v in zip(names, acc):
        print('%-32s %8.1f us' % (nm, 1e6 * v / n))

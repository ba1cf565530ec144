#!/usr/bin/env python3
"""Per-layout cost of the layout walker: homogeneous batches (every question the same layout),
HIP-event time of the walker launch alone.  Usage: python tools/walk_microbench.py [N ...]"""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from n2nmn_amd import synth
from n2nmn_amd.engine import Engine
from n2nmn_amd.nmn3_assembler import Assembler
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES

LAYOUTS = {
    'scene_exist': ['_Scene', '_Exist'],
    'find_exist': ['_Find', '_Exist'],
    'find_count': ['_Find', '_Count'],
    'find2_equal': ['_Find', '_Find', '_EqualNum'],
    'find_describe': ['_Find', '_Describe'],
    'scene_describe': ['_Scene', '_Describe'],
    'find_tr_count': ['_Find', '_Transform', '_Count'],
    'scene_tr_exist': ['_Scene', '_Transform', '_Exist'],
    'find_fsp_count': ['_Find', '_FindSameProperty', '_Count'],
    'find2_same': ['_Find', '_Find', '_SameProperty'],
    'tpl8': ['_Find', '_Transform', '_Find', '_Transform', '_And', '_Filter', '_Count'],
    'invalid': [],
}


def main():
    ns = [int(a) for a in sys.argv[1:]] or [64]
    for N in ns:
        d = Dims(N=N)
        asm = Assembler(list(CLEVR_MODULE_NAMES))
        eng = Engine(d, asm)
        eng.load_weights(synth.make_weights(d, seed=0))
        batch = synth.make_inputs(d, seed=1)
        feat = torch.as_tensor(batch['image_feat_batch']).cuda()
        wv = torch.randn((d.T_decoder, N, d.embed_dim_txt), device='cuda') * 0.3
        for name, layout in LAYOUTS.items():
            toks = np.array([asm.module_list2tokens(layout, d.T_decoder)] * N, np.int32).T.copy()
            if not layout:
                toks[:] = 0
            tok = torch.as_tensor(toks).cuda()
            eng.conv_image(feat, tok, d.T_decoder)
            for _ in range(3):
                eng.execute_tokens(tok, feat, wv, conv_done=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                eng.execute_tokens(tok, feat, wv, conv_done=True)
            e1.record()
            torch.cuda.synchronize()
            print('N=%4d %-16s %8.2f us per walker launch (back-to-back)' %
                  (N, name, e0.elapsed_time(e1) * 1e3 / reps), flush=True)


if __name__ == "__main__" and not os.environ.get("WALK_TIMELINE"):
    main()


def timeline(layout_name='tpl8', N=64):
    """per-node phase times (shader clocks of thread 0) of one walker launch"""
    import ctypes as C
    from n2nmn_amd import _lib
    d = Dims(N=N)
    asm = Assembler(list(CLEVR_MODULE_NAMES))
    eng = Engine(d, asm)
    eng.load_weights(synth.make_weights(d, seed=0))
    batch = synth.make_inputs(d, seed=1)
    feat = torch.as_tensor(batch['image_feat_batch']).cuda()
    wv = torch.randn((d.T_decoder, N, d.embed_dim_txt), device='cuda') * 0.3
    layout = LAYOUTS[layout_name]
    toks = np.array([asm.module_list2tokens(layout, d.T_decoder)] * N, np.int32).T.copy()
    tok = torch.as_tensor(toks).cuda()
    tl = torch.zeros((N, 32, 4), dtype=torch.int64, device='cuda')
    eng.conv_image(feat, tok, d.T_decoder)
    for _ in range(3):
        eng.execute_tokens(tok, feat, wv, conv_done=True)
    _lib.check(eng._lib.n2nmn_debug_walk_timeline(eng._ctx, tl.data_ptr()))
    eng.execute_tokens(tok, feat, wv, conv_done=True)
    torch.cuda.synchronize()
    _lib.check(eng._lib.n2nmn_debug_walk_timeline(eng._ctx, None))
    t = tl.cpu().numpy()[0]
    t0 = t[0, 0]
    for i, name in enumerate(layout):
        a, b, c, e = [int(x - t0) for x in t[i]]
        print('%-20s start %7d  text %6d  pool+fc_att %6d  op %6d  (clocks)' %
              (name, a, b - a, c - b, e - c))


if __name__ == '__main__' and os.environ.get('WALK_TIMELINE'):
    timeline(os.environ['WALK_TIMELINE'])

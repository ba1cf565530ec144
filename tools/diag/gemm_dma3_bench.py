#!/usr/bin/env python3
"""gemm_dma3_kernel (split-operand bf16, opt-in mode) against gemm_dma_kernel (exact fp32) on the dense
contractions of a pass, through n2nmn_debug_gemm (n2nmn_debug_set "debug_gemm_b3" = launches in the bf16x3
form, negative = launches in the fp32 form): max error against torch fp64, us per launch, TFLOP/s."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from n2nmn_amd import _lib, synth                      # noqa: E402
from n2nmn_amd.engine import Engine                    # noqa: E402
from n2nmn_amd.nmn3_assembler import Assembler         # noqa: E402
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES     # noqa: E402

d = Dims(N=16)
eng = Engine(d, Assembler(list(CLEVR_MODULE_NAMES)))
L = eng._lib
dev = eng.device
SHAPES = [('conv_image 1024 images', 153600, 250, 512), ('encoder_h_transform', 25600, 512, 512),
          ('q = out . W_a', 20480, 512, 512), ('models_vqa conv_image 256 images', 50176, 1024, 2064),
          ('ragged', 1000, 250, 300)]
g = torch.Generator(device='cpu').manual_seed(1)
for name, M, N, K in SHAPES:
    A = (torch.randn((M, K), generator=g) * torch.rand((M, 1), generator=g) * 3).to(dev)
    B = (torch.randn((K, N), generator=g) / np.sqrt(K)).to(dev)
    bias = torch.randn((N,), generator=g).to(dev)
    ref = (A[:4096].double() @ B.double() + bias.double()).cpu().numpy()
    out = {}
    for mode in ('fp32', 'bf16x3'):
        Cbuf = torch.zeros((M, N), device=dev)

        def call(n):
            eng.debug_set('debug_gemm_b3', n if mode == 'bf16x3' else -n)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _lib.check(L.n2nmn_debug_gemm(eng._ctx, A.data_ptr(), B.data_ptr(), bias.data_ptr(),
                                          Cbuf.data_ptr(), M, N, K, eng.stream()))
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        call(1)
        t1 = min(call(1) for _ in range(3))
        n = 41
        tn = min(call(n) for _ in range(3))
        us = (tn - t1) / (n - 1) * 1e6
        err = float(np.abs(Cbuf[:4096].cpu().numpy() - ref).max())
        out[mode] = (us, err)
        print('%-34s %-7s M=%6d N=%4d K=%4d  %8.1f us  %6.1f TFLOP/s  max |err| vs fp64 %.2e' %
              (name, mode, M, N, K, us, 2.0 * M * N * K / us / 1e6, err))
    print('    bf16x3 / fp32 time: %.2f' % (out['bf16x3'][0] / out['fp32'][0]))
eng.debug_set('debug_gemm_b3', None)

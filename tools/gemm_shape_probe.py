"""Kernel time of gemm_pk on given [M, K] x [K, N] shapes (run under rocprofv3 --kernel-trace and
read the gemm_pk_kernel rows in launch order): how well the LDS-staged 64x64 tile does on the
recurrent step's shapes at super-bucket sizes."""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')
import torch
from n2nmn_amd import synth
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
from n2nmn_amd.nmn3_assembler import Assembler
from n2nmn_amd.engine import Engine

shapes = [(512, 512, 2048), (512, 1024, 2048), (1024, 1024, 2048), (512, 1024, 4096)]
eng = Engine(Dims(N=64), Assembler(list(CLEVR_MODULE_NAMES)))
for (M, K, N) in shapes:
    A = torch.randn((M, K), device='cuda'); B = torch.randn((K, N), device='cuda')
    for _ in range(5):
        Cm = eng.gemm(A, B)
    err = (Cm - A @ B).abs().max().item()
    print(M, K, N, 'max err vs torch', err, flush=True)

#!/usr/bin/env python3
"""gemm_tn (weight-gradient GEMM, C += A^T . B) on the shapes of one CLEVR training step, through
n2nmn_debug_gemm_tn.  Prints us per launch and TFLOP/s; used for same-box A/B of kernel revisions
(profiles/r01_gemm_tn.txt)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from n2nmn_amd import _lib  # noqa: E402
from n2nmn_amd.engine import Engine  # noqa: E402
from n2nmn_amd.nmn3_assembler import Assembler  # noqa: E402
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES  # noqa: E402

# (label, M, N, R, launches per step)
SHAPES = [
    ('enc dW (h^T dz), active rows', 512, 2048, 1600, 3),
    ('dec dW (h^T dz)', 512, 2048, 640, 3),
    ('conv find/fsp image', 512, 250, 9600, 2),
    ('conv find/fsp image (deep level)', 512, 250, 19200, 0),
    ('W_eht', 512, 512, 2880, 1),
    ('att W', 512, 512, 640, 1),
    ('token W', 512, 15, 640, 2),
    ('enc W0x (emb^T dxtab)', 300, 2048, 82, 1),
    ('fc_text (1 of 5)', 300, 250, 200, 1),
]


def main():
    d = Dims()
    eng = Engine(d, Assembler(list(CLEVR_MODULE_NAMES)))
    dev = eng.device
    total = 0.0
    for label, M, N, R, per_step in SHAPES:
        ldb = (N + 3) // 4 * 4
        A = torch.randn(R, M, device=dev)
        B = torch.randn(R, ldb, device=dev)
        Cm = torch.zeros(M, N, device=dev)

        def run():
            _lib.check(eng._lib.n2nmn_debug_gemm_tn(eng._ctx, A.data_ptr(), M, M, B.data_ptr(), ldb, N,
                                                    R, Cm.data_ptr(), N, None, None, None, 0,
                                                    eng.stream()))
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        reps = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = torch.cuda.ExternalStream(eng.stream()) if hasattr(torch.cuda, 'ExternalStream') else None
        with torch.cuda.stream(st):
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        # check
        Cm.zero_()
        run()
        torch.cuda.synchronize()
        ref = A.double().t() @ B[:, :N].double()
        err = float((Cm.double() - ref).abs().max() / ref.abs().max())
        tf = 2.0 * M * N * R / us * 1e-6
        total += us * per_step
        print('%-36s M %4d N %4d R %5d  %8.2f us  %6.1f TFLOP/s  (%4.1f %% of 157.3)  err %.1e' % (
            label, M, N, R, us, tf, 100 * tf / 157.3, err))
    print('sum over one step (listed launches): %.1f us' % total)


if __name__ == '__main__':
    main()

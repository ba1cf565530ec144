"""Unit parity of the training step's generic kernels through the C-ABI debug entries: gemm_tn
(weight-gradient GEMM: ragged shapes, row gather, one-hot operand, row selection, split-R) and
colsum, against numpy fp64."""
import ctypes as C

import numpy as np
import pytest

from n2nmn_amd import _lib
from util import t2n

pytestmark = pytest.mark.gpu


def _tn(eng, A, B, Cinit, a_idx=None, onehot=None, sel=None, sel_val=0, M=None):
    import torch
    dev = eng.device
    tB = torch.as_tensor(B, dtype=torch.float32, device=dev).contiguous()
    tC = torch.as_tensor(Cinit, dtype=torch.float32, device=dev).contiguous()
    tA = torch.as_tensor(A, dtype=torch.float32, device=dev).contiguous() if A is not None else None
    ti = torch.as_tensor(a_idx, dtype=torch.int32, device=dev) if a_idx is not None else None
    to = torch.as_tensor(onehot, dtype=torch.int32, device=dev) if onehot is not None else None
    ts = torch.as_tensor(sel, dtype=torch.int32, device=dev) if sel is not None else None
    R = tB.shape[0]
    M = tA.shape[1] if M is None else M
    _lib.check(eng._lib.n2nmn_debug_gemm_tn(
        eng._ctx, tA.data_ptr() if tA is not None else None, tA.shape[1] if tA is not None else 0, M,
        tB.data_ptr(), tB.shape[1], Cinit.shape[1], R, tC.data_ptr(), tC.shape[1],
        ti.data_ptr() if ti is not None else None, to.data_ptr() if to is not None else None,
        ts.data_ptr() if ts is not None else None, sel_val, eng.stream()))
    torch.cuda.synchronize()
    return t2n(tC)


@pytest.mark.parametrize('M,N,R', [(300, 250, 150), (512, 2048, 2880), (64, 15, 640), (4, 28, 1),
                                   (512, 250, 9600), (132, 67, 33)])
def test_gemm_tn_plain(clevr_engine, M, N, R):
    eng = clevr_engine[0]
    rng = np.random.default_rng(M + N + R)
    ldb = (N + 3) // 4 * 4
    A = rng.standard_normal((R, M))
    B = np.zeros((R, ldb)); B[:, :N] = rng.standard_normal((R, N))
    C0 = rng.standard_normal((M, N))
    got = _tn(eng, A, B, C0)
    want = C0 + A.T @ B[:, :N]
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 2e-5 * scale + 1e-6


def test_gemm_tn_gather_select_onehot(clevr_engine):
    eng = clevr_engine[0]
    rng = np.random.default_rng(7)
    R, M, N, S = 200, 300, 250, 500
    src = rng.standard_normal((S, M))
    idx = rng.integers(0, S, size=R)
    B = rng.standard_normal((R, 256)); B[:, N:] = 0
    sel = rng.integers(0, 5, size=R)
    C0 = np.zeros((M, N))
    got = _tn(eng, src, B, C0, a_idx=idx, sel=sel, sel_val=3)
    keep = sel == 3
    want = src[idx][keep].T @ B[keep, :N]
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max() + 1e-6
    # one-hot operand: dxtab[v] = sum_{r: tok[r]==v} dz[r]
    V, Ncol = 82, 2048
    tok = rng.integers(0, V, size=2880)
    dz = rng.standard_normal((2880, Ncol))
    got = _tn(eng, None, dz, np.zeros((V, Ncol)), onehot=tok, M=V)
    want = np.zeros((V, Ncol))
    np.add.at(want, tok, dz)
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max() + 1e-6


@pytest.mark.parametrize('R,ncols,ld', [(2880, 2048, 2048), (640, 15, 16), (17, 250, 256), (1, 1, 4)])
def test_colsum(clevr_engine, R, ncols, ld):
    import torch
    eng = clevr_engine[0]
    rng = np.random.default_rng(R + ncols)
    src = rng.standard_normal((R, ld))
    sel = rng.integers(0, 3, size=R)
    for use_sel in (False, True):
        ts = torch.as_tensor(src, dtype=torch.float32, device=eng.device)
        td = torch.ones(ncols, dtype=torch.float32, device=eng.device)
        tsel = torch.as_tensor(sel, dtype=torch.int32, device=eng.device)
        _lib.check(eng._lib.n2nmn_debug_colsum(eng._ctx, ts.data_ptr(), R, ncols, ld,
                                               tsel.data_ptr() if use_sel else None, 1, td.data_ptr(),
                                               eng.stream()))
        torch.cuda.synchronize()
        rows = src[sel == 1] if use_sel else src
        want = 1.0 + rows[:, :ncols].sum(0)
        assert np.abs(t2n(td) - want).max() <= 2e-5 * max(1.0, np.abs(want).max())

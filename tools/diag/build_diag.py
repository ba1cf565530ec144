#!/usr/bin/env python3
"""Diagnostic builds of the library: csrc/ compiled with -DN2NMN_DIAG (+ sub-flags) into
tools/diag/lib/libn2nmn_hip_diag[_<variant>].so.

The product library (n2nmn_amd/lib/libn2nmn_hip.so) carries none of this: the in-kernel LDS verification, the
per-anomaly hardware-id records and the n2nmn_diag_* exports exist only behind the macros
(csrc/kernels_gemm_dma3.hip).  `use_diag_lib()` points n2nmn_amd at a diagnostic library for one process.
Only kernels_gemm_dma3.hip differs between the variants; every other object is compiled once.

    python tools/diag/build_diag.py [variant ...]        (default: all)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libn2nmn_hip_diag.so')

# variant -> extra flags for kernels_gemm_dma3.hip (N2NMN_DIAG = the n2nmn_diag_* exports, always on)
VARIANTS = {
    'full': ['-DN2NMN_DIAG_WG', '-DN2NMN_DIAG_VERIFY', '-DN2NMN_DIAG_ACC2', '-DN2NMN_DIAG_LAUNCH'],
    'lite': [],                                   # the product's kernel and launch path + the exports
    'launch': ['-DN2NMN_DIAG_LAUNCH'],            # hipFuncSetAttribute before every launch
    'wg': ['-DN2NMN_DIAG_WG'],                    # per-workgroup LDS_ALLOC / HW_ID records
    'acc2': ['-DN2NMN_DIAG_ACC2'],                # a second accumulator chain (more VGPRs)
    'verify': ['-DN2NMN_DIAG_VERIFY'],            # LDS contents against a direct global read
    'lb2': ['-DN2NMN_DIAG_WAVES_PER_EU=2'],       # __launch_bounds__(512, 2)
    'twice': ['-DN2NMN_DIAG_VERIFY', '-DN2NMN_DIAG_TWICE'],   # the K loop twice over the same operands; input stability
    'sums': ['-DN2NMN_DIAG_SUMS'],                # per-launch checksums of A / tokens / C (which launch differs)
    'endwait': ['-DN2NMN_DIAG_END=1'],            # s_waitcnt vmcnt(0) behind the epilogue stores
    'endfence': ['-DN2NMN_DIAG_END=2'],           # __threadfence() behind the epilogue stores
    'endsleep': ['-DN2NMN_DIAG_END=3'],           # control: a delay of the same order, no memory effect
    'epilds': ['-DN2NMN_DMA3_EPI_LDS'],           # epilogue staged through LDS: whole 128-byte lines per store
    'nop': ['-DN2NMN_DIAG_NOP'],                  # (experiments: see kernels_gemm_dma3.hip)
}


def lib_of(variant):
    return LIB if variant == 'full' else os.path.join(LIBDIR, 'libn2nmn_hip_diag_%s.so' % variant)


def _cc(b, src, obj, extra):
    return [b.hipcc(), '--offload-arch=' + b.ARCH, '-O3', '-std=c++17', '-fPIC', '-x', 'hip', '-DN2NMN_DIAG=1',
            '-Wall', '-Wno-unused-function', '-I' + b.CSRC] + extra + ['-c', src, '-o', obj]


def build(variants=None, verbose=True):
    from n2nmn_amd import build as b
    variants = list(variants or VARIANTS)
    objdir = os.path.join(LIBDIR, 'obj')
    os.makedirs(objdir, exist_ok=True)
    procs, common = [], []
    special = os.path.join(HERE, 'csrc', 'kernels_gemm_dma3.hip')     # (not a source of the product library)
    for src in b.sources():
        obj = os.path.join(objdir, os.path.basename(src) + '.o')
        common.append(obj)
        procs.append((src, subprocess.Popen(_cc(b, src, obj, []), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    vobjs = {}
    for v in variants:
        obj = os.path.join(objdir, 'kernels_gemm_dma3.%s.o' % v)
        vobjs[v] = obj
        procs.append((special + ' [' + v + ']', subprocess.Popen(_cc(b, special, obj, VARIANTS[v]),
                                                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    bad = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            bad = True
            sys.stderr.write('FAILED %s\n%s\n' % (src, out.decode(errors='replace')))
    if bad:
        raise RuntimeError('hipcc failed')
    for v in variants:
        subprocess.check_call([b.hipcc(), '--offload-arch=' + b.ARCH, '-shared', '-fPIC', '-o', lib_of(v)] + common +
                              [vobjs[v]])
        if verbose:
            print(lib_of(v))
    return [lib_of(v) for v in variants]


def use_diag_lib(which='1'):
    """Make n2nmn_amd._lib load a diagnostic library in this process (call before the first lib()).
    which: '1' / 'full' = the full diagnostic build, anything else = a variant name"""
    from n2nmn_amd import build as b
    lib = lib_of('full' if which in ('1', '', None) else which)
    if not os.path.exists(lib):
        raise RuntimeError('build it first: python tools/diag/build_diag.py')
    b.LIB = lib
    b.is_stale = lambda: False
    return lib


if __name__ == '__main__':
    build(sys.argv[1:] or None)

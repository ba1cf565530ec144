"""In-tree build of the gfx950 shared library (hipcc cross-compiles without a GPU).

    python -m n2nmn_amd.build [--force]

Produces n2nmn_amd/lib/libn2nmn_hip.so from csrc/*.hip + csrc/*.cpp.  The .so is git-ignored but
travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libn2nmn_hip.so')
ARCH = 'gfx950'


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.cpp')))


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, '*.h')) + \
        [os.path.join(HERE, '..', 'include', 'n2nmn.h')]


STAMP = os.path.join(LIBDIR, '.csrc.sha256')


def source_digest() -> str:
    """sha256 over the names and contents of everything the library is built from"""
    import hashlib
    h = hashlib.sha256()
    for p in sorted(_deps()):
        h.update(os.path.basename(p).encode() + b'\0')
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def is_stale() -> bool:
    """The library is stale when the sources it was built from are not the sources on disk.  Decided
    by CONTENT (digest recorded next to the .so at build time), not by mtimes: a checkout or rsync that
    does not preserve timestamps must not turn a working install into a rebuild -- or, on a box
    without hipcc, into an import error.  Without a recorded digest (library from an older build)
    the mtimes decide."""
    if not os.path.exists(LIB):
        return True
    try:
        with open(STAMP) as f:
            return f.read().strip() != source_digest()
    except OSError:
        t = os.path.getmtime(LIB)
        return any(os.path.getmtime(p) > t for p in _deps())


def hipcc() -> str:
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: the HIP extension cannot be built')
    return exe


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile csrc/ into LIB.  Several processes may get here at once (one rank per GPU under
    torch.distributed.run, all finding a stale library): an exclusive file lock serialises them, the
    losers re-check staleness and return, and the library appears atomically (os.replace)."""
    if not force and not is_stale():
        return LIB
    import fcntl
    os.makedirs(LIBDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, '.build.lock'), 'a') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():
                return LIB
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose: bool) -> str:
    objs = []
    procs = []
    objdir = os.path.join(LIBDIR, 'obj')
    os.makedirs(objdir, exist_ok=True)
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + '.o')
        objs.append(obj)
        cmd = [hipcc(), '--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-x', 'hip',
               '-Wall', '-Wno-unused-function', '-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out and verbose:
            sys.stdout.write(out.decode(errors='replace'))
        if p.returncode != 0:
            failed = True
            sys.stderr.write('FAILED: %s\n%s\n' % (src, out.decode(errors='replace')))
    if failed:
        raise RuntimeError('hipcc failed')
    tmp = LIB + '.tmp.%d' % os.getpid()
    cmd = [hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', tmp] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    digest = source_digest()
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
        with open(STAMP, 'w') as f:
            f.write(digest + '\n')
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))

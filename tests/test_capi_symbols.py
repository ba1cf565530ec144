"""The C-ABI library loads (no GPU needed) and exports every symbol include/n2nmn.h declares."""
import ctypes
import os
import re

from n2nmn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'n2nmn.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(n2nmn_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 25
    L = ctypes.CDLL(_lib.lib_path()) if os.path.exists(_lib.lib_path()) else _lib.lib()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_binding_table_covers_header():
    bound = {s[0] for s in _lib.SYMBOLS}
    assert bound == set(_declared())


def test_version_and_error_strings():
    L = _lib.lib()
    assert b'gfx950' in L.n2nmn_version()
    assert L.n2nmn_program_status(None, 0, None, None, None) < 0
    assert b'program_status' in L.n2nmn_last_error()


def test_struct_sizes_match_header():
    assert ctypes.sizeof(_lib.Dims) == 17 * 4
    assert ctypes.sizeof(_lib.Node) == 8 * 4
    # 2 ptr + 4 int32 + 13 ptr
    # ... + flags (4 + 4 pad) + image_feat + 2 dropout pointers + seq_length_host + gt_length_host
    assert ctypes.sizeof(_lib.Seq2SeqIO) == 2 * 8 + 4 * 4 + 13 * 8 + 8 + 8 + 2 * 8 + 8 + 8
    # ctx + 5 buffers + atts / input_seq / seq_length
    assert ctypes.sizeof(_lib.WalkBatch) == 9 * 8
    # 2 ptr + 3 int32 (+4 pad) + 3 ptr + float (+4 pad) + 3 ptr
    # ... + objective (4 + 4 pad) + expr_validity + 3 floats (+ 4 pad) + baseline
    assert ctypes.sizeof(_lib.TrainIO) == 2 * 8 + 16 + 3 * 8 + 8 + 3 * 8 + 8 + 8 + 16 + 8 + 4 * 8

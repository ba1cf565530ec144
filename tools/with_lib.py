#!/usr/bin/env python3
"""Developer hook: run a script of this repo against another build of the library
(tools/with_lib.py path/to/libvariant.so script.py [args...]) for same-box A/B of kernel variants."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from n2nmn_amd import build as _b  # noqa: E402

_b.LIB = os.path.abspath(sys.argv[1])
_b.is_stale = lambda: False
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')

import json
import os
import sys

import numpy as np
import pytest

# several tests import modules straight from the read-only reference checkout: never leave
# __pycache__ directories behind in it
sys.dont_write_bytecode = True

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no HIP device in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    with open(os.path.join(GOLDEN, 'assembler_golden.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def clevr_engine():
    """One Engine with the CLEVR eval dims and seed-0 synthetic weights (GPU tests only)."""
    from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.engine import Engine
    from n2nmn_amd import synth
    d = Dims()
    asm = Assembler(list(CLEVR_MODULE_NAMES))
    eng = Engine(d, asm)
    w = synth.make_weights(d, seed=0)
    eng.load_weights(w)
    return eng, d, asm, w

"""n2nmn_amd: MI355X-native N2NMN (CLEVR) forward hot path behind the reference's operator API.

Python host code (this package) -> C-ABI (include/n2nmn.h, lib/libn2nmn_hip.so) -> hand-written
gfx950 HIP kernels (csrc/).  See DESIGN.md / INTEGRATION.md.
"""
from .spec import Dims, CLEVR_MODULE_NAMES, INVALID_EXPR  # noqa: F401

__all__ = ['Dims', 'CLEVR_MODULE_NAMES', 'INVALID_EXPR']
__version__ = '0.1'

"""conex_amd -- MI355X-native Newton-step KKT path of the conex cone solver.

The product is the shared library ``conex_amd/lib/libconex.so`` (host C++ + hand-written HIP
for gfx950) which exports two C-ABIs:

* ``include/conex.h``          -- the reference's own 21-function ``CONEX_*`` interface
* ``include/conex_kkt_hip.h``  -- the device-resident Newton-step path (``cxk_*``)

This Python package is only a ctypes door onto that library for tests and ``bench.py``;
there is no Python or CPU fallback: importing :mod:`conex_amd.kkt` raises if the library has
not been built (``python __graft_entry__.py`` or ``make -C conex_amd/csrc``).
"""
from .kkt import KktContext, load_library, LIB_PATH  # noqa: F401

__all__ = ["KktContext", "load_library", "LIB_PATH"]

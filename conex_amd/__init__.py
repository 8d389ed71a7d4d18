"""conex_amd -- MI355X-native Newton-step KKT path of the conex cone solver.

The product is the shared library ``conex_amd/lib/libconex.so`` (host C++ + hand-written HIP
for gfx950) which exports two C-ABIs:

* ``include/conex.h``          -- the reference's own 21-function ``CONEX_*`` interface
* ``include/conex_kkt_hip.h``  -- the device-resident Newton-step path (``cxk_*``)

This Python package is a ctypes door onto that library: :mod:`conex_amd.kkt` (the ``cxk_*`` path,
used by the tests and ``bench.py``), :mod:`conex_amd.capi` (the ``CONEX_*`` table) and
:mod:`conex_amd.program` (a Python-3 ``Conex`` class with the surface of the reference's
interfaces/python/ConexProgram.py).  There is no Python or CPU fallback: importing :mod:`conex_amd.kkt` raises if the library has
not been built (``python __graft_entry__.py`` or ``make -C conex_amd/csrc``).
"""
from .kkt import KktContext, load_library, LIB_PATH  # noqa: F401

__all__ = ["KktContext", "load_library", "LIB_PATH"]

"""pb-starphase_amd -- MI355X (gfx950) native hot path of pb-StarPhase.

The product is the C-ABI shared library ``libstarphase_hip.so`` (sources in ``csrc/``, declared in
``include/starphase_hip.h``).  This Python package is only the thin ctypes binding used by the tests and
``bench.py``; it never falls back to a CPU implementation: if the library or a gfx950 device is missing,
import of ``ffi`` / creation of a context raises.

The directory name contains a hyphen (it mirrors the reference repository name), so it is loaded through
``__graft_entry__.load_package()`` and registered as module ``pb_starphase_amd``.
"""
from . import ffi          # noqa: F401
from . import database     # noqa: F401
from .ffi import (Context, SeqSet, HlaDb, StarphaseError, lib_path)   # noqa: F401

"""epic_amd -- MI355X-native log-space harmonic relaxation behind the libepic C-ABI.

The product is ``epic_amd/lib/libepic.so`` (hand-written HIP for gfx950, sources in epic_amd/csrc).  The python
modules mirror the reference's ``libepic/python/epic`` package on top of it:

    epic_harmonic -- ctypes binding (EpicHarmonic, _epic)
    harmonic      -- Harmonic.solve()
    harmonic_map  -- image -> grid loader
    synthetic     -- seeded synthetic grids for the benchmark configs
"""
from .epic_harmonic import EpicHarmonic, LIB_PATH, _epic  # noqa: F401
from .harmonic import Harmonic  # noqa: F401
from .harmonic_map import HarmonicMap  # noqa: F401

__version__ = "0.1.0"

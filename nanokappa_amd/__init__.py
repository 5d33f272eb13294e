"""nanokappa_amd -- MI355X-native engine for Nano-kappa's Population timestep loop.

Host side (Python) mirrors the reference's Geometry / Phonon / Population surface; the particle state and the
advect-scatter-tally loop live in libnanokappa_hip.so (include/nanokappa_hip.h), bound with ctypes.
"""
from .constants import Constants          # noqa: F401
from .phonon import Phonon                # noqa: F401
from .mesh import Mesh                    # noqa: F401
from .geometry import Geometry            # noqa: F401
from .engine import Engine, NkError       # noqa: F401
from .population import Population        # noqa: F401

__all__ = ['Constants', 'Phonon', 'Mesh', 'Geometry', 'Engine', 'NkError', 'Population']

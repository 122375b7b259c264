"""fiesta_amd -- MI355X-native incremental ESDF engine behind FIESTA's ``ESDFMap`` operator API.

The product is ``libfiesta_hip.so`` (hand-written HIP for gfx950 behind the C ABI of
include/fiesta_hip.h); this package is the thin host-side mirror of the reference interface.
"""
from ._lib import FiestaHipError, LIB_PATH, device_count, load  # noqa: F401
from .esdf_map import D2_INF, INFINITY, UNDEFINED, ESDFMap, signed_distance  # noqa: F401

__all__ = ["ESDFMap", "signed_distance", "FiestaHipError", "device_count", "load", "LIB_PATH", "UNDEFINED", "INFINITY", "D2_INF"]

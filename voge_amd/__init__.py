"""voge_amd: MI355X-native ray-trace + aggregation hot path of the VoGE renderer.

Public API mirrors the reference package (`VoGE.Renderer`, `VoGE.RayTracing`,
`VoGE.Aggregation`, `VoGE.Meshes`, `VoGE.Converter`); the `VoGE/` directory at the repo root
re-exports these modules under the reference's import names.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401  (ctypes binding; the .so itself is loaded on first use)
from . import Aggregation, Meshes, RayTracing, Renderer, Sampler, Utils, cameras  # noqa: F401
from . import Converter  # noqa: F401
from .Converter import IO, Converters, Cuboid  # noqa: F401


def load_library():
    """Load libvoge_hip.so now (raises VogeHipError if it is missing)."""
    return _lib.load()

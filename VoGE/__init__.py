"""`VoGE` import names of the reference package, served by the MI355X implementation in
`voge_amd` (every module here is an alias, there is no second implementation)."""
import sys as _sys

import voge_amd as _impl
from voge_amd import Aggregation, Converter, Meshes, RayTracing, Renderer, Sampler, Utils  # noqa: F401
from voge_amd.Converter import IO, Converters, Cuboid  # noqa: F401

__version__ = _impl.__version__
for _name in ("Aggregation", "Meshes", "RayTracing", "Renderer", "Sampler", "Utils", "Converter", "cameras"):
    _sys.modules[__name__ + "." + _name] = getattr(_impl, _name)
for _name in ("IO", "Converters", "Cuboid"):
    _sys.modules[__name__ + ".Converter." + _name] = getattr(_impl.Converter, _name)

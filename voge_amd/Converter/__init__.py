"""Mesh / point-cloud -> Gaussian converters and OFF/GOFF IO: host-side
pre-processing that feeds the renderer, with the names of VoGE/Converter/."""
from . import Converters, Cuboid, IO  # noqa: F401

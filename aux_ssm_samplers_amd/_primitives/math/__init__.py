"""reference: aux_samplers/_primitives/math/__init__.py:1-2."""
from . import mvn
from .utils import normalize, logsubexp, log1mexp

__all__ = ["mvn", "normalize", "logsubexp", "log1mexp"]

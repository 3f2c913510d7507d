from .utils import normalize

__all__ = ["normalize"]

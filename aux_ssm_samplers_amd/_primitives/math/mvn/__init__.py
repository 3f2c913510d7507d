"""reference: aux_samplers/_primitives/math/mvn/__init__.py:1."""
from .base import logpdf, rvs, tril_log_det, get_optimal_covariance

__all__ = ["logpdf", "rvs", "tril_log_det", "get_optimal_covariance"]

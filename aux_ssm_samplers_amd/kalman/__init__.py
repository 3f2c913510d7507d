from .generic import get_kernel, KalmanSampler
from .models import LGConcatModel

__all__ = ["get_kernel", "KalmanSampler", "LGConcatModel"]

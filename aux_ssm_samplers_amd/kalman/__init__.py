from .generic import get_kernel, KalmanSampler
from .models import LGConcatModel, SVModel

__all__ = ["get_kernel", "KalmanSampler", "LGConcatModel", "SVModel"]

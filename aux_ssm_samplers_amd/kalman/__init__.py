from .generic import get_kernel, KalmanSampler
from .models import LGConcatModel, SVModel, LorenzModel

__all__ = ["get_kernel", "KalmanSampler", "LGConcatModel", "SVModel", "LorenzModel"]

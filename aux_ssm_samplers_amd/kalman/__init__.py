from .generic import get_kernel, KalmanSampler, DeviceChains
from .models import LGConcatModel, SVModel, LorenzModel

__all__ = ["get_kernel", "KalmanSampler", "DeviceChains", "LGConcatModel", "SVModel", "LorenzModel"]

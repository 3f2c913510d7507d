"""Caller-side helpers of the sweep (reference: aux_samplers/common.py)."""
import numpy as np


def delta_adaptation(delta, target_rate, acceptance_rate, adaptation_rate, min_delta=1e-20, max_delta=1e20):
    """Robbins-Monro style step-size rule of the auxiliary samplers (common.py:4-32):
    delta <- clip(delta * exp(adaptation_rate * (acceptance_rate - target_rate)), min_delta, max_delta).
    Scalars or arrays (one delta per time step / per chain); host arithmetic -- it runs once per adaptation window on
    the acceptance flags the sweep leaves in `DeviceChains.accepted`."""
    rate = np.exp(np.asarray(adaptation_rate, np.float64) * (np.asarray(acceptance_rate, np.float64) - target_rate))
    out = np.clip(np.asarray(delta, np.float64) * rate, min_delta, max_delta)
    return float(out) if np.ndim(out) == 0 else out

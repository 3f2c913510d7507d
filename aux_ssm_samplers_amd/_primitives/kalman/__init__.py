from .base import LGSSM, posterior_logpdf, log_likelihood, prior_logpdf, joint_logpdf
from .filtering import filtering
from .sampling import sampling

__all__ = ["LGSSM", "posterior_logpdf", "log_likelihood", "prior_logpdf", "joint_logpdf", "filtering", "sampling"]

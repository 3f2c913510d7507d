// instantiation unit: real = float, dx = 1, every dy in 1..8 (filter, log-likelihood pass, joint logpdf) + the sampler
#include "kernels.hip.h"
AX_DEFINE_UNIT(f32_d1, float, 1)

// instantiation unit: real = float, dx = 4, every dy in 1..8 (filter, log-likelihood pass, joint logpdf) + the sampler
#include "kernels.hip.h"
AX_DEFINE_UNIT(f32_d4, float, 4)

// instantiation unit: real = double, dx = 4, every dy in 1..8 (filter, log-likelihood pass, joint logpdf) + the sampler
#include "kernels.hip.h"
AX_DEFINE_UNIT(f64_d4, double, 4)

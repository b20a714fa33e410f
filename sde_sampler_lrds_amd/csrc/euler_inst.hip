// EulerIntegrator kernels for SDEs without a drift net (euler_kernel.hpp), all four feature-tile counts.
#include "euler_kernel.hpp"
SD_DEFINE_EULER(1)
SD_DEFINE_EULER(2)
SD_DEFINE_EULER(4)
SD_DEFINE_EULER(8)

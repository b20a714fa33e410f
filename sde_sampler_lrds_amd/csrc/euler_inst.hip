// EulerIntegrator kernels for SDEs without a drift net (euler_kernel.hpp), every feature-tile count.
#include "euler_kernel.hpp"
SD_DEFINE_EULER(1)
SD_DEFINE_EULER(2)
SD_DEFINE_EULER(3)
SD_DEFINE_EULER(4)
SD_DEFINE_EULER(5)
SD_DEFINE_EULER(6)
SD_DEFINE_EULER(7)
SD_DEFINE_EULER(8)

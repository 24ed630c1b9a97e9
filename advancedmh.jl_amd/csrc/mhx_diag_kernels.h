// mhx_diag_kernels.h -- chain diagnostics on the device sample buffer.  (filled in below)
#pragma once
#include "mhx_device_math.h"

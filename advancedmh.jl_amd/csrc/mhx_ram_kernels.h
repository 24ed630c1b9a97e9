// mhx_ram_kernels.h -- robust adaptive Metropolis kernels.  (filled in below)
#pragma once
#include "mhx_targets.h"

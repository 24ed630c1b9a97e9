// mhx_emcee_kernels.h -- affine-invariant ensemble (stretch move) kernels.  (filled in below)
#pragma once
#include "mhx_targets.h"

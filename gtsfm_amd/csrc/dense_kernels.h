// Launchers and weight packers of the fp32-MFMA dense kernels: convolution (conv_kernels.h) and GEMM (gemm_kernels.h). Two
// headers so that a change to the GEMM's parameter block does not recompile the convolution kernel (minutes).
#pragma once

#include "conv_kernels.h"
#include "gemm_kernels.h"

// Oracle-build-only (TEST INFRASTRUCTURE): what nvcc puts in front of every translation unit --
// the execution-space keywords (empty on the host) and the runtime declarations -- force-included
// (`-include`) by oracle/ref/build_ref.py so that the reference's CUDA-flavoured headers parse.
#pragma once
#define __device__
#define __host__
#define __global__
#include "cuda_runtime.h"

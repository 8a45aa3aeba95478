// Oracle-build-only stand-in (TEST INFRASTRUCTURE): see cuda_runtime.h beside this file.
#pragma once
#include "cuda_runtime.h"

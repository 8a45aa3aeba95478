// Built-in ristretto255 Pedersen generators g_i (reference: sxt/seqcommit/generator):
//   device derivation kernel   <- gpu_generator.cc:34-57 (K15)
//   init-time cache            <- precomputed_generators.cc:32-91
//   one-commit prefix sums     <- cpu_one_commitments.cc:29-59, precomputed_one_commitments.cc:56-70
#pragma once

#include "blitzar_amd/csrc/base/device.h"
#include "blitzar_amd/csrc/curve/ed29.h"

namespace bz {
// d_out[i] = g_{first + i} as raw extended coordinates (limb-identical to the reference)
void builtin_generators_enqueue(ed_point* d_out, u64 first, u64 n, hipStream_t stream);
} // namespace bz

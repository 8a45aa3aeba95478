// gfx950 code object of k_accumulate for curve25519 generators in their Z = 1 form (resident sets,
// built-in generators, fixed-base handles): a translation unit of its own because this loop, unlike
// the caller-generators one, fits three waves per SIMD under the iterative-ilp scheduling strategy
// (159 VGPRs, no spill, 1361 -> 1172 instructions per addition; blitzar_amd/build.py, TU_FLAGS).
#include "blitzar_amd/csrc/msm/curve_traits.h"
#include "blitzar_amd/csrc/msm/kernels.h"

namespace bz {
BZ_ACCUMULATE_INSTANCE(, ed25519_niels_msm);
} // namespace bz

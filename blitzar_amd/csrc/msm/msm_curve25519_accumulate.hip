// gfx950 code object of k_accumulate for curve25519 (caller generators; the Z = 1 form of resident
// sets lives in msm_curve25519_niels_accumulate.hip): the bucket accumulation loop, 65-85 % of every
// MSM, in a translation unit of its own so that it can be compiled with the scheduling strategy
// that suits it (blitzar_amd/build.py, TU_FLAGS) without touching the other kernels of the curve.
#include "blitzar_amd/csrc/msm/curve_traits.h"
#include "blitzar_amd/csrc/msm/kernels.h"

namespace bz {
BZ_ACCUMULATE_INSTANCE(, ed25519_msm);
} // namespace bz

// gfx950 code object for the curve25519 MSM kernels (see curve_tu.h / kernels.h).
#include "blitzar_amd/csrc/msm/curve_tu.h"

namespace bz {
BZ_ACCUMULATE_INSTANCE(extern, ed25519_msm); // msm_curve25519_accumulate.hip
BZ_ACCUMULATE_INSTANCE(extern, ed25519_niels_msm); // msm_curve25519_niels_accumulate.hip
// Per-call caller generators keep the projective (Y+X, Y-X, Z, 2dT) addends: normalising them to
// Z = 1 on every call (k_prepare_addends_batched, one shared inversion per 1024 generators) makes
// k_accumulate 8 % faster (0.72 -> 0.66 ms at config 2) but the shared inversion is a 50-80 us
// dependent chain that every workgroup of the launch waits for at the same time: +0.25 ms of
// prepare, only partly hidden beside recode + sort (measured A/B on MI355X, profiles/round2_ab.md:
// 1.90 -> 1.79 ms with, 1.70 ms without the normalisation on the same kind of box).  Resident
// generator sets are normalised once, with the batched kernel.
const curve_vtable& curve25519_vtable() { return curve_tu<ed25519_msm, ed25519_niels_msm, ed25519_msm>::vtable(); }
} // namespace bz

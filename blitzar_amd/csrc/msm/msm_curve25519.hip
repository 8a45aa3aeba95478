// gfx950 code object for the curve25519 MSM kernels (see curve_tu.h / kernels.h).
#include "blitzar_amd/csrc/msm/curve_tu.h"

namespace bz {
const curve_vtable& curve25519_vtable() { return curve_tu<ed25519_msm, ed25519_niels_msm>::vtable(); }
} // namespace bz

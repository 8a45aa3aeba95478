// gfx950 code object for the bls12_381 MSM kernels (see curve_tu.h / kernels.h).
#include "blitzar_amd/csrc/msm/curve_tu.h"

namespace bz {
const curve_vtable& bls12_381_vtable() { return curve_tu<bls12_381_msm>::vtable(); }
} // namespace bz

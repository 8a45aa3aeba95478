// gfx950 code object for the bn254 MSM kernels (see curve_tu.h / kernels.h).
#include "blitzar_amd/csrc/msm/curve_tu.h"

namespace bz {
BZ_ACCUMULATE_INSTANCE(extern, bn254_msm); // msm_bn254_accumulate.hip
const curve_vtable& bn254_vtable() { return curve_tu<bn254_msm>::vtable(); }
} // namespace bz

// gfx950 code object for the grumpkin MSM kernels (see curve_tu.h / kernels.h).
#include "blitzar_amd/csrc/msm/curve_tu.h"

namespace bz {
BZ_ACCUMULATE_INSTANCE(extern, grumpkin_msm); // msm_grumpkin_accumulate.hip
const curve_vtable& grumpkin_vtable() { return curve_tu<grumpkin_msm>::vtable(); }
} // namespace bz

// gfx950 code object for the bls12_381 MSM kernels (see curve_tu.h / kernels.h).
// 14-limb products: pinning the accumulation order of the Montgomery columns (field/mont29.h) is
// worth 3 % of k_accumulate here (config 3: 14.74 -> 14.35 ms, A/B on one box,
// profiles/round2_ab_mont_mad.log); on the 9-limb curves it gains < 1 % there and slows the
// lone-wave k_reduce by 25-55 %, so they keep the compiler's order.
#define BZ_MONT29_MAD_MODE 1
#include "blitzar_amd/csrc/msm/curve_tu.h"

namespace bz {
BZ_ACCUMULATE_INSTANCE(extern, bls12_381_msm); // msm_bls12_381_accumulate.hip
const curve_vtable& bls12_381_vtable() { return curve_tu<bls12_381_msm>::vtable(); }
} // namespace bz

"""The lane-spread Weierstrass arithmetic of k_horner (curve/sw_wave.h) has no host build: its
algorithm -- limb layout (the engine's limbs plus one), the Montgomery product computed by the lanes
of a DPP row together (zero-filling row shifts, three-piece carries, the one carry that crosses from
the low half read off its two top lanes), the rounds of the doubling and of the complete addition,
the limb-wise multiple of p used for subtraction -- is restated in tools/models/sw_wave_model.py with
interval propagation of every limb bound.  This runs the model for the three base fields: values
against big-integer arithmetic (products, a doubling / addition chain, the special cases of the
complete formulas), bounds against the 32-bit / 64-bit limits of the instructions the kernel uses.
(The kernel itself is covered by every -m gpu parity test on bn254 / grumpkin / bls12-381: all their
MSM results pass through it.)"""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_model():
    spec = importlib.util.spec_from_file_location(
        "sw_wave_model", os.path.join(ROOT, "tools", "models", "sw_wave_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("name", ["bn254", "grumpkin", "bls12_381"])
def test_field_products_and_bounds(name):
    # carried results: every limb at most 2^LB + 2
    assert load_model().check_field_products(name) < 1.0001


@pytest.mark.parametrize("name", ["bn254", "grumpkin", "bls12_381"])
def test_doubling_addition_chain(name):
    assert load_model().check_point_chain(name) < 1.001


def test_constants_match_the_generated_header():
    """P::wave_ninv / wave_one / wave_bias of field/mont29_params.h are what the model computes"""
    import re
    m = load_model()
    text = open(os.path.join(ROOT, "blitzar_amd", "csrc", "field", "mont29_params.h")).read()
    structs = {"bn254": "bn254_fq29_params", "grumpkin": "grumpkin_fq29_params",
               "bls12_381": "bls12_381_fp28_params"}
    for name, struct in structs.items():
        f = m.FIELDS[name]
        body = text[text.index("struct " + struct):]
        body = body[:body.index("};")]

        def table(key):
            line = re.search(r"BZ_LIMB32_FN\(" + key + r", ([^)]*)\)", body).group(1)
            return [int(x.strip().rstrip("u"), 16) for x in line.split(",")]
        assert table("wave_ninv") == f.ninv_limbs
        assert table("wave_bias") == f.bias_limbs
        assert table("p") == f.p_limbs[:f.N]
        one = f.Rw % f.p
        assert table("wave_one") == [(one >> (f.LB * i)) & f.mask for i in range(f.NW)]

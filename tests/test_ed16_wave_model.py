"""The lane-spread edwards25519 arithmetic of k_horner (curve/ed16_wave.h) has no host build: its
algorithm -- limb layout, rotations with the 2^256 = 38 wrap, carry rounds, the formula rounds and
the limb-wise multiples of p used for subtraction -- is restated in tools/models/ed16_wave_model.py
with interval propagation of every limb bound.  This runs the model: values against big-integer
edwards25519 arithmetic, bounds against the 24-bit / 32-bit / 48-bit / 64-bit limits of the
instructions the kernel uses.  (The kernel itself is covered by every -m gpu parity test: all MSM
results pass through it.)"""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_model():
    spec = importlib.util.spec_from_file_location(
        "ed16_wave_model", os.path.join(ROOT, "tools", "models", "ed16_wave_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_field_products_and_bounds():
    limb0, others = load_model().check_field_products()
    assert limb0 < 16.01 and others < 16.001


def test_doubling_addition_chain():
    assert load_model().check_point_chain() < 16.01

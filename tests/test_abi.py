"""The drop-in boundary: libblitzar_amd.so loads without a GPU, exports every symbol that
include/*.h declares (and nothing else), and keeps the reference's struct layouts and error
convention (cbindings/blitzar_api.h, cbindings/libblitzar-export-map.ld:1-5,
cbindings/backend.cc:114-134, cbindings/get_generators.cc:40-48).  No compute calls here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np

from blitzar_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DECL = re.compile(r"\b((?:sxt|bzamd)_[a-z0-9_]+)\s*\(", re.M)


def declared_symbols(header):
    with open(os.path.join(ROOT, "include", header)) as fh:
        text = fh.read()
    # strip comments so that prose mentioning a function does not count as a declaration
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return sorted(set(DECL.findall(text)))


def exported_symbols():
    out = subprocess.run(["nm", "-D", "--defined-only", api.LIB_PATH], check=True,
                         capture_output=True, text=True).stdout
    return sorted({ln.split()[-1] for ln in out.splitlines() if ln.split()[1:2] == ["T"]})


def test_every_declared_symbol_is_exported():
    lib = api.load()
    declared = declared_symbols("blitzar_api.h") + declared_symbols("blitzar_amd.h")
    assert len(declared) >= 18 + 10
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"


def test_only_the_c_abi_is_exported():
    # the reference's version script exports sxt_* only; ours adds the bzamd_* extensions
    for name in exported_symbols():
        assert name.startswith(("sxt_", "bzamd_")), f"unexpected exported symbol {name}"


def test_hot_path_symbol_set_matches_reference_header():
    want = {
        "sxt_init", "sxt_curve25519_compute_pedersen_commitments",
        "sxt_curve25519_compute_pedersen_commitments_with_generators",
        "sxt_bls12_381_g1_compute_pedersen_commitments_with_generators",
        "sxt_bn254_g1_uncompressed_compute_pedersen_commitments_with_generators",
        "sxt_grumpkin_uncompressed_compute_pedersen_commitments_with_generators",
        "sxt_ristretto255_get_generators", "sxt_curve25519_get_one_commit",
        "sxt_multiexp_handle_new", "sxt_multiexp_handle_new_from_file",
        "sxt_multiexp_handle_write_to_file", "sxt_multiexp_handle_free",
        "sxt_fixed_multiexponentiation", "sxt_fixed_packed_multiexponentiation",
        "sxt_fixed_vlen_multiexponentiation",
        # link compatibility only (abort when called)
        "sxt_curve25519_prove_inner_product", "sxt_curve25519_verify_inner_product",
        "sxt_prove_sumcheck",
    }
    assert want <= set(exported_symbols())
    assert want == set(declared_symbols("blitzar_api.h"))


def test_struct_layouts():
    # SURVEY Appendix A
    assert ctypes.sizeof(api.sxt_config) == 16
    assert ctypes.sizeof(api.sxt_sequence_descriptor) == 32
    assert api.sxt_sequence_descriptor.n.offset == 8
    assert api.sxt_sequence_descriptor.data.offset == 16
    assert api.sxt_sequence_descriptor.is_signed.offset == 24
    assert api.CURVE_LAYOUT == {0: (160, 32, 160), 1: (104, 48, 144), 2: (72, 72, 96),
                                3: (72, 72, 96)}


def test_version_and_device_probe_do_not_need_a_gpu():
    lib = api.load()
    assert b"gfx950" in lib.bzamd_version()
    assert lib.bzamd_device_count() >= 0
    assert lib.bzamd_active_backend() in (0, api.SXT_CPU_BACKEND, api.SXT_GPU_BACKEND)


def test_init_error_convention(cpu_backend):
    lib = cpu_backend.load()
    assert lib.bzamd_active_backend() == api.SXT_CPU_BACKEND
    # num > 0 with a null pointer -> 1; num == 0 -> 0 (cbindings/get_generators.cc:40-48)
    assert lib.sxt_ristretto255_get_generators(None, 3, 0) == 1
    assert lib.sxt_ristretto255_get_generators(None, 0, 0) == 0
    # num_sequences == 0 returns before touching anything (cbindings/pedersen.cc:77-78)
    lib.sxt_curve25519_compute_pedersen_commitments(None, 0, None, 0)
    lib.sxt_bn254_g1_uncompressed_compute_pedersen_commitments_with_generators(None, 0, None, None)


def _run_child(code, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=120)


def test_unknown_backend_returns_1_and_misuse_aborts():
    # unknown backend id -> 1 (cbindings/backend.cc:127-131)
    r = _run_child("from blitzar_amd import api; import sys; sys.exit(10 + api.init(7, 0))")
    assert r.returncode == 11, r.stderr
    # compute before sxt_init aborts the process (SXT_RELEASE_ASSERT -> std::abort)
    r = _run_child(
        "import numpy as np\nfrom blitzar_amd import api\n"
        "api.compute_pedersen_commitments(0, [(np.ones(4, np.uint32), False)])\n")
    assert r.returncode < 0 and "not initialised" in r.stderr
    # re-initialisation aborts (cbindings/backend.cc:116-117)
    r = _run_child("from blitzar_amd import api\napi.init(1, 0)\napi.init(1, 0)\n")
    assert r.returncode < 0 and "reinitialize" in r.stderr
    # element_nbytes outside [1, 32] aborts (cbindings/pedersen.cc:46-55)
    r = _run_child(
        "import ctypes, numpy as np\nfrom blitzar_amd import api\napi.init(1, 0)\n"
        "d = (api.sxt_sequence_descriptor * 1)()\nbuf = np.zeros(64, np.uint8)\n"
        "d[0] = api.sxt_sequence_descriptor(33, 1, buf.ctypes.data, 0)\n"
        "out = np.zeros(32, np.uint8)\n"
        "api.load().sxt_curve25519_compute_pedersen_commitments(out.ctypes.data, 1, d, 0)\n")
    assert r.returncode < 0 and "element_nbytes" in r.stderr


def test_environment_override_of_backend():
    # BLITZAR_BACKEND=cpu overrides config.backend (cbindings/backend.cc:72-89)
    r = _run_child(
        "from blitzar_amd import api\nassert api.init(2, 0) == 0\n"
        "print(api.load().bzamd_active_backend())", {"BLITZAR_BACKEND": "cpu"})
    assert r.returncode == 0 and r.stdout.strip() == "1", r.stderr


def test_sumcheck_rejects_null_arguments_loudly():
    r = _run_child("from blitzar_amd import api\nlib = api.load()\napi.init(1, 0)\n"
                   "lib.sxt_prove_sumcheck(None, None, 0, None, None, None)\n")
    assert r.returncode < 0 and "sxt_prove_sumcheck" in r.stderr


def test_gpu_backend_without_gpu_fails_loudly():
    lib = api.load()
    if lib.bzamd_device_count() > 0:
        return
    r = _run_child("from blitzar_amd import api\napi.init(2, 0)\n")
    assert r.returncode < 0 and "no supported GPUs" in r.stderr


def test_numpy_views_are_little_endian_host():
    assert np.dtype(np.uint32).byteorder in ("=", "<") and sys.byteorder == "little"


def test_build_recipe_is_consistent():
    """blitzar_amd/build.py: every per-unit flag set names a source that is built, every source
    exists, and every curve's accumulation loop has its translation unit (the units that launch
    k_accumulate only declare the instantiation: a missing one would be a link error on the box)"""
    import os
    from blitzar_amd import build
    for src in build.SOURCES:
        assert os.path.exists(os.path.join(build.CSRC, src)), src
    assert set(build.TU_FLAGS) <= set(build.SOURCES)
    accumulate_units = [s for s in build.SOURCES if s.endswith("_accumulate.hip")]
    instances = ""
    for s in accumulate_units:
        with open(os.path.join(build.CSRC, s)) as fh:
            instances += fh.read()
    for curve in ("ed25519_msm", "ed25519_niels_msm", "bls12_381_msm", "bn254_msm", "grumpkin_msm"):
        assert f"BZ_ACCUMULATE_INSTANCE(, {curve});" in instances, curve
        declared = 0
        for s in build.SOURCES:
            if s.startswith("msm/msm_") and not s.endswith("_accumulate.hip"):
                with open(os.path.join(build.CSRC, s)) as fh:
                    declared += fh.read().count(f"BZ_ACCUMULATE_INSTANCE(extern, {curve});")
        assert declared == 1, curve

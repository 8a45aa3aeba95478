"""Build libblitzar_amd.so (gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m blitzar_amd.build            # incremental
    python -m blitzar_amd.build --force

One object per translation unit, compiled in parallel; objects and the shared library live under
blitzar_amd/lib/ (git-ignored, shipped to the GPU box by gpurun).  hipcc cross-compiles for
gfx950 without a GPU present.
"""
import concurrent.futures
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "blitzar_amd", "csrc")
OUT = os.path.join(ROOT, "blitzar_amd", "lib")
LIB = os.path.join(OUT, "libblitzar_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SOURCES = [
    "msm/msm_curve25519.hip",
    "msm/msm_curve25519_accumulate.hip",
    "msm/msm_curve25519_niels_accumulate.hip",
    "msm/msm_bls12_381.hip",
    "msm/msm_bls12_381_accumulate.hip",
    "msm/msm_bn254_accumulate.hip",
    "msm/msm_grumpkin_accumulate.hip",
    "msm/msm_bn254.hip",
    "msm/msm_grumpkin.hip",
    "msm/context.hip",
    "generators/builtin.hip",
    "proof/inner_product.hip",
    "proof/sumcheck.hip",
    "api/capi.hip",
]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-I" + ROOT,
         "-Wno-unused-result"]
# Instruction scheduling per translation unit (measured on MI355X, A/B on one box,
# profiles/round3_ab_sched_strategy.log).  hipcc's default strategy schedules for occupancy and
# pads every dependent v_mad_u64_u32 pair it could not separate with an s_nop; "max-ilp" fills
# those slots with independent instructions at the price of registers:
#   * the Weierstrass accumulation loops lose 1200 of their 1400 s_nop per addition and still fit
#     their wave budget (bn254 160, grumpkin 168, bls12-381 203 VGPRs): k_accumulate -3.6 % / -3.1 %
#     / -1.8 %;
#   * the curve25519 loops spill under max-ilp at three waves per SIMD (the Z = 1 form: 0.583 ->
#     0.602 ms); they take the default strategy with a launch bound of two waves instead, which
#     drops 145 of 429 s_nop at the same 140 VGPRs (still three waves resident): -1.6 %; the
#     "iterative-ilp" strategy fits the Z = 1 loop into three waves without a spill (159 VGPRs, 25
#     s_nop left: 0.578 -> 0.567 ms, a sequence on resident generators 0.869 -> 0.855 ms per call)
#     but spills 92 bytes in the caller-generators loop, which then loses in a sequence (0.966 ->
#     0.975) what it gains alone (0.623 -> 0.605);
#   * the other curve25519 kernels are latency chains on few wavefronts (k_reduce, k_horner): max-ilp
#     0.198 -> 0.183 and 0.189 -> 0.183 ms, a lone config-2 call 1.187 -> 1.155 ms;
#   * the other Weierstrass kernels keep the default (grumpkin's k_reduce: +23 % under max-ilp).
MAX_ILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
TU_FLAGS = {
    "msm/msm_curve25519.hip": MAX_ILP,
    "msm/msm_curve25519_accumulate.hip": ["-DBZ_ACCUMULATE_WAVES_OVERRIDE=2"],
    "msm/msm_curve25519_niels_accumulate.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
    "msm/msm_bls12_381_accumulate.hip": MAX_ILP,
    "msm/msm_bn254_accumulate.hip": MAX_ILP,
    "msm/msm_grumpkin_accumulate.hip": MAX_ILP,
}
# A/B variants (tools/prof/ab_env.sh): BZ_VARIANT=<tag> builds blitzar_amd/lib/variants/<tag>/ with
# BZ_EXTRA_FLAGS (e.g. "-DBZ_F29_MAD_MODE=0") appended; the default library is untouched
VARIANT = os.environ.get("BZ_VARIANT")
if VARIANT:
    OUT = os.path.join(OUT, "variants", VARIANT)
    LIB = os.path.join(OUT, "libblitzar_amd.so")
    FLAGS = FLAGS + os.environ.get("BZ_EXTRA_FLAGS", "").split()


def _headers():
    hs = []
    for d, _, fs in os.walk(CSRC):
        hs += [os.path.join(d, f) for f in fs if f.endswith(".h")]
    hs += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    return hs


def _compile(src, newest_header, force):
    obj = os.path.join(OUT, src.replace("/", "_").replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(path), newest_header)):
        return obj, False
    t0 = time.time()
    subprocess.run([HIPCC, *FLAGS, *TU_FLAGS.get(src, []), "-c", path, "-o", obj], check=True)
    print(f"[blitzar_amd] compiled {src} in {time.time() - t0:.0f}s", flush=True)
    return obj, True


def build(force=False, verbose=True, only=None):
    """`only`: development shortcut -- recompile just the sources whose name contains one of the
    given substrings and relink against the existing objects of the others (valid while the
    cross-TU interface, msm/dispatch.h, is unchanged)."""
    os.makedirs(OUT, exist_ok=True)
    newest = max(os.path.getmtime(h) for h in _headers() + [os.path.abspath(__file__)])

    def job(src):
        if only is not None and not any(tag in src for tag in only):
            name = src.replace("/", "_").replace(".hip", ".o")
            obj = os.path.join(OUT, name)
            if VARIANT and not os.path.exists(obj):
                # a variant that only touches some translation units links the default objects
                # of the others
                obj = os.path.join(ROOT, "blitzar_amd", "lib", name)
            assert os.path.exists(obj), f"--only needs an existing {obj}"
            return obj, False
        return _compile(src, newest, force or only is not None)

    with concurrent.futures.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        results = list(ex.map(job, SOURCES))
    objs = [o for o, _ in results]
    if force or any(c for _, c in results) or not os.path.exists(LIB):
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs,
                        "-Wl,-rpath,/opt/rocm/lib",
                        "-Wl,--version-script=" + os.path.join(CSRC, "api", "export.map")],
                       check=True)
        if verbose:
            print(f"[blitzar_amd] built {LIB}")
    elif verbose:
        print(f"[blitzar_amd] up to date: {LIB}")
    return LIB


if __name__ == "__main__":
    only = None
    for a in sys.argv[1:]:
        if a.startswith("--only="):
            only = a[len("--only="):].split(",")
    build(force="--force" in sys.argv, only=only)

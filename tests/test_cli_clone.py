"""tools/multi_commitment: the command-line clone of the reference's benchmark/multi_commitment
(SURVEY 8(f) rank 3), driven like benchmark/scripts/run_benchmarks.py drives the original, on the
host backend; its commitments must equal the reference oracle's on the same mt19937{0} byte
stream (benchmark.m.cc:136-165)."""
import os
import re
import subprocess

import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import baseline_workloads as wl  # noqa: E402
SRC = os.path.join(ROOT, "tools", "multi_commitment", "benchmark.cc")
EXE = os.path.join(ROOT, "tools", "multi_commitment", "_build", "multi_commitment")


def build():
    lib_dir = os.path.join(ROOT, "blitzar_amd", "lib")
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), SRC,
                    "-L" + lib_dir, "-lblitzar_amd", "-Wl,-rpath," + lib_dir,
                    "-Wl,-rpath,/opt/rocm/lib", "-o", EXE], check=True)


def mt19937_bytes(count, boolean):
    """std::mt19937{0} through libstdc++'s uniform_int_distribution<uint8_t>"""
    return wl.mt19937_bytes(count, 0, boolean)


def run_multi_commitment(oracle, backend, n, commitments, nbytes):
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(SRC):
        build()
    out = subprocess.run([EXE, backend, str(n), "2", str(commitments), str(nbytes), "1"],
                         capture_output=True, text=True, timeout=300, check=True).stdout
    assert f"backend : {backend}" in out
    assert "throughput (exponentiations / s) :" in out and "compute duration (s) :" in out
    assert f"num_exponentations : {n * commitments}" in out
    got = [bytes.fromhex(h) for h in re.findall(r"commitment \d+ = 0x([0-9a-f]{64})", out)]
    width = max(nbytes, 1)
    table = mt19937_bytes(n * commitments * width, nbytes == 0).reshape(commitments, n, width)
    want = oracle.commit(0, [(table[c], False) for c in range(commitments)],
                         oracle.ristretto_generators(n))
    assert [bytes(w) for w in want] == got


@pytest.mark.parametrize("n,commitments,nbytes", [(300, 2, 32), (1000, 3, 1), (257, 1, 0)])
def test_multi_commitment_cli(oracle, n, commitments, nbytes):
    run_multi_commitment(oracle, "cpu", n, commitments, nbytes)


@pytest.mark.gpu
@pytest.mark.parametrize("n,commitments,nbytes", [(300, 2, 32), (20000, 3, 1), (257, 1, 0)])
def test_multi_commitment_cli_gpu(oracle, n, commitments, nbytes):
    """the same clone with the `gpu` argument: the HIP engine behind the drop-in sxt_* entry points
    with host buffers, in a process that never loaded torch"""
    run_multi_commitment(oracle, "gpu", n, commitments, nbytes)


#--------------------------------------------------------------------------------------------------
# tools/multi_exp: clones of benchmark/multi_exp_pip and benchmark/multi_exp_triangle
#--------------------------------------------------------------------------------------------------
EXP_SRC = os.path.join(ROOT, "tools", "multi_exp", "benchmark.cc")
EXP_EXE = {False: os.path.join(ROOT, "tools", "multi_exp", "_build", "multi_exp_pip"),
           True: os.path.join(ROOT, "tools", "multi_exp", "_build", "multi_exp_triangle")}
FIELD_P = {2: 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47,
           3: 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001}


def build_multi_exp(triangle):
    exe = EXP_EXE[triangle]
    if os.path.exists(exe) and os.path.getmtime(exe) >= os.path.getmtime(EXP_SRC):
        return exe
    lib_dir = os.path.join(ROOT, "blitzar_amd", "lib")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), EXP_SRC]
                   + (["-DBZ_TRIANGLE"] if triangle else [])
                   + ["-L" + lib_dir, "-lblitzar_amd", "-Wl,-rpath," + lib_dir,
                      "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def reference_generators(oracle, cid, n):
    """the generators the reference benchmarks use (multi_exp_pip/benchmark.m.cc:84-112), from the
    reference's own code"""
    if cid == 0:
        return oracle.ristretto_generators(n)
    return np.stack([oracle.random_affine(cid, i + 1, i + 2) for i in range(n)])


def run_multi_exp(oracle, curve, cid, n, outputs, nbytes, triangle, backend="cpu"):
    exe = build_multi_exp(triangle)
    env = dict(os.environ, BLITZAR_BACKEND=backend)
    out = subprocess.run([exe, curve, str(n), "1", str(outputs), str(nbytes), "1"], env=env,
                         capture_output=True, text=True, timeout=600, check=True).stdout
    assert f"running {curve} benchmark..." in out and "compute duration (s): " in out
    # exponents are drawn output-major, stored row-major (fill_exponents, benchmark.m.cc:117-131)
    table = mt19937_bytes(outputs * n * nbytes, False).reshape(outputs, n, nbytes)
    lengths = [n] * outputs
    if triangle:
        counter = n - outputs if n > outputs else 0
        lengths = [min(counter + k + 1, n) for k in range(outputs)]
    want = oracle.commit(cid, [(table[o][:lengths[o]], False) for o in range(outputs)],
                         reference_generators(oracle, cid, n))
    lines = re.findall(r"^(\d+): \{(.*)\}$", out, flags=re.M)
    assert [int(k) for k, _ in lines] == list(range(outputs))
    for (_, body), w in zip(lines, want):
        if cid in (0, 1):
            assert [int(v) for v in body.rstrip(",").split(",")] == list(w)
        else:
            x, y = (int(v.split("_")[0], 16) for v in body.split(", "))
            p = FIELD_P[cid]
            r_inv = pow(1 << 256, -1, p)
            assert x == int.from_bytes(bytes(w[:32]), "little") * r_inv % p
            assert y == int.from_bytes(bytes(w[32:64]), "little") * r_inv % p
            assert body.endswith("_f25" if cid == 2 else "_fgk")


@pytest.mark.parametrize("curve,cid", [("curve25519", 0), ("bls12_381", 1), ("bn254", 2),
                                       ("grumpkin", 3)])
def test_multi_exp_pip_cli(oracle, curve, cid):
    run_multi_exp(oracle, curve, cid, n=37, outputs=3, nbytes=5, triangle=False)


@pytest.mark.parametrize("curve,cid,n,outputs", [("curve25519", 0, 40, 6), ("bn254", 2, 5, 9)])
def test_multi_exp_triangle_cli(oracle, curve, cid, n, outputs):
    run_multi_exp(oracle, curve, cid, n=n, outputs=outputs, nbytes=2, triangle=True)


@pytest.mark.gpu
@pytest.mark.parametrize("curve,cid", [("curve25519", 0), ("bls12_381", 1), ("bn254", 2),
                                       ("grumpkin", 3)])
def test_multi_exp_pip_cli_gpu(oracle, curve, cid):
    """BLITZAR_BACKEND=gpu: fixed-base handles on the HIP engine (the reference benchmark's gpu
    mode, benchmark/multi_exp_pip/benchmark.m.cc)"""
    run_multi_exp(oracle, curve, cid, n=37, outputs=3, nbytes=5, triangle=False, backend="gpu")


@pytest.mark.gpu
@pytest.mark.parametrize("curve,cid,n,outputs", [("curve25519", 0, 40, 6), ("bn254", 2, 5, 9)])
def test_multi_exp_triangle_cli_gpu(oracle, curve, cid, n, outputs):
    run_multi_exp(oracle, curve, cid, n=n, outputs=outputs, nbytes=2, triangle=True, backend="gpu")

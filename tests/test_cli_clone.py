"""tools/multi_commitment: the command-line clone of the reference's benchmark/multi_commitment
(SURVEY 8(f) rank 3), driven like benchmark/scripts/run_benchmarks.py drives the original, on the
host backend; its commitments must equal the reference oracle's on the same mt19937{0} byte
stream (benchmark.m.cc:136-165)."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "multi_commitment", "benchmark.cc")
EXE = os.path.join(ROOT, "tools", "multi_commitment", "_build", "multi_commitment")


def build():
    lib_dir = os.path.join(ROOT, "blitzar_amd", "lib")
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), SRC,
                    "-L" + lib_dir, "-lblitzar_amd", "-Wl,-rpath," + lib_dir,
                    "-Wl,-rpath,/opt/rocm/lib", "-o", EXE], check=True)


def mt19937_bytes(count, boolean):
    """std::mt19937{0} through libstdc++'s uniform_int_distribution<uint8_t> (Lemire's method on
    a 32-bit engine: the top bits of every draw)"""
    raw = np.random.RandomState(0).randint(0, 2**32, size=count, dtype=np.uint64)
    return ((raw * (2 if boolean else 256)) >> 32).astype(np.uint8)


@pytest.mark.parametrize("n,commitments,nbytes", [(300, 2, 32), (1000, 3, 1), (257, 1, 0)])
def test_multi_commitment_cli(oracle, n, commitments, nbytes):
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(SRC):
        build()
    out = subprocess.run([EXE, "cpu", str(n), "2", str(commitments), str(nbytes), "1"],
                         capture_output=True, text=True, timeout=300, check=True).stdout
    assert "throughput (exponentiations / s) :" in out and "compute duration (s) :" in out
    assert f"num_exponentations : {n * commitments}" in out
    got = [bytes.fromhex(h) for h in re.findall(r"commitment \d+ = 0x([0-9a-f]{64})", out)]
    width = max(nbytes, 1)
    table = mt19937_bytes(n * commitments * width, nbytes == 0).reshape(commitments, n, width)
    want = oracle.commit(0, [(table[c], False) for c in range(commitments)],
                         oracle.ristretto_generators(n))
    assert [bytes(w) for w in want] == got

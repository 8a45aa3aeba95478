"""SXT_CPU_BACKEND of the product through the C ABI (BASELINE configs[0]: plumbing, no GPU)
against the reference oracle and the committed golden fixtures.  The host backend shares the
field / curve / recoding headers with the gfx950 kernels, so this suite is also the CPU-side
check of that arithmetic."""
import os

import numpy as np
import pytest

from tests import util
from tests.golden import make_golden

GOLDEN = np.load(os.path.join(os.path.dirname(__file__), "golden", "msm_golden.npz"))


def test_rust_golden_vectors(cpu_backend):
    cols = [(row.copy(), False) for row in GOLDEN["rust_kat_data"]]
    out = cpu_backend.compute_pedersen_commitments(0, cols)
    assert np.array_equal(out, GOLDEN["rust_kat"])
    # the same through the _with_generators entry point (null generators = built-in, offset 0)
    lib = cpu_backend.load()
    descs, keep = cpu_backend.make_descriptors(cols)
    out2 = np.zeros((3, 32), np.uint8)
    lib.sxt_curve25519_compute_pedersen_commitments_with_generators(out2.ctypes.data, 3, descs, None)
    assert np.array_equal(out2, GOLDEN["rust_kat"])


def test_builtin_generators_and_one_commit_are_limb_exact(cpu_backend):
    # init cached 10 generators: inside, straddling and beyond the cache
    assert np.array_equal(cpu_backend.get_generators(8, 0), GOLDEN["ristretto_generators_0_8"])
    assert np.array_equal(cpu_backend.get_generators(4, 1000),
                          GOLDEN["ristretto_generators_1000_4"])
    both = cpu_backend.get_generators(12, 0)
    assert np.array_equal(both[:8], GOLDEN["ristretto_generators_0_8"])
    got = np.stack([cpu_backend.get_one_commit(k) for k in (0, 1, 5, 33)])
    assert np.array_equal(got, GOLDEN["one_commit_0_1_5_33"])


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_golden_commitments(cpu_backend, cid):
    cols = make_golden.golden_columns(1000 + cid, 48)
    got = cpu_backend.compute_pedersen_commitments(cid, cols,
                                                   generators=GOLDEN[f"curve{cid}_generators"])
    assert np.array_equal(got, GOLDEN[f"curve{cid}_commitments"])


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_random_sweep_matches_oracle(cpu_backend, oracle, cid):
    """the randomized part of the reference exerciser (sxt/multiexp/test/multiexponentiation.cc:
    290-451): 1-10 sequences, length 0-100, 1-32 byte scalars, alternating signedness"""
    rng = np.random.default_rng(7000 + cid)
    gens = util.generators_for(cid, 100)
    for _ in range(4):
        cols = []
        for s in range(int(rng.integers(1, 11))):
            signed = bool(s % 2)
            nb = int(rng.choice([1, 2, 4, 8, 16])) if signed else int(rng.integers(1, 33))
            cols.append((rng.integers(0, 256, (int(rng.integers(0, 101)), nb), dtype=np.uint8),
                         signed))
        got = cpu_backend.compute_pedersen_commitments(cid, cols,
                                                       generators=util.api_generators(cid, gens))
        assert np.array_equal(got, oracle.commit(cid, cols, gens))


def test_config1_column_shape(cpu_backend, oracle):
    """BASELINE configs[0] at a size the oracle finishes in seconds: one 32-byte column,
    built-in generators, cpu backend"""
    n = 1 << 11
    col = [(np.random.default_rng(0).integers(0, 256, (n, 32), dtype=np.uint8), False)]
    got = cpu_backend.compute_pedersen_commitments(0, col)
    assert np.array_equal(got, oracle.commit(0, col, oracle.ristretto_generators(n)))


def test_offset_generators(cpu_backend, oracle):
    rng = np.random.default_rng(5)
    cols = [(rng.integers(0, 256, (20, 8), dtype=np.uint8), False)]
    for off in (0, 3, 9, 10, 500):
        got = cpu_backend.compute_pedersen_commitments(0, cols, offset_generators=off)
        assert np.array_equal(got, oracle.commit(0, cols, oracle.ristretto_generators(20, off)))


def test_homomorphism(cpu_backend):
    """cbindings/pedersen.t.cc:287-316: commit(a) + commit(b) == commit(a + b)"""
    from tests import hooks
    a = np.array([1, 2, 3, 4 << 60], dtype=np.uint64)
    b = np.array([7, 0, 2**63, 5], dtype=np.uint64)
    c = (a.astype(object) + b.astype(object))
    wide = np.zeros((4, 16), np.uint8)
    for i, v in enumerate(c):
        wide[i] = np.frombuffer(int(v).to_bytes(16, "little"), np.uint8)
    out = cpu_backend.compute_pedersen_commitments(0, [(a, False), (b, False), (wide, False)])
    pa, pb = hooks.ristretto_decode(out[0]), hooks.ristretto_decode(out[1])
    assert np.array_equal(hooks.ristretto_encode(hooks.ed_add(pa, pb)), out[2])


def test_gpu_only_queries_answer_on_the_host_backend(cpu_backend):
    """the device queries of include/blitzar_amd.h have defined answers without a GPU backend"""
    lib = cpu_backend.load()
    assert lib.bzamd_slow_instruction_fetch() == -1
    assert lib.bzamd_concurrent_calls_high_water() == 0

"""Pin the oracle before trusting it (CPU only).

oracle/_ref is the reference's own CPU backend built from /root/reference; oracle/fixed_base.py
restates the one part that cannot be compiled here.  Checks:
  * the three byte-level KATs of rust/tests/src/main.rs:22-47,
  * homomorphism (third KAT = sum of the first two, as the Rust test asserts, :77-79),
  * the committed golden fixtures are exactly what the oracle produces today,
  * the fixed-base restatement agrees with the reference's variable-base backend on the same
    scalars unpacked to columns, and with the known answers of cbindings/fixed_pedersen.t.cc,
  * the reference's OWN fixed-base path compiled in place (oracle/ref/ref_fixed_base.cc: its
    accessor, mtxpp2::multiexponentiate, and -- on host stand-ins for the CUDA runtime -- its GPU
    control flow async_multiexponentiate) gives the same answers, tables and files.
"""
import hashlib
import os

import numpy as np
import pytest

from tests import util
from tests.golden import make_golden

GOLDEN = np.load(os.path.join(os.path.dirname(__file__), "golden", "msm_golden.npz"))


def test_rust_kats(oracle):
    data = GOLDEN["rust_kat_data"]
    cols = [(row.copy(), False) for row in data]
    got = oracle.commit(0, cols, oracle.ristretto_generators(4))
    assert got.tolist() == make_golden.RUST_KAT
    assert np.array_equal(got, GOLDEN["rust_kat"])


def test_golden_fixtures_match_oracle(oracle):
    assert np.array_equal(oracle.ristretto_generators(8, 0), GOLDEN["ristretto_generators_0_8"])
    assert np.array_equal(oracle.ristretto_generators(4, 1000),
                          GOLDEN["ristretto_generators_1000_4"])
    want = np.stack([oracle.one_commit(k) for k in (0, 1, 5, 33)])
    assert np.array_equal(want, GOLDEN["one_commit_0_1_5_33"])
    n = 48
    for cid in (0, 1, 2, 3):
        gens = util.generators_for(cid, n)
        assert np.array_equal(util.api_generators(cid, gens), GOLDEN[f"curve{cid}_generators"])
        cols = make_golden.golden_columns(1000 + cid, n)
        assert np.array_equal(oracle.commit(cid, cols, gens), GOLDEN[f"curve{cid}_commitments"])


def test_one_commit_is_prefix_sum_of_generators(oracle):
    g = oracle.ristretto_generators(6)
    acc = oracle.one_commit(0)
    for i in range(6):
        assert np.array_equal(oracle.ristretto_compress(acc),
                              oracle.ristretto_compress(oracle.one_commit(i)))
        acc = oracle.add_projective(0, acc, g[i])


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_fixed_base_restatement_matches_variable_base_reference(oracle, cid):
    from oracle import fixed_base
    rng = np.random.default_rng(40 + cid)
    m = 11  # not a multiple of the window width -> identity padding
    gens = util.generators_for(cid, m)
    proj = gens if cid == 0 else oracle.affine_to_projective(cid, gens)
    for w, bit_table in ((4, [3, 1, 8, 13, 64, 256]), (3, [1, 1, 7]), (16, [9])):
        table = fixed_base.PartitionTable(cid, proj, w)
        row = (sum(bit_table) + 7) // 8
        scalars = rng.integers(0, 256, (m, row), dtype=np.uint8)
        res = fixed_base.multiexponentiate(table, bit_table, m, scalars)
        got = np.stack([oracle.canonical(cid, r).view(np.uint8).reshape(-1) for r in res])
        want = oracle.commit(cid, fixed_base.unpack_columns(bit_table, m, scalars), gens)
        want = want if cid in (0, 1) else want[:, :]
        assert np.array_equal(got[:, :want.shape[1]], want)
    # variable length: ascending lengths, a zero-length output yields the identity
    bit_table, lengths = [5, 2, 16], [0, 4, 11]
    table = fixed_base.PartitionTable(cid, proj, 4)
    row = (sum(bit_table) + 7) // 8
    scalars = rng.integers(0, 256, (m, row), dtype=np.uint8)
    res = fixed_base.multiexponentiate(table, bit_table, m, scalars, lengths)
    got = np.stack([oracle.canonical(cid, r).view(np.uint8).reshape(-1) for r in res])
    want = oracle.commit(cid, fixed_base.unpack_columns(bit_table, m, scalars, lengths), gens)
    assert np.array_equal(got[:, :want.shape[1]], want)


def test_fixed_base_known_answers(oracle):
    """cbindings/fixed_pedersen.t.cc:51-200, answers as group expressions"""
    from oracle import fixed_base
    g = oracle.ristretto_generators(3, 7)  # any three distinct points

    def lin(coeffs):
        cols = [(np.array(coeffs, dtype=np.uint64), False)]
        return oracle.commit(0, cols, g)[0]

    def canon(p):
        return oracle.ristretto_compress(p)

    t2 = fixed_base.PartitionTable(0, g[:2], 16)
    # plain, element_num_bytes = 2, scalars {1,0, 0,2}: out = g0 + 512 g1
    r = fixed_base.multiexponentiate_bytes(t2, 2, 1, 2, np.array([1, 0, 0, 2], np.uint8))
    assert np.array_equal(canon(r[0]), lin([1, 512]))
    # packed {0b1010, 0b0101} with bit table {3, 1}: {2 g0 + 5 g1, g0}
    r = fixed_base.multiexponentiate(t2, [3, 1], 2, np.array([0b1010, 0b0101], np.uint8))
    assert np.array_equal(canon(r[0]), lin([2, 5]))
    assert np.array_equal(canon(r[1]), lin([1, 0]))
    # vlen {0b1011, 0b1101}, bits {3, 1}, lengths {1, 2}: {3 g0, g0 + g1}
    r = fixed_base.multiexponentiate(t2, [3, 1], 2, np.array([0b1011, 0b1101], np.uint8), [1, 2])
    assert np.array_equal(canon(r[0]), lin([3, 0]))
    assert np.array_equal(canon(r[1]), lin([1, 1]))
    # three generators (identity padding), packed {1,1,1} bits {8}: g0 + g1 + g2
    t3 = fixed_base.PartitionTable(0, g, 16)
    r = fixed_base.multiexponentiate(t3, [8], 3, np.array([1, 1, 1], np.uint8))
    assert np.array_equal(canon(r[0]), lin([1, 1, 1]))


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_golden_fixed_base_fixture(oracle, cid):
    from oracle import fixed_base
    proj = GOLDEN[f"curve{cid}_fixed_projective_generators"]
    table = fixed_base.PartitionTable(cid, proj, 4)
    digest = hashlib.sha256(table.file_bytes()).digest()
    assert digest == GOLDEN[f"curve{cid}_table_w4_sha256"].tobytes()
    res = fixed_base.multiexponentiate(table, GOLDEN["fixed_bit_table"], proj.shape[0],
                                       GOLDEN[f"curve{cid}_fixed_scalars"])
    got = np.stack([oracle.canonical(cid, r).view(np.uint8).reshape(-1) for r in res])
    assert np.array_equal(got, GOLDEN[f"curve{cid}_fixed_canonical"])


#--------------------------------------------------------------------------------------------------
# the reference's own fixed-base path, compiled in place (round 4)
#--------------------------------------------------------------------------------------------------
def _canon_all(oracle, cid, res):
    return np.stack([oracle.canonical(cid, r).view(np.uint8).reshape(-1) for r in res])


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_compiled_fixed_base_matches_restatement_and_variable_base(oracle, cid):
    from oracle import fixed_base
    rng = np.random.default_rng(7400 + cid)
    m = 43  # not a multiple of any window width used below
    gens = util.generators_for(cid, m)
    proj = gens if cid == 0 else oracle.affine_to_projective(cid, gens)
    for w, bit_table in ((4, [3, 1, 8, 13, 64, 256]), (3, [1, 1, 7]), (8, [9, 32, 8])):
        h = oracle.FixedHandle(cid, proj, w)
        table = fixed_base.PartitionTable(cid, proj, w)
        row = (sum(bit_table) + 7) // 8
        scalars = rng.integers(0, 256, (m, row), dtype=np.uint8)
        want = oracle.commit(cid, fixed_base.unpack_columns(bit_table, m, scalars), gens)
        restated = _canon_all(oracle, cid, fixed_base.multiexponentiate(table, bit_table, m, scalars))
        for gpu_flow in (False, True):
            got = _canon_all(oracle, cid, h.packed_multiexponentiation(bit_table, m, scalars,
                                                                        gpu_flow=gpu_flow))
            assert np.array_equal(got, restated)
            assert np.array_equal(got[:, :want.shape[1]], want)
        h.close()
    # byte-aligned outputs
    h = oracle.FixedHandle(cid, proj, 4)
    table = fixed_base.PartitionTable(cid, proj, 4)
    scalars = rng.integers(0, 256, (m, 3 * 5), dtype=np.uint8)
    restated = _canon_all(oracle, cid, fixed_base.multiexponentiate_bytes(table, 3, 5, m, scalars))
    for gpu_flow in (False, True):
        got = _canon_all(oracle, cid, h.multiexponentiation(3, 5, m, scalars, gpu_flow=gpu_flow))
        assert np.array_equal(got, restated)
    # ascending lengths; no zero-length output here, so the host loop is sound too
    bit_table, lengths = [5, 2, 16, 40], [2, 4, 11, 43]
    row = (sum(bit_table) + 7) // 8
    scalars = rng.integers(0, 256, (m, row), dtype=np.uint8)
    restated = _canon_all(oracle, cid,
                          fixed_base.multiexponentiate(table, bit_table, m, scalars, lengths))
    for gpu_flow in (False, True):
        got = _canon_all(oracle, cid, h.vlen_multiexponentiation(bit_table, lengths, scalars,
                                                                  gpu_flow=gpu_flow))
        assert np.array_equal(got, restated)
    h.close()


@pytest.mark.parametrize("cid", [0, 2])
def test_zero_length_outputs_follow_the_reference_gpu_flow(oracle, cid):
    """the reference's host vlen loop mutates its product index inside the loop
    (variable_length_partition_product.h:145-148) and goes wrong once an output has length zero;
    its GPU lambda (:101-109) does not.  The restatement -- and the product -- follow the GPU
    lambda, which is checked here through the reference's own async_multiexponentiate."""
    from oracle import fixed_base
    rng = np.random.default_rng(7500 + cid)
    m = 37
    gens = util.generators_for(cid, m)
    proj = gens if cid == 0 else oracle.affine_to_projective(cid, gens)
    h = oracle.FixedHandle(cid, proj, 4)
    table = fixed_base.PartitionTable(cid, proj, 4)
    bit_table, lengths = [4, 12, 1, 64], [0, 5, 5, 37]
    scalars = rng.integers(0, 256, (m, (sum(bit_table) + 7) // 8), dtype=np.uint8)
    restated = _canon_all(oracle, cid,
                          fixed_base.multiexponentiate(table, bit_table, m, scalars, lengths))
    got = _canon_all(oracle, cid, h.vlen_multiexponentiation(bit_table, lengths, scalars,
                                                              gpu_flow=True))
    assert np.array_equal(got, restated)
    want = oracle.commit(cid, fixed_base.unpack_columns(bit_table, m, scalars, lengths), gens)
    assert np.array_equal(got[:, :want.shape[1]], want)
    h.close()


def test_compiled_fixed_base_known_answers(oracle):
    """cbindings/fixed_pedersen.t.cc:51-200 through the reference's own accessor and
    multiexponentiate (window width 16, its default)"""
    g = oracle.ristretto_generators(3, 7)

    def lin(coeffs):
        return oracle.commit(0, [(np.array(coeffs, dtype=np.uint64), False)], g)[0]

    def canon(p):
        return oracle.ristretto_compress(p)

    h2 = oracle.FixedHandle(0, g[:2], 16)
    for gpu_flow in (False, True):
        r = h2.multiexponentiation(2, 1, 2, np.array([1, 0, 0, 2], np.uint8), gpu_flow=gpu_flow)
        assert np.array_equal(canon(r[0]), lin([1, 512]))
        r = h2.packed_multiexponentiation([3, 1], 2, np.array([0b1010, 0b0101], np.uint8),
                                          gpu_flow=gpu_flow)
        assert np.array_equal(canon(r[0]), lin([2, 5]))
        assert np.array_equal(canon(r[1]), lin([1, 0]))
        r = h2.vlen_multiexponentiation([3, 1], [1, 2], np.array([0b1011, 0b1101], np.uint8),
                                        gpu_flow=gpu_flow)
        assert np.array_equal(canon(r[0]), lin([3, 0]))
        assert np.array_equal(canon(r[1]), lin([1, 1]))
    h2.close()
    h3 = oracle.FixedHandle(0, g, 16)
    r = h3.packed_multiexponentiation([8], 3, np.array([1, 1, 1], np.uint8))
    assert np.array_equal(canon(r[0]), lin([1, 1, 1]))
    h3.close()


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_compiled_accessor_files_match_restated_table(oracle, cid, tmp_path):
    """write_to_file / the file constructor of in_memory_partition_table_accessor
    (in_memory_partition_table_accessor.h:42-59,98-105) against the restated file format"""
    from oracle import fixed_base
    proj = GOLDEN[f"curve{cid}_fixed_projective_generators"]
    h = oracle.FixedHandle(cid, proj, 4)
    path = str(tmp_path / "table.bin")
    h.write_to_file(path)
    raw = open(path, "rb").read()
    assert raw == fixed_base.PartitionTable(cid, proj, 4).file_bytes()
    assert hashlib.sha256(raw).digest() == GOLDEN[f"curve{cid}_table_w4_sha256"].tobytes()
    h2 = oracle.FixedHandle(cid, filename=path)
    res = h2.packed_multiexponentiation(GOLDEN["fixed_bit_table"], proj.shape[0],
                                        GOLDEN[f"curve{cid}_fixed_scalars"])
    assert np.array_equal(_canon_all(oracle, cid, res), GOLDEN[f"curve{cid}_fixed_canonical"])
    h.close()
    h2.close()

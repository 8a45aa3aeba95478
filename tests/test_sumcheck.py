"""sxt_prove_sumcheck through the C ABI against the reference's own prover (oracle/ref/
ref_sumcheck.cc: prfsk::prove_sum with its cpu_driver, compiled from /root/reference).  Round
polynomials and the evaluation point must be byte-identical for both fields (curve25519 scalars,
Grumpkin base field elements in Montgomery form), ragged n (rows without a partner), products of
different lengths sharing MLEs, n = 1.  The transcript is the caller's callback: here a
deterministic hash of the round polynomial, so both provers see the same challenges only if
their polynomials agree byte for byte."""
import ctypes
import hashlib

import numpy as np
import pytest

L_ORDER = 2**252 + 27742317777372353535851937790883648493
GK_P = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def elements(rng, field_id, count):
    """`count` canonical field elements in the caller's representation, uint8 [count, 32]"""
    p = L_ORDER if field_id == 0 else GK_P
    out = np.zeros((count, 32), np.uint8)
    for i in range(count):
        v = int.from_bytes(rng.integers(0, 256, 32, dtype=np.uint8).tobytes(), "little") % p
        if field_id == 1:
            v = v * (1 << 256) % p  # Montgomery form, R = 2^256
        out[i] = np.frombuffer(v.to_bytes(32, "little"), np.uint8)
    return out


def challenge_callback(field_id, log):
    p = L_ORDER if field_id == 0 else GK_P

    def cb(r_ptr, ctx, poly_ptr, length):
        poly = ctypes.string_at(poly_ptr, 32 * length)
        log.append(poly)
        v = int.from_bytes(hashlib.sha256(poly).digest(), "little") % p
        ctypes.memmove(r_ptr, v.to_bytes(32, "little"), 32)
    return cb


def product_table(field_id, multipliers, lengths, stride):
    table = np.zeros((len(lengths), stride), np.uint8)
    for i, (m, k) in enumerate(zip(multipliers, lengths)):
        table[i, :32] = m
        table[i, 32:36] = np.frombuffer(np.uint32(k).tobytes(), np.uint8)
    return table


CASES = [  # n, num_mles, products as lists of MLE indices
    (1, 1, [[0]]),
    (2, 2, [[0, 1]]),
    (5, 3, [[0, 1], [2]]),
    (8, 4, [[0, 1, 2], [3], [1, 3]]),
    (37, 5, [[0, 1, 2, 3], [4, 0], [2]]),
    (300, 3, [[0, 1, 2], [0, 0]]),
]


def run_case(api, oracle, field_id, n, num_mles, products, seed):
    rng = np.random.default_rng(seed)
    mles = elements(rng, field_id, n * num_mles).reshape(num_mles, n, 32)
    lengths = [len(t) for t in products]
    terms = [i for t in products for i in t]
    mults = elements(rng, field_id, len(products))
    degree = max(lengths)
    stride = oracle.sumcheck_product_stride(field_id)
    assert stride == api.SUMCHECK_PRODUCT_STRIDE[field_id]
    table = product_table(field_id, mults, lengths, stride)
    want_log, got_log = [], []
    want = oracle.prove_sumcheck(field_id, mles, table, terms, n, degree,
                                 challenge_callback(field_id, want_log))
    got = api.prove_sumcheck(field_id, mles, table, terms, n, degree,
                             challenge_callback(field_id, got_log))
    assert np.array_equal(got[0], want[0]), "round polynomials differ"
    assert np.array_equal(got[1], want[1]), "evaluation points differ"
    assert got_log == want_log and len(got_log) == max((n - 1).bit_length(), 1)


@pytest.mark.parametrize("field_id", [0, 1])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_sumcheck_host_backend(cpu_backend, oracle, field_id, case):
    n, num_mles, products = CASES[case]
    run_case(cpu_backend, oracle, field_id, n, num_mles, products, 50 + case)


@pytest.mark.gpu
@pytest.mark.parametrize("field_id", [0, 1])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_sumcheck_gpu_backend(gpu_backend, oracle, field_id, case):
    n, num_mles, products = CASES[case]
    before = gpu_backend.load().bzamd_kernel_launch_count()
    run_case(gpu_backend, oracle, field_id, n, num_mles, products, 150 + case)
    assert gpu_backend.load().bzamd_kernel_launch_count() > before


@pytest.mark.gpu
@pytest.mark.parametrize("field_id", [0, 1])
def test_sumcheck_gpu_backend_long(gpu_backend, oracle, field_id):
    """2^14 + 77 rows, degree 3: every workgroup of k_sumcheck_round contributes"""
    run_case(gpu_backend, oracle, field_id, (1 << 14) + 77, 4, [[0, 1, 2], [3, 1], [2]], 9)


def _reentrant_case(api, oracle, field_id):
    """the transcript callback calls back into the library while the prover is between rounds
    (round 3 held a process-wide lock across it and would have deadlocked)"""
    n, num_mles, products = 300, 3, [[0, 1], [2]]
    rng = np.random.default_rng(77)
    mles = elements(rng, field_id, n * num_mles).reshape(num_mles, n, 32)
    lengths = [len(t) for t in products]
    terms = [i for t in products for i in t]
    mults = elements(rng, field_id, len(products))
    table = product_table(field_id, mults, lengths, api.SUMCHECK_PRODUCT_STRIDE[field_id])
    inner, seen = challenge_callback(field_id, []), []

    def callback(r_ptr, ctx, poly_ptr, length):
        seen.append(api.get_one_commit(3).copy())  # a blocking sxt_* call from inside the callback
        inner(r_ptr, ctx, poly_ptr, length)

    want = oracle.prove_sumcheck(field_id, mles, table, terms, n, max(lengths),
                                 challenge_callback(field_id, []))
    got = api.prove_sumcheck(field_id, mles, table, terms, n, max(lengths), callback)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert len(seen) == 9 and all(np.array_equal(s, oracle.one_commit(3)) for s in seen)


def test_sumcheck_callback_may_reenter_host_backend(cpu_backend, oracle):
    _reentrant_case(cpu_backend, oracle, 0)


@pytest.mark.gpu
def test_sumcheck_callback_may_reenter_gpu_backend(gpu_backend, oracle):
    _reentrant_case(gpu_backend, oracle, 0)

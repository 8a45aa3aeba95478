"""Inner-product argument (sxt_curve25519_prove_inner_product / _verify_inner_product, SURVEY 8(f)
rank 4) through the C ABI against the reference's own prover and verifier
(oracle/ref/ref_inner_product.cc: prfip::prove_inner_product / verify_inner_product compiled from
/root/reference).  The proof (compressed L and R points per round, the final scalar) and the
203-byte transcript state after the call must be byte-identical; proofs made here verify there
and vice versa; tampered proofs are rejected.  Cases follow the reference's own exerciser
(sxt/proof/inner_product/driver_test.cc, cbindings/inner_product_proof.t.cc): n = 1, powers of
two, ragged n, generator offsets, unreduced input scalars."""
import numpy as np
import pytest

L_ORDER = 2**252 + 27742317777372353535851937790883648493


def scalars(rng, n, reduced=True):
    raw = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    if not reduced:
        return raw
    vals = [int.from_bytes(row.tobytes(), "little") % L_ORDER for row in raw]
    return np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), np.uint8).reshape(n, 32)


def test_transcript_matches_reference(oracle):
    from blitzar_amd import api
    for label in ("abc", "", "a much longer application label 0123456789" * 5):
        assert np.array_equal(api.transcript_new(label), oracle.transcript_new(label))


def check_prove_and_verify(api, oracle, n, offset, seed, reduced=True):
    rng = np.random.default_rng(seed)
    a, b = scalars(rng, n, reduced), scalars(rng, n, reduced)
    t0 = oracle.transcript_new("blitzar_amd test")
    want_l, want_r, want_ap, want_t = oracle.ip_prove(t0, n, offset, a, b)
    got_l, got_r, got_ap, got_t = api.prove_inner_product(t0, n, offset, a, b)
    assert np.array_equal(got_l, want_l) and np.array_equal(got_r, want_r)
    assert np.array_equal(got_ap, want_ap)
    assert np.array_equal(got_t, want_t)
    # verification, both directions and both implementations
    np_ = 1 << oracle._rounds(n)
    gens = oracle.ristretto_generators(np_ + 1, offset)
    commit = oracle.msm_projective(0, [(a, False)], gens[:n])[0]
    product = oracle.s25_inner_product(a, b)
    ok, t_after = api.verify_inner_product(t0, n, offset, b, product, commit, got_l, got_r, got_ap)
    ref_ok, ref_t_after = oracle.ip_verify(t0, n, offset, b, product, commit, got_l, got_r, got_ap)
    assert ok and ref_ok and np.array_equal(t_after, ref_t_after)
    # tampering: final scalar, product, one L, one b entry
    bad_ap = got_ap.copy()
    bad_ap[0] ^= 1
    assert not api.verify_inner_product(t0, n, offset, b, product, commit, got_l, got_r, bad_ap)[0]
    bad_product = product.copy()
    bad_product[3] ^= 0x10
    assert not api.verify_inner_product(t0, n, offset, b, bad_product, commit, got_l, got_r,
                                        got_ap)[0]
    if n > 1:
        bad_l = got_l.copy()
        bad_l[-1] = got_r[-1]
        assert not api.verify_inner_product(t0, n, offset, b, product, commit, bad_l, got_r,
                                            got_ap)[0]
        bad_b = b.copy()
        bad_b[n - 1, 0] ^= 1
        assert not api.verify_inner_product(t0, n, offset, bad_b, product, commit, got_l, got_r,
                                            got_ap)[0]
        # a different transcript label changes every challenge
        other = oracle.transcript_new("another label")
        assert not api.verify_inner_product(other, n, offset, b, product, commit, got_l, got_r,
                                            got_ap)[0]


@pytest.mark.parametrize("n,offset", [(1, 0), (2, 0), (3, 5), (4, 0), (5, 1000), (8, 3), (31, 0),
                                      (64, 7), (100, 0)])
def test_prove_and_verify_host_backend(cpu_backend, oracle, n, offset):
    check_prove_and_verify(cpu_backend, oracle, n, offset, 40 + n)


def test_unreduced_scalars_host_backend(cpu_backend, oracle):
    """inputs need not be reduced modulo the group order (the reference multiplies them as given)"""
    check_prove_and_verify(cpu_backend, oracle, 6, 2, 77, reduced=False)


@pytest.mark.gpu
@pytest.mark.parametrize("n,offset", [(1, 0), (2, 0), (5, 9), (64, 0), (1000, 17), (4096, 0)])
def test_prove_and_verify_gpu_backend(gpu_backend, oracle, n, offset):
    before = gpu_backend.load().bzamd_kernel_launch_count()
    check_prove_and_verify(gpu_backend, oracle, n, offset, 140 + n)
    if n > 1:
        assert gpu_backend.load().bzamd_kernel_launch_count() > before


@pytest.mark.gpu
def test_unreduced_scalars_gpu_backend(gpu_backend, oracle):
    check_prove_and_verify(gpu_backend, oracle, 37, 4, 78, reduced=False)

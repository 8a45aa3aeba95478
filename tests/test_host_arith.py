"""The product's field / curve headers (shared verbatim by the host backend and the gfx950
kernels), compiled for the host and compared with the reference oracle.

GF(2^255-19) raw limbs are observable through the C ABI (sxt_ristretto255_get_generators returns
5x51 limbs), so f51 and the generator derivation are checked limb-for-limb; Montgomery fields are
canonical, so any correct schedule must equal the reference's limbs; group operations are
compared after canonicalisation (the parity contract of SURVEY 8(a))."""
import ctypes

import numpy as np
import pytest

from tests import hooks, util

MASK51 = (1 << 51) - 1


def rand_f51(rng, loose):
    """limbs as they occur in practice: < 2^51 ("tight") or a sum of two such ("loose")"""
    v = rng.integers(0, 1 << 51, 5, dtype=np.uint64)
    if loose:
        v = v + rng.integers(0, 1 << 51, 5, dtype=np.uint64)
    return v


def test_f51_limb_exact(oracle):
    rng = np.random.default_rng(1)
    lib = oracle.lib()

    def ref(op, *args):
        out = np.zeros(5, np.uint64)
        getattr(lib, f"ref_f51_{op}")(out.ctypes.data_as(ctypes.c_void_p),
                                      *[a.ctypes.data_as(ctypes.c_void_p) for a in args])
        return out

    edge = [np.zeros(5, np.uint64), np.full(5, MASK51, np.uint64),
            np.array([MASK51 - 18, MASK51, MASK51, MASK51, MASK51], np.uint64),  # p
            np.array([1, 0, 0, 0, 0], np.uint64)]
    cases = [(rand_f51(rng, i % 2 == 0), rand_f51(rng, i % 3 == 0)) for i in range(200)]
    cases += [(a, b) for a in edge for b in edge]
    for f, g in cases:
        assert np.array_equal(hooks.f51("mul", f, g), ref("mul", f, g))
        assert np.array_equal(hooks.f51("sq", f), ref("sq", f))
        assert np.array_equal(hooks.f51("sub", f, g), ref("sub", f, g))
    for f, _ in cases[:20]:
        assert np.array_equal(hooks.f51("invert", f), ref("invert", f))


def test_builtin_generator_derivation_limb_exact(oracle):
    for first, n in ((0, 40), (12345, 8), (2**40, 4), (2**64 - 6, 4)):
        assert np.array_equal(hooks.ed_base_elements(first, n),
                              oracle.ristretto_generators(n, first))


def test_ed25519_group_ops(oracle):
    g = oracle.ristretto_generators(12, 3)
    canon = oracle.ristretto_compress
    for i in range(0, 12, 2):
        a, b = g[i], g[i + 1]
        # the raw-limb add is what sxt_curve25519_get_one_commit exposes: limb-exact
        assert np.array_equal(hooks.ed_add(a, b), oracle.add_projective(0, a, b))
        assert np.array_equal(canon(hooks.ed_dbl(a)), canon(oracle.double_projective(0, a)))
        assert np.array_equal(canon(hooks.ed_add(a, a)), canon(oracle.double_projective(0, a)))
        d5 = a
        for _ in range(5):
            d5 = oracle.double_projective(0, d5)
        assert np.array_equal(canon(hooks.ed_dbl(a, 5)), canon(d5))
        # a - b + b == a ; a + (-a) == identity
        assert np.array_equal(canon(hooks.ed_add(hooks.ed_sub(a, b), b)), canon(a))
        assert np.array_equal(canon(hooks.ed_add(a, hooks.ed_neg(a))), np.zeros(32, np.uint8))
        # ristretto encode / decode
        enc = hooks.ristretto_encode(a)
        assert np.array_equal(enc, canon(a))
        assert np.array_equal(hooks.ristretto_encode(hooks.ristretto_decode(enc)), enc)


@pytest.mark.parametrize("cid", [1, 2, 3])
def test_montgomery_fields_limb_exact(oracle, cid):
    rng = np.random.default_rng(10 + cid)
    pfx, nl = oracle.CURVES[cid][0], oracle.CURVES[cid][1]
    # field elements in Montgomery form, always reduced: take coordinates of curve points
    gens = util.weierstrass_generators(cid, 24, distinct_seeds=24)
    elems = [gens[i, :8 * nl].view(np.uint64) for i in range(24) if i != 5]
    elems += [gens[i, 8 * nl:16 * nl].view(np.uint64) for i in range(24) if i != 5]
    elems.append(np.zeros(nl, np.uint64))
    lib = oracle.lib()
    for _ in range(200):
        f = elems[int(rng.integers(len(elems)))]
        g = elems[int(rng.integers(len(elems)))]
        want = np.zeros(nl, np.uint64)
        getattr(lib, f"ref_{pfx}_field_mul")(want.ctypes.data_as(ctypes.c_void_p),
                                             f.ctypes.data_as(ctypes.c_void_p),
                                             g.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(hooks.sw_field_mul(cid, f, g), want)
        s, d = hooks.sw_field_addsub(cid, f, g)
        # (f + g) - g == f and (f - g) + g == f in canonical form
        assert np.array_equal(hooks.sw_field_addsub(cid, s, g)[1], f)
        assert np.array_equal(hooks.sw_field_addsub(cid, d, g)[0], f)


@pytest.mark.parametrize("cid", [1, 2, 3])
def test_weierstrass_group_ops(oracle, cid):
    nl = oracle.CURVES[cid][1]
    gens = util.weierstrass_generators(cid, 10, distinct_seeds=10)
    proj = oracle.affine_to_projective(cid, gens)
    ident = proj[5]  # tests/util.py places the identity at index 5
    canon = lambda p: oracle.to_affine(cid, p)  # noqa: E731
    for i in (0, 1, 2, 3, 6, 7):
        a, b = proj[i], proj[i + 1]
        want = oracle.add_projective(cid, a, b)
        assert np.array_equal(canon(hooks.sw_add(cid, a, b)), canon(want))
        if i + 1 != 5:
            assert np.array_equal(canon(hooks.sw_add_mixed(cid, a, b[:2 * nl])), canon(want))
        dbl = oracle.double_projective(cid, a)
        assert np.array_equal(canon(hooks.sw_dbl(cid, a)), canon(dbl))
        assert np.array_equal(canon(hooks.sw_add(cid, a, a)), canon(dbl))      # complete formulas
        assert np.array_equal(canon(hooks.sw_add_mixed(cid, a, a[:2 * nl])), canon(dbl))
        assert np.array_equal(canon(hooks.sw_add(cid, a, ident)), canon(a))
        assert np.array_equal(canon(hooks.sw_add(cid, ident, a)), canon(a))
        assert np.array_equal(canon(hooks.sw_add_mixed(cid, ident, a[:2 * nl])), canon(a))
        xy, inf = hooks.sw_to_affine(cid, want)
        ref_aff = canon(want)
        assert not inf and np.array_equal(xy.view(np.uint8), ref_aff[:16 * nl])
    xy, inf = hooks.sw_to_affine(cid, ident)
    assert inf and np.array_equal(xy.view(np.uint8), oracle.identity_affine(cid)[:16 * nl])
    if cid == 1:
        for i in (0, 3, 5, 8):
            assert np.array_equal(hooks.bls_compress(proj[i]), oracle.bls_compress(proj[i]))


def test_signed_digit_recoding_reconstructs_the_scalar():
    rng = np.random.default_rng(3)
    for _ in range(400):
        width = int(rng.integers(1, 257))
        offset = int(rng.integers(0, 8))
        signed = bool(rng.integers(0, 2)) and width <= 128 and width >= 2
        c = int(rng.integers(2, 17 if not signed else 16))
        nbytes = (offset + width + 7) // 8
        raw = rng.integers(0, 256, nbytes, dtype=np.uint8)
        if rng.integers(0, 6) == 0:
            raw[:] = 0xff
        whole = int.from_bytes(raw.tobytes(), "little")
        x = (whole >> offset) & ((1 << width) - 1)
        if signed and x >> (width - 1):
            x -= 1 << width
        W = (width + 1 + c - 1) // c
        d = hooks.recode(raw, offset, width, signed, c, W)
        assert sum(int(d[w]) << (c * w) for w in range(W)) == x
        assert all(abs(int(v)) <= 1 << (c - 1) for v in d)


def test_planner_invariants():
    ns = [0, 1, 97, 5000, 1 << 16, 1 << 20, (1 << 20) + 3, 1 << 22]
    widths = [256, 8, 32, 256, 64, 256, 1, 256]
    per, totals = hooks.plan(ns, widths, [0] * len(ns))
    assert per[0].tolist() == [1, 0, 0, 0]  # empty column: no tasks
    covered = 0
    for n, bw, (c, W, G, rpg) in zip(ns, widths, per.tolist()):
        if n == 0:
            continue
        assert 2 <= c <= 16 and W * c >= bw + 1          # top digit never carries out
        assert G * rpg >= n and (G - 1) * rpg < n        # groups tile the rows
        assert rpg <= 1 << 20                             # 31-bit row index + sign per entry
        covered += W * n
    assert int(totals[4]) == covered
    assert int(totals[0]) == sum(W * G for (_, W, G, _) in per.tolist())
    # signed columns are planned with c <= 15 by the engine (digits must fit int16 negated)
    per, _ = hooks.plan([1 << 20], [128], [1], max_window_bits=15)
    assert per[0][0] <= 15

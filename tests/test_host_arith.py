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
        # (widths above 16: the merged tasks of wide window tables, unsigned columns only)
        c = int(rng.integers(2, 21 if not signed else 16))
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
    ns = [0, 1, 97, 5000, 1 << 16, (1 << 16) + 1, 1 << 20, (1 << 20) + 3, 1 << 22, 1 << 26,
          (1 << 31) - 1]
    widths = [256, 8, 32, 256, 64, 16, 256, 1, 256, 256, 128]
    per, totals = hooks.plan(ns, widths, [0] * len(ns))
    assert per[0].tolist()[:3] == [1, 0, 0]  # empty column: no tasks
    seg = 1 << int(totals[6])  # sorted entries per accumulation lane, one value per launch
    assert seg == 128          # a launch with this many entries takes the longest segments
    covered = tasks = segs = part = 0
    for n, bw, (c, W, slices, first, slice_rows, s) in zip(ns, widths, per.tolist()):
        if n == 0:
            continue
        assert first == tasks
        assert 2 <= c <= 16 and W * c >= bw + 1          # top digit never carries out
        # partition geometry of the two-pass sort: slices cover the rows, the 32-bit record
        # sign | bucket-in-group (s bits) | row has room for every row index
        assert slice_rows in (16384, 65536)
        assert slices == (n + slice_rows - 1) // slice_rows
        assert 0 <= s <= min(c - 1, 10) and n <= 1 << (31 - s)
        groups = 1 << (c - 1 - s)
        covered += W * n
        tasks += W
        segs += W * ((n + seg - 1) // seg)
        part += W * (groups + 1)
    assert int(totals[0]) == tasks and int(totals[4]) == covered
    assert int(totals[3]) == segs and int(totals[5]) == part
    # config 2 / config 3 shapes: groups of ~4096 entries, slices staged in LDS
    per, totals = hooks.plan([1 << 20, 1 << 22], [256, 256], [0, 0])
    assert per[0].tolist()[4:] == [16384, 7] and per[1].tolist()[4:] == [16384, 5]
    # work per lane of the bucket kernels: a single column is latency-bound (32 entries per
    # accumulation lane so that its lanes fill the machine, 8 buckets per reduce lane: shortest
    # chain); 256 columns are throughput-bound (128 entries, 64 buckets: least total work) and
    # price a bucket at its throughput cost, which still leaves c = 16 for 2^20 rows
    _, totals = hooks.plan([1 << 20], [256], [0])
    assert totals[6:].tolist() == [5, 3]
    _, totals = hooks.plan([1 << 22], [256], [0])
    assert totals[6:].tolist() == [7, 3]
    per, totals = hooks.plan([1 << 20] * 256, [256] * 256, [0] * 256)
    assert totals[6:].tolist() == [7, 6] and per[0][0] == 16
    # a call in throughput mode: k_reduce has a whole step to finish beside the next call and the
    # machine runs at its power limit, so long columns take twice the buckets per lane (fewer
    # additions); short ones (under 2^23 entries: their step is all tail) and many-column calls
    # that are at the throughput geometry anyway keep theirs
    _, totals = hooks.plan([1 << 20], [256], [0], in_sequence=True)
    assert totals[6:].tolist() == [5, 4]
    _, totals = hooks.plan([1 << 22], [256], [0], in_sequence=True)
    assert totals[6:].tolist() == [7, 4]
    _, totals = hooks.plan([1 << 18], [256], [0], in_sequence=True)
    assert totals[7] == 3
    for columns, plain, in_sequence in ((2, 3, 4), (4, 4, 4), (32, 6, 6)):
        shape = ([1 << 20] * columns, [256] * columns, [0] * columns)
        assert hooks.plan(*shape)[1][7] == plain
        assert hooks.plan(*shape, in_sequence=True)[1][7] == in_sequence
    _, totals = hooks.plan([1 << 24], [8], [0], in_sequence=True)   # 128 buckets per task
    assert totals[7] == 0
    # narrow columns: 256 (1 byte) or 1024 (4 bytes) buckets per task spread over the 256 lanes
    # of their one reduce block, down to one bucket per lane
    _, totals = hooks.plan([1 << 20], [8], [0])
    assert totals[6:].tolist() == [4, 0]     # (2^20 entries: 16 per lane fill every SIMD, below)
    _, totals = hooks.plan([1 << 20], [32], [0])
    assert totals[6:].tolist() == [5, 1]     # (a lone launch of few buckets: two per lane, below)
    # a LONE short launch (<= 2^18 buckets in all) takes two buckets per reduce lane -- k_reduce is the
    # longer of its two tail chains -- and keeps the sequence's geometry in throughput mode
    _, totals = hooks.plan([1 << 16], [256], [0])
    assert totals[7] == 1
    _, totals = hooks.plan([1 << 16], [256], [0], in_sequence=True)
    assert totals[7] == 3
    # ... and a short launch cannot fill the machine at 32 entries per accumulation lane: fewer, down
    # to 8, while it has less than a wavefront per SIMD (2^14 rows: k_accumulate 0.10 -> 0.04 ms lone)
    assert [int(hooks.plan([1 << k], [256], [0])[1][6]) for k in (12, 14, 16, 18, 20)] == [3, 3, 4, 5, 5]
    # many short columns (the reference's bucket_method2 regime): entries per accumulation lane follow
    # the rows of a task (4096 rows: 128 lanes x 32 entries, never half a wavefront of 128-entry lanes),
    # and the 256 buckets of a task go to 64 reduce lanes x 4
    per, totals = hooks.plan([4096] * 1024, [256] * 1024, [0] * 1024)
    assert per[0][0] == 9 and totals[6:].tolist() == [5, 2]
    per, totals = hooks.plan([1024] * 1024, [256] * 1024, [0] * 1024)
    assert totals[6] == 3
    # ... and a task costs something whatever its size (its k_reduce block, its share of the sort): the
    # window-width model takes one bit more than rows + buckets alone would (measured: 1024 rows c = 8,
    # 3.50 against 3.73 ms at c = 7; 256 rows c = 7, 2.58 against 3.02 at c = 6); a lone column does not
    assert per[0][0] == 8
    assert hooks.plan([256] * 1024, [256] * 1024, [0] * 1024)[0][0][0] == 7
    assert hooks.plan([1024], [256], [0])[0][0][0] == 7
    assert hooks.plan([256], [256], [0])[0][0][0] == 6
    # ... by the task a typical ENTRY lives in: one 2^22-row column among 200 columns of 256 rows keeps
    # the long column's geometry (its 17 tasks hold 99.9 % of the entries)
    _, totals = hooks.plan([1 << 22] + [256] * 200, [256] * 201, [0] * 201)
    assert totals[6] >= 5
    # blocks of the bucket reduction stay full: 2^13 buckets per task leave 32 per lane
    per, totals = hooks.plan([1 << 20] * 256, [256] * 256, [0] * 256, max_window_bits=14)
    assert per[0][0] == 14 and totals[6:].tolist() == [7, 5]
    # signed columns are planned with c <= 15 by the engine (digits must fit int16 negated)
    per, _ = hooks.plan([1 << 20], [128], [1], max_window_bits=15)
    assert per[0][0] <= 15


def test_recode_from_aligned_words():
    """digit_recoder::load_words32 (what k_recode_packed reads from its LDS tile) == ::load"""
    rng = np.random.default_rng(9)
    for _ in range(300):
        width = int(rng.integers(1, 257))
        offset = int(rng.integers(0, 64))
        signed = bool(rng.integers(0, 2)) and width <= 128
        c = int(rng.integers(2, 16 if signed else 17))
        windows = (width + 1 + c - 1) // c
        row = rng.integers(0, 256, 48, dtype=np.uint8)
        want = hooks.recode(row, offset, width, signed, c, windows)
        for skew in range(4):
            assert np.array_equal(hooks.recode_words32(row, skew, offset, width, signed, c, windows),
                                  want)


def test_packed_recode_ranges():
    """which batches k_recode_packed takes (packed fixed-base calls: bit fields of the same rows)
    and how their columns are cut into LDS tiles of at most 1984 row bytes"""
    # the reference's packed layout: outputs of 8 / 32 / 256 bits back to back (blitzar_api.h:688-712)
    widths = [[8, 32, 256][i % 3] for i in range(128)]
    bit_pos = np.concatenate([[0], np.cumsum(widths)[:-1]])
    stride = (sum(widths) + 7) // 8
    ranges = hooks.packed_ranges(bit_pos // 8, [stride] * 128, bit_pos % 8, widths)
    assert ranges == [(0, 128, 0, stride)]                     # 1579 bytes: one tile
    # a wider row is cut where a column would end beyond 1984 bytes from the range's base
    widths = [256] * 100
    bit_pos = np.arange(100) * 256
    ranges = hooks.packed_ranges(bit_pos // 8, [3200] * 100, bit_pos % 8, widths)
    assert ranges == [(0, 62, 0, 1984), (62, 38, 1984, 1216)]
    # unaligned fields: the span counts the straddled bytes
    ranges = hooks.packed_ranges([0, 0, 1], [4, 4, 4], [0, 3, 1], [3, 6, 23])
    assert ranges == [(0, 3, 0, 4)]
    # not packed: separate buffers, different strides, descending bases, a null column
    assert hooks.packed_ranges([0, 3200], [32, 32], [0, 0], [256, 256]) == []
    assert hooks.packed_ranges([0, 4], [32, 16], [0, 0], [8, 8]) == []
    assert hooks.packed_ranges([4, 0], [32, 32], [0, 0], [8, 8]) == []
    assert hooks.packed_ranges([0, 2**64 - 1], [32, 32], [0, 0], [8, 8]) == []
    assert hooks.packed_ranges([0], [32], [0], [8]) == []      # single columns use k_recode


#--------------------------------------------------------------------------------------------------
# the 9 x 29-bit field / curve code the gfx950 kernels compute in (field/f29.h, curve/ed29.h)
#--------------------------------------------------------------------------------------------------
P25519 = (1 << 255) - 19


def f29_value(limbs):
    return sum(int(v) << (29 * i) for i, v in enumerate(limbs)) % P25519


def f29_loose(rng, bound):
    """limbs up to bound * 2^29 (the B of field/f29.h's contract), with saturated corner cases"""
    hi = int(bound * (1 << 29))
    v = rng.integers(0, hi, 9, dtype=np.uint64).astype(np.uint32)
    k = int(rng.integers(0, 4))
    if k == 0:
        v[:] = hi - 1
    elif k == 1:
        v[int(rng.integers(0, 9))] = hi - 1
    return v


def test_f29_matches_integers_at_the_contract_bounds():
    rng = np.random.default_rng(29)
    for _ in range(300):
        bf = float(rng.choice([1.0, 2.0, 3.0]))
        bg = float(rng.choice([1.0, 2.0])) if bf > 2 else float(rng.choice([1.0, 2.0, 3.0]))
        f, g = f29_loose(rng, bf + 0.001), f29_loose(rng, bg + 0.001)
        h = hooks.f29("mul", f, g)
        assert f29_value(h) == f29_value(f) * f29_value(g) % P25519
        assert max(int(x) for x in h) < (1 << 29) + (1 << 18)
        assert hooks.f29_to_int(h) == f29_value(h)
        f2 = f29_loose(rng, 2.4)
        s = hooks.f29("sq", f2)
        assert f29_value(s) == f29_value(f2) ** 2 % P25519
        assert max(int(x) for x in s) < (1 << 29) + (1 << 18)
        # sub: B(g) < 1.99, any f up to B 5
        f5, g2 = f29_loose(rng, 5.0), f29_loose(rng, 1.98)
        d = hooks.f29("sub", f5, g2)
        assert f29_value(d) == (f29_value(f5) - f29_value(g2)) % P25519
        w = hooks.f29("weak_reduce", f29_loose(rng, 7.0))
        assert max(int(x) for x in w) <= (1 << 29)
    z = f29_loose(rng, 1.0)
    assert f29_value(hooks.f29("invert", z)) * f29_value(z) % P25519 == 1


def test_f29_fe51_conversions():
    rng = np.random.default_rng(30)
    edge = [np.zeros(5, np.uint64), np.full(5, MASK51, np.uint64),
            np.array([MASK51 - 18, MASK51, MASK51, MASK51, MASK51], np.uint64),
            np.array([MASK51 - 19, MASK51, MASK51, MASK51, MASK51], np.uint64),
            np.full(5, (1 << 54) - 1, np.uint64)]
    cases = edge + [rand_f51(rng, i % 2 == 0) for i in range(200)]
    for f in cases:
        want = sum(int(v) << (51 * i) for i, v in enumerate(f)) % P25519
        h = hooks.f29_from_fe51(f)
        assert f29_value(h) == want
        assert hooks.f29_to_int(h) == want      # canonical: exactly the residue in [0, p)


def test_ed29_group_law_matches_reference(oracle):
    g = oracle.ristretto_generators(40, 11)
    canon = oracle.ristretto_compress
    for i in range(0, 12, 2):
        a, b = g[i], g[i + 1]
        assert np.array_equal(canon(hooks.ed29_add(a, b)), canon(oracle.add_projective(0, a, b)))
        assert np.array_equal(canon(hooks.ed29_add(a, b, True)), canon(hooks.ed_sub(a, b)))
        assert np.array_equal(canon(hooks.ed29_add(a, a)), canon(oracle.double_projective(0, a)))
        assert np.array_equal(canon(hooks.ed29_add(a, a, True)), np.zeros(32, np.uint8))
        # k_accumulate's form: packed row gathered in sign order, 2dT negated by xor-add
        for neg in (False, True):
            assert np.array_equal(canon(hooks.ed29_add_gathered(a, b, neg)),
                                  canon(hooks.ed29_add(a, b, neg)))
        d = a
        for k in range(1, 18):
            d = oracle.double_projective(0, d)
            if k in (1, 2, 16, 17):
                assert np.array_equal(canon(hooks.ed29_dbl_n(a, k)), canon(d))
    # a long signed accumulation chain (what one bucket lane does), including the identity start
    rng = np.random.default_rng(31)
    signs = rng.integers(0, 2, 40)
    acc = oracle.one_commit(0)
    for q, s in zip(g, signs):
        acc = hooks.ed_sub(acc, q) if s else oracle.add_projective(0, acc, q)
    assert np.array_equal(canon(hooks.ed29_chain(g, signs)), canon(acc))
    assert np.array_equal(canon(hooks.ed29_chain(g, signs, niels=True)), canon(acc))
    # k_accumulate loads the first entry of a segment instead of adding it to the identity
    ident = oracle.one_commit(0)
    for n in (1, 2, 40):
        for first_sign in (0, 1):
            sg = np.array(signs[:n])
            sg[0] = first_sign
            want = ident
            for q, s in zip(g[:n], sg):
                want = hooks.ed_sub(want, q) if s else oracle.add_projective(0, want, q)
            for niels in (False, True):
                assert np.array_equal(canon(hooks.ed29_chain_first(g[:n], sg, niels=niels)),
                                      canon(want))
    # doubling and cancellation through the Z = 1 addends (unified formulas)
    twice = np.stack([g[0], g[0], g[1], g[1]])
    assert np.array_equal(canon(hooks.ed29_chain(twice, [0, 0, 0, 1], niels=True)),
                          canon(oracle.double_projective(0, g[0])))


#--------------------------------------------------------------------------------------------------
# unsaturated-limb Montgomery fields / Weierstrass curves of the gfx950 kernels
# (field/mont29.h, curve/sw29.h); the hook library asserts the limb contracts (BZ_MONT29_CHECK)
#--------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cid", [1, 2, 3])
def test_mont29_field_matches_reference(oracle, cid):
    rng = np.random.default_rng(50 + cid)
    pfx, nl = oracle.CURVES[cid][0], oracle.CURVES[cid][1]
    gens = util.weierstrass_generators(cid, 20, distinct_seeds=20)
    elems = [gens[i, :8 * nl].view(np.uint64) for i in range(20) if i != 5]
    elems += [gens[i, 8 * nl:16 * nl].view(np.uint64) for i in range(20) if i != 5]
    elems.append(np.zeros(nl, np.uint64))
    one = oracle.identity_affine(cid)[8 * nl:16 * nl].view(np.uint64)  # R mod p
    elems.append(one)
    lib = oracle.lib()
    for f in elems:
        assert np.array_equal(hooks.sw29_field(cid, "roundtrip", f), f)
    for _ in range(150):
        f = elems[int(rng.integers(len(elems)))]
        g = elems[int(rng.integers(len(elems)))]
        want = np.zeros(nl, np.uint64)
        getattr(lib, f"ref_{pfx}_field_mul")(want.ctypes.data_as(ctypes.c_void_p),
                                             f.ctypes.data_as(ctypes.c_void_p),
                                             g.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(hooks.sw29_field(cid, "mul", f, g), want)
    # the inversion (field/mont29.h: Kaliski's almost inverse with bulk shifts): random elements and
    # the ones that stress its shifts -- powers of two (long runs of trailing zeros, whole zero
    # words), 1, p - 1, small odd values
    special = [one]
    for bit in (1, 63, 64, 65, 127, 128, 200, 64 * nl - 4):
        w = np.zeros(nl, np.uint64)
        w[bit // 64] = np.uint64(1) << np.uint64(bit % 64)
        special.append(w)            # Montgomery form of some element: any canonical value < p
    for small in (2, 3, 5, 0xffffffffffffffff):
        w = np.zeros(nl, np.uint64)
        w[0] = np.uint64(small)
        special.append(w)
    for f in list(elems[:38]) + special:
        inv = hooks.sw29_field(cid, "invert", f)
        assert np.array_equal(hooks.sw29_field(cid, "mul", f, inv), one)
    # mul2: (2a) b + c (3d) with ONE Montgomery reduction == 2 ab + 3 cd
    p = {1: 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
         2: 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47,
         3: 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001}[cid]
    to_int = lambda w: int.from_bytes(w.tobytes(), "little")  # noqa: E731
    r_inv = pow(1 << (64 * nl), -1, p)
    for _ in range(60):
        a, b, c, d = (elems[int(rng.integers(len(elems)))] for _ in range(4))
        got = to_int(hooks.sw29_field(cid, "mul2", a, b, c, d))
        want = (2 * to_int(a) * to_int(b) + 3 * to_int(c) * to_int(d)) * r_inv % p
        assert got == want


@pytest.mark.parametrize("cid", [1, 2, 3])
def test_sw29_group_law_matches_reference(oracle, cid):
    nl = oracle.CURVES[cid][1]
    gens = util.weierstrass_generators(cid, 40, distinct_seeds=12)
    proj = oracle.affine_to_projective(cid, gens)
    ident = proj[5]
    canon = lambda p: oracle.to_affine(cid, p)  # noqa: E731
    for i in (0, 1, 2, 3, 6, 7):
        a, b = proj[i], proj[i + 1]
        assert np.array_equal(canon(hooks.sw29_add(cid, a, b)), canon(oracle.add_projective(cid, a, b)))
        assert np.array_equal(canon(hooks.sw29_add(cid, a, a)), canon(oracle.double_projective(cid, a)))
        assert np.array_equal(canon(hooks.sw29_add(cid, a, ident)), canon(a))
        assert np.array_equal(canon(hooks.sw29_add(cid, ident, a)), canon(a))
        d = a
        for k in range(1, 18):
            d = oracle.double_projective(cid, d)
            if k in (1, 2, 16, 17):
                assert np.array_equal(canon(hooks.sw29_dbl_n(cid, a, k)), canon(d))
    assert np.array_equal(canon(hooks.sw29_dbl_n(cid, ident, 3)), canon(ident))
    # a long signed accumulation chain from the identity (what one bucket lane does), with
    # repeated points (complete formulas: doubling and cancellation inside the chain)
    rng = np.random.default_rng(60 + cid)
    order = [i for i in range(40) if i != 5]
    order = order + order[:7] + order[:3]
    signs = rng.integers(0, 2, len(order))
    signs[-3:] = 1 - signs[:3]  # the last three cancel the first three
    acc = ident
    for i, s in zip(order, signs):
        q = proj[i].copy()
        if s:
            neg = oracle.add_projective(cid, ident, q)  # copy
            # -q: negate Y via the oracle: 0 - q is not exposed, use p - y through canonical affine
            aff = oracle.to_affine(cid, q).copy()
            y = int.from_bytes(aff[8 * nl:16 * nl].tobytes(), "little")
            pmod = {1: 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
                    2: 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47,
                    3: 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001}[cid]
            aff[8 * nl:16 * nl] = np.frombuffer(((pmod - y) % pmod).to_bytes(8 * nl, "little"), np.uint8)
            q = oracle.affine_to_projective(cid, aff)[0]
            del neg
        acc = oracle.add_projective(cid, acc, q)
    xy = np.stack([gens[i, :16 * nl].view(np.uint64) for i in order])
    got = hooks.sw29_chain(cid, ident, xy, signs)
    assert np.array_equal(canon(got), canon(acc))
    # a few thousand additions: the hook aborts if k_accumulate's form (add_mixed_acc) ever leaves
    # its invariant or differs from the general form in any coordinate
    long_order = rng.integers(0, len(order), 4000)
    hooks.sw29_chain(cid, ident, xy[long_order], rng.integers(0, 2, 4000))
    # k_accumulate lifts the first entry of a segment (Z = 1) instead of adding it to the identity
    for n in (1, 2, len(order)):
        for first_sign in (0, 1):
            sg = np.array(signs[:n])
            sg[0] = first_sign
            want = hooks.sw29_chain(cid, ident, xy[:n], sg)
            assert np.array_equal(canon(hooks.sw29_chain_lifted(cid, xy[:n], sg)), canon(want))
    hooks.sw29_chain_lifted(cid, xy[long_order], rng.integers(0, 2, 4000))


@pytest.mark.parametrize("name,max_v,b3,negative,reduce_b3,k_minus,inv", [
    ("bn254", 169, 9, False, False, 16, (5, 4, 1.6)),
    ("bls12-381", 2520, 12, False, False, 16, (2, 2, 1.1)),
    ("grumpkin", 169, 51, True, True, 8, (1.5, 2, 1.5)),
])
def test_bucket_accumulator_invariant_is_a_fixed_point(name, max_v, b3, negative, reduce_b3,
                                                        k_minus, inv):
    """curve/sw29.h add_mixed_acc drops partial reductions because k_accumulate's accumulator stays
    inside (V_X, V_Y, V_Z) <= P::acc_v.  Random chains never reach the worst case, so the bounds of
    every intermediate are propagated here with exact fractions, through the same steps, and every
    precondition of field/mont29.h is checked on the way: the invariant must map into itself."""
    from fractions import Fraction as Fr
    M = Fr(max_v)
    a, b, c = (Fr(v).limit_denominator(1000) for v in inv)

    def mul(va, vb):      # Montgomery product: (V_a V_b p^2 + m p) / R < (V_a V_b / max_v + 1) p
        assert va * vb / M + 1 < M
        return va * vb / M + 1

    def mul2(va, vb, vc, vd):
        assert (va * vb + vc * vd) / M + 1 < M
        return (va * vb + vc * vd) / M + 1

    def sub(k, va, vb):   # a + k p - b needs V_b < k - 0.01
        assert vb < k - Fr(1, 100)
        return va + k

    def mul_b3(v):
        assert b3 * v < M                 # mul_small must fit below max_v p
        return Fr(4) if reduce_b3 else b3 * v   # reduce: V < 4

    y2 = Fr(2)                            # y or 2 p - y
    t0 = mul(a, 1)
    t1 = mul(b, y2)
    t3 = sub(4, mul(1 + y2, a + b), t0 + t1)
    t4 = mul(y2, c) + b
    y3 = mul(1, c) + a
    t0x3 = 3 * t0
    u2, u3 = mul_b3(c), mul_b3(y3)
    plus, minus = t1 + u2, sub(k_minus, t1, u2)
    if not negative:
        z3, t1m = plus, minus
        t4n = sub(8, 0, t4)
        x = mul2(t3, t1m, t4n, u3)
        y = mul2(t1m, z3, u3, t0x3)
        z = mul2(z3, t4, t0x3, t3)
    else:
        z3, t1m = minus, plus
        t0n = sub(4, 0, t0x3)
        x = mul2(t3, t1m, t4, u3)
        y = mul2(t1m, z3, u3, t0n)
        z = mul2(z3, t4, t0x3, t3)
    assert x <= a and y <= b and z <= c, (name, float(x), float(y), float(z))
    assert max(x, y, z) < 6               # the general contract of a point handed on


def test_batched_inversion_tree_model():
    """k_prepare_addends_batched (msm/kernels.h) shares one inversion among the 1024 generators of
    a workgroup: lane prefix products, a binary product tree in heap order, the root inverted once,
    the inverse pushed back down (children get inv * sibling) and through the lanes' prefixes.
    The index logic, restated over integers mod p with the kernel's exact loop structure."""
    import random
    p = 2**255 - 19
    T, K = 256, 4
    rng = random.Random(5)
    for n in (1, 5, 255, 256, 1000, 1024):
        zs = [rng.randrange(1, p) for _ in range(n)]
        z = [[zs[j * T + t] if j * T + t < n else 1 for j in range(K)] for t in range(T)]
        prefix = [[0] * K for _ in range(T)]
        tree = [0] * (2 * T)
        for t in range(T):
            for j in range(K):
                prefix[t][j] = z[t][0] if j == 0 else prefix[t][j - 1] * z[t][j] % p
            tree[T + t] = prefix[t][K - 1]
        s = T // 2
        while s >= 1:
            for t in range(s):
                tree[s + t] = tree[2 * (s + t)] * tree[2 * (s + t) + 1] % p
            s >>= 1
        tree[1] = pow(tree[1], p - 2, p)
        s = 1
        while s < T:
            new = {}
            for t in range(s):
                i = s + t
                new[2 * i] = tree[i] * tree[2 * i + 1] % p
                new[2 * i + 1] = tree[i] * tree[2 * i] % p
            for k, v in new.items():
                tree[k] = v
            s <<= 1
        for t in range(T):
            inv = tree[T + t]
            for j in range(K - 1, -1, -1):
                zinv = inv if j == 0 else inv * prefix[t][j - 1] % p
                if j != 0:
                    inv = inv * z[t][j] % p
                i = j * T + t
                if i < n:
                    assert zinv * zs[i] % p == 1


def test_planner_window_tables():
    """msm/plan.h with a window table of 17 slices, `stride` rows apart: a column merges into ONE
    task of (windows - 1) * stride + n virtual rows when it is unsigned, fills at least half a
    slice, has more than one window and the cost model agrees (or merging is forced)"""
    stride = 1 << 18
    ns = [stride, stride - 100, stride, stride, stride // 4, 0, stride]
    widths = [256, 252, 8, 128, 256, 32, 32]
    signed = [0, 0, 0, 1, 0, 0, 0]
    per, totals = hooks.plan_tables(ns, widths, signed, stride, 17, force=True)
    # 256-bit: 17 windows of 16 bits in one task
    assert per[0].tolist()[:4] == [16, 17, 1, stride]
    assert int(per[0][4]) + (int(per[0][5]) << 32) == 16 * stride + ns[0]
    # 252-bit, ragged: 16 windows, rows up to the last window's end
    assert per[1].tolist()[:4] == [16, 16, 1, stride]
    assert int(per[1][4]) == 15 * stride + ns[1]
    assert per[2][2] == per[2][1] and per[2][3] == 0        # one window: nothing to merge
    assert per[3][3] == 0 and per[3][2] == per[3][1]        # signed: separate windows
    assert per[4][3] == 0                                   # short column: separate windows
    assert per[5].tolist()[:4] == [1, 0, 0, 0]              # empty column
    assert per[6].tolist()[:4] == [16, 3, 1, stride]        # 32-bit: 3 windows merged
    assert int(totals[3]) == 16 * stride + stride           # k_accumulate's grid: virtual rows
    assert int(totals[4]) == stride and int(totals[5]) == stride
    # the cost model: many long 256-bit columns merge (one bucket reduction instead of ~20, and
    # c = 16 becomes affordable); a single column does not when the table misses the cache
    many, _ = hooks.plan_tables([stride] * 8, [256] * 8, [0] * 8, stride, 17, table_penalty=1.03)
    assert all(row[2] == 1 and row[0] == 16 for row in many.tolist())
    one, _ = hooks.plan_tables([1 << 20], [252], [0], 1 << 20, 17, table_penalty=1.15)
    assert one[0][2] == one[0][1] == 16 and one[0][3] == 0
    # wide tables (bits > 16: 32-bit digits).  2^20 generators at 18 bits: 15 slices, a 256-bit
    # column is ONE task of 14 * stride + n < 2^24 virtual rows and 2^17 buckets, which the 32-bit
    # partition record still cuts into 1024 groups (sign + 7 bucket bits + 24 row bits)
    big = 1 << 20
    per, totals = hooks.plan_tables([big] * 8, [256] * 8, [0] * 8, big, 15, table_penalty=1.03, bits=18)
    assert all(row[:4] == [18, 15, 1, big] for row in per.tolist())
    assert int(totals[6]) == 1 and int(totals[7]) == 7 and int(totals[8]) == 1024
    assert int(totals[1]) == 8 << 17                       # one set of 2^17 buckets per column
    # 17 bits -- the DEFAULT of Weierstrass sets of 2^20 generators and more (api/state.h): 16 slices,
    # one task of 15 * stride + n virtual rows and 2^16 buckets per column, wide digits, 6 + 24 record bits
    per, totals = hooks.plan_tables([big] * 8, [256] * 8, [0] * 8, big, 16, table_penalty=1.15, bits=17)
    assert all(row[:4] == [17, 16, 1, big] for row in per.tolist())
    assert all(int(row[4]) + (int(row[5]) << 32) == 15 * big + big for row in per.tolist())
    assert int(totals[6]) == 1 and int(totals[7]) == 6 and int(totals[8]) == 1024
    assert int(totals[1]) == 8 << 16
    # ... and 2^18 generators at 20 bits: 13 slices, 2^19 buckets, 9 + 22 record bits, 1024 groups
    per, totals = hooks.plan_tables([1 << 18] * 8, [256] * 8, [0] * 8, 1 << 18, 13, force=True, bits=20)
    assert all(row[:4] == [20, 13, 1, 1 << 18] for row in per.tolist())
    assert int(totals[6]) == 1 and int(totals[7]) == 9 and int(totals[8]) == 1024
    # 16-bit tables keep int16 digits
    _, totals = hooks.plan_tables([big] * 8, [256] * 8, [0] * 8, big, 17, table_penalty=1.03)
    assert int(totals[6]) == 0

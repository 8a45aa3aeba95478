#!/usr/bin/env python3
"""Executable model of blitzar_amd/csrc/curve/sw_wave.h: the three Weierstrass base fields with ONE
field element spread over a DPP row of 16 lanes (limb j of the engine's unsaturated-limb form in
lane j, one more limb than the engine uses) and a projective point over the rows of a wavefront;
Montgomery products computed by all lanes of a row at once; the doubling and the complete addition of
Renes-Costello-Batina 2015 as rounds of four simultaneous products.

Every quantity carries a per-lane upper bound that is pushed through the same (monotone) operations
as the value, so one run shows that no 32-bit limb, 32-bit carry or 64-bit column sum of the kernel
overflows; the values are checked against plain big-integer arithmetic.  tests/test_sw_wave_model.py
runs it on the CPU.

    python tools/models/sw_wave_model.py

The lane program (this file is its specification; names as in sw_wave.h):

  product u * v / Rw mod p, Rw = 2^(LB NW), of a row (lane j < NW holds limb j; u is read by every lane
  as NW broadcast limbs, v is this lane's limb, zero in lanes >= NW):
    1  lo_j = sum_i u_i shr_i(v)_j         columns 0 .. NW-1        (shr_i / shl_i: DPP row shifts,
       hi_j = sum_{i>=1} u_i shl_(NW-i)(v)_j   columns NW .. 2NW-1   zero filled)
    2  l = lo mod Rw with limbs < 2^LB + 2: split lo into three LB-bit pieces, pass two of them up one
       and two lanes, one more round (what leaves lane NW-1 is dropped: multiples of Rw)
    3  d_j = sum_i n'_i shr_i(l)_j,  n' = -1/p mod Rw;  q = d mod Rw carried the same way
    4  lo'_j = sum_i p_i shr_i(q)_j,  hi'_j = sum_{i>=1} p_i shl_(NW-i)(q)_j
    5  the low half lo + lo' is k Rw for an integer k, which is read off its two top lanes:
       k = ceil((s_(NW-1) + s_(NW-2) / 2^LB) / 2^LB)  (the lanes below contribute < 2^-19)
    6  result = hi + hi' + k (into lane 0), carried to limbs < 2^LB + 2
"""
import random

L = 64
J = [l & 15 for l in range(L)]
ROW = [l >> 4 for l in range(L)]


class Field:
    def __init__(s, name, p, lb, n, b3, b3_negative):
        s.name, s.p, s.LB, s.N, s.NW = name, p, lb, n, n + 1
        s.mask = (1 << lb) - 1
        s.Rw = 1 << (lb * s.NW)
        s.R29 = 1 << (lb * n)
        s.b3, s.b3_negative = b3, b3_negative
        s.p_limbs = [(p >> (lb * i)) & s.mask for i in range(s.NW)]
        ninv = (-pow(p, -1, s.Rw)) % s.Rw
        s.ninv_limbs = [(ninv >> (lb * i)) & s.mask for i in range(s.NW)]
        s.bias_limbs, s.bias_value = wave_bias(p, lb, n)


def wave_bias(p, lb, n):
    """a multiple of p whose limbs 0 .. n-1 all lie in [4 2^lb, 5 2^lb): dominates, limb by limb,
    any sum of up to three carried elements.  (tools/gen_mont29_params.py emits the same table.)"""
    bmin = 4 << lb
    t0 = sum(bmin << (lb * j) for j in range(n))
    m = (t0 + p - 1) // p * p
    d = m - t0
    assert 0 <= d < p
    out = [bmin + ((d >> (lb * j)) & ((1 << lb) - 1)) for j in range(n)]
    assert d >> (lb * n) == 0 and sum(v << (lb * j) for j, v in enumerate(out)) == m
    return out + [0], m


FIELDS = {
    "bn254": Field("bn254", 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47, 29, 9, 9, False),
    "grumpkin": Field("grumpkin", 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001, 29, 9,
                      51, True),
    "bls12_381": Field(
        "bls12_381",
        0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
        28, 14, 12, False),
}


class V:
    """64 lanes: value and upper bound"""

    def __init__(s, val, bnd):
        s.val, s.bnd = list(val), list(bnd)


def const(xs):
    return V(xs, xs)


def zeros():
    return V([0] * L, [0] * L)


class Wave:
    def __init__(s, f):
        s.f = f
        s.live = [j < f.NW for j in J]  # lanes whose values and bounds are checked

    def chk(s, a, bits, what):
        for l in range(L):
            if s.live[l]:
                assert a.bnd[l] < (1 << bits), (what, l, a.bnd[l].bit_length())
                assert 0 <= a.val[l] <= a.bnd[l], (what, l)
        return a

    # DPP row shifts, zero filled.  shr: lane j receives lane j - i; shl: lane j receives lane j + i.
    # What a shr moves out of lane NW-1 is lost: `lossless` asserts that it is zero.
    def shr(s, a, i):
        def src(l):
            return l - i if J[l] - i >= 0 else None
        return V([0 if src(l) is None else a.val[src(l)] for l in range(L)],
                 [0 if src(l) is None else a.bnd[src(l)] for l in range(L)])

    def shl(s, a, i):
        def src(l):
            return l + i if J[l] + i <= 15 else None
        for l in range(L):  # the operand contract: nothing but zeros above lane NW-1
            if J[l] >= s.f.NW:
                assert a.val[l] == 0 and a.bnd[l] == 0, "shl of an unmasked operand"
        return V([0 if src(l) is None else a.val[src(l)] for l in range(L)],
                 [0 if src(l) is None else a.bnd[src(l)] for l in range(L)])

    def lanemask(s, a):
        return V([a.val[l] if s.live[l] else 0 for l in range(L)], [a.bnd[l] if s.live[l] else 0 for l in range(L)])

    def add(s, a, b, bits=32, what="add"):
        return s.chk(V([x + y for x, y in zip(a.val, b.val)], [x + y for x, y in zip(a.bnd, b.bnd)]), bits, what)

    def andc(s, a, m):
        return V([x & m for x in a.val], [min(b, m) for b in a.bnd])

    def shrc(s, a, k):
        return V([x >> k for x in a.val], [b >> k for b in a.bnd])

    def mad(s, a, b, c, what="mad"):
        return s.chk(V([x * y + z for x, y, z in zip(a.val, b.val, c.val)],
                       [x * y + z for x, y, z in zip(a.bnd, b.bnd, c.bnd)]), 64, what)

    def bcast(s, u, urow, i):
        """limb i of the row this lane's row reads its u operand from"""
        return V([u.val[urow[ROW[l]] * 16 + i] for l in range(L)], [u.bnd[urow[ROW[l]] * 16 + i] for l in range(L)])

    # ---- carries ----
    def carry3(s, x, what):
        """64-bit columns -> limbs < 2^LB + 2, value unchanged except for what leaves lane NW-1"""
        f = s.f
        m0 = s.andc(x, f.mask)
        m1 = s.andc(s.shrc(x, f.LB), f.mask)
        m2 = s.chk(s.shrc(x, 2 * f.LB), 32, what + " m2")
        a1 = s.add(s.add(m0, s.shr(m1, 1)), s.shr(m2, 2), 32, what + " round 1")
        return s.carry1(a1, what)

    def carry1(s, x, what, lossless=False):
        """one round: limbs < 2^LB + (largest carry).  `lossless`: the top lane provably carries
        nothing out (its BOUND is below 2^LB), so the value is unchanged"""
        f = s.f
        if lossless:
            for l in range(L):
                if J[l] == f.NW - 1:
                    assert x.bnd[l] >> f.LB == 0, (what, "the top lane may carry out", x.bnd[l])
        return s.add(s.andc(x, f.mask), s.shr(s.shrc(x, f.LB), 1), 32, what + " round 2")

    def scale(s, x, c, what):
        """c x carried: limbs < 2^LB + c + 1 (the top lane of x must be small enough to lose nothing)"""
        f = s.f
        wide = s.chk(V([v * c for v in x.val], [b * c for b in x.bnd]), 64, what)
        h = s.chk(s.shrc(wide, f.LB), 32, what + " carry")
        for l in range(L):
            if J[l] == f.NW - 1:
                assert h.bnd[l] == 0, (what, "top lane carries out")
        return s.add(s.andc(wide, f.mask), s.shr(h, 1), 32, what)

    def sub(s, a, b, what):
        """a + bias - b, bias dominating b limb by limb"""
        f = s.f
        bias = [f.bias_limbs[j] if j < f.NW else 0 for j in J]
        for l in range(L):
            if s.live[l]:
                assert bias[l] >= b.bnd[l], (what, "bias too small", J[l], b.bnd[l].bit_length())
        return s.chk(V([x + bi - y for x, y, bi in zip(a.val, b.val, bias)],
                       [x + bi for x, bi in zip(a.bnd, bias)]), 32, what)

    # ---- the product ----
    def mul(s, u, urow, v):
        f = s.f
        NW, LB, mask = f.NW, f.LB, f.mask
        v = s.lanemask(v)
        lo, hi = zeros(), zeros()
        for i in range(NW):
            ui = s.bcast(u, urow, i)
            lo = s.mad(ui, v if i == 0 else s.shr(v, i), lo, "lo")
            if i >= 1:
                hi = s.mad(ui, s.shl(v, NW - i), hi, "hi")
        l2 = s.carry3(lo, "l")
        d = zeros()
        for i in range(NW):
            d = s.mad(const([f.ninv_limbs[i]] * L), l2 if i == 0 else s.shr(l2, i), d, "d")
        q = s.lanemask(s.carry3(d, "q"))
        lo2, hi2 = zeros(), zeros()
        for i in range(f.N):  # p has N limbs
            pi = const([f.p_limbs[i]] * L)
            lo2 = s.mad(pi, q if i == 0 else s.shr(q, i), lo2, "lo'")
            if i >= 1:
                hi2 = s.mad(pi, s.shl(q, NW - i), hi2, "hi'")
        ssum = s.add(lo, lo2, 64, "s")
        # the low half is an exact multiple of Rw (checked on the values)
        for r in range(4):
            low = sum(ssum.val[r * 16 + j] << (LB * j) for j in range(NW))
            assert low % f.Rw == 0
        sp = s.shr(ssum, 1)
        a = s.shrc(sp, LB)
        b = s.andc(sp, mask)
        t = s.add(ssum, a, 64, "t")
        m = s.shrc(t, LB)
        r_ = s.andc(t, mask)
        k = V([mv + (1 if (rv | bv) != 0 else 0) for mv, rv, bv in zip(m.val, r_.val, b.val)], [mb + 1 for mb in m.bnd])
        for r in range(4):
            low = sum(ssum.val[r * 16 + j] << (LB * j) for j in range(NW))
            assert k.val[r * 16 + NW - 1] == low // f.Rw, "k"
        kk = V([k.val[l] if J[l] == NW - 1 else 0 for l in range(L)], [k.bnd[l] if J[l] == NW - 1 else 0 for l in range(L)])
        total = s.add(s.add(hi, hi2, 64, "hi + hi'"), s.shl(kk, NW - 1), 64, "hi + hi' + k")
        # nothing may leave the top lane here: the result is far below Rw
        m2 = s.shrc(total, 2 * LB)
        m1 = s.andc(s.shrc(total, LB), mask)
        for l in range(L):
            if J[l] == NW - 1:
                assert m1.val[l] == 0 and m2.val[l] == 0
            if J[l] == NW - 2:
                assert m2.val[l] == 0
        out = s.carry3(total, "result")
        # value bound: (T + Q p) / Rw < U V / Rw + (1 + 2^-20) p with U, V the operands' largest values;
        # limbs are non-negative, so limb j <= value >> (LB j): that bounds the top lanes
        for r in range(4):
            ub = sum(u.bnd[urow[r] * 16 + j] << (LB * j) for j in range(NW))
            vb = sum(v.bnd[r * 16 + j] << (LB * j) for j in range(NW))
            res = ub * vb // f.Rw + f.p + (f.p >> 20) + 1
            assert s.value(out, r) < res
            assert res < f.R29, "a product's result must fit the engine's N limbs"
            for j in range(NW):
                out.bnd[r * 16 + j] = min(out.bnd[r * 16 + j], res >> (LB * j))
        return s.lanemask(out)

    # ---- values ----
    def value(s, a, row):
        return sum(a.val[row * 16 + j] << (s.f.LB * j) for j in range(s.f.NW))

    def from_rows(s, xs):
        f = s.f
        vals, bnds = [], []
        for r in range(4):
            x = xs[r]
            for j in range(16):
                if j < f.NW:
                    limb = (x >> (f.LB * j)) & f.mask if j < f.NW - 1 else x >> (f.LB * j)
                else:
                    limb = 0
                vals.append(limb)
                bnds.append(0 if j >= f.NW else (limb if j == f.NW - 1 else max(limb, f.mask + 2)))
        return V(vals, bnds)

    def exchange(s, a):
        """limb j of all four rows in every lane: e[r] in lane (row, j) = a in lane (r, j)"""
        return [V([a.val[r * 16 + J[l]] for l in range(L)], [a.bnd[r * 16 + J[l]] for l in range(L)]) for r in range(4)]

    def by_row(s, a0, a1, a2, a3):
        src = [a0, a1, a2, a3]
        return V([src[ROW[l]].val[l] for l in range(L)], [src[ROW[l]].bnd[l] for l in range(L)])


#------------------------------------------------------------------------------------------------
# group law: state = lane (row, j) holds limb j of X | Y | Z (row 3 unused)
#------------------------------------------------------------------------------------------------
def dbl(w, st):
    """RCB15 Alg. 9 (a = 0) as 2 rounds of 4 products"""
    f = w.f
    # round 1: Y Y | Y Z | Z Z | X Y
    e = w.exchange(st)
    x, y, z = e[0], e[1], e[2]
    m = w.mul(st, [1, 1, 2, 0], w.by_row(y, z, z, y))
    t0, t1, zz, xy = w.exchange(m)
    z3 = w.scale(t0, 8, "8 t0")
    ub = w.scale(zz, f.b3, "|3b| zz")
    ub3 = w.scale(zz, 3 * f.b3, "3 |3b| zz")
    xy2 = w.add(xy, xy)
    if not f.b3_negative:
        y3 = w.add(t0, ub)
        t0m = w.carry1(w.sub(t0, ub3, "t0 - 3u"), "t0m", True)
    else:
        y3 = w.carry1(w.sub(t0, ub, "t0 - u"), "y3", True)
        t0m = w.add(t0, ub3)
    # round 2: t1 z3 (Z3) | u z3 | t0m y3 | t0m 2xy (X3)
    u2 = w.by_row(t1, ub, t0m, t0m)
    h = w.mul(u2, [0, 1, 2, 3], w.by_row(z3, z3, y3, xy2))
    h0, h1, h2, h3 = w.exchange(h)
    if not f.b3_negative:
        ny = w.carry1(w.add(h1, h2), "Y3", True)
    else:
        ny = w.carry1(w.sub(h2, h1, "Y3"), "Y3", True)
    return w.by_row(h3, ny, h0, h0)


def add(w, st, q):
    """RCB15 Alg. 7 (a = 0) as 3 rounds of 4 products; q: the second point in the same layout"""
    f = w.f
    x1, y1, z1, _ = w.exchange(st)
    x2, y2, z2, _ = w.exchange(q)
    # round 1: X1 X2 | Y1 Y2 | Z1 Z2 | (X1 + Y1)(X2 + Y2)
    u1 = w.by_row(x1, y1, z1, w.add(x1, y1))
    v1 = w.by_row(x2, y2, z2, w.add(x2, y2))
    t0, t1, t2, t3p = w.exchange(w.mul(u1, [0, 1, 2, 3], v1))
    ub = w.scale(t2, f.b3, "|3b| t2")
    t00 = w.add(w.add(t0, t0), t0)
    t3 = w.carry1(w.sub(t3p, w.add(t0, t1), "t3"), "t3", True)
    if not f.b3_negative:
        z3 = w.add(t1, ub)
        t1m = w.carry1(w.sub(t1, ub, "t1 - u"), "t1m", True)
    else:
        z3 = w.carry1(w.sub(t1, ub, "t1 - u"), "z3", True)
        t1m = w.add(t1, ub)
    # round 2: (Y1 + Z1)(Y2 + Z2) | (X1 + Z1)(X2 + Z2) | t1m z3 | 3 t0 t3
    u2 = w.by_row(w.add(y1, z1), w.add(x1, z1), t1m, t00)
    v2 = w.by_row(w.add(y2, z2), w.add(x2, z2), z3, t3)
    t4p, t5p, mm, nn = w.exchange(w.mul(u2, [0, 1, 2, 3], v2))
    t4 = w.carry1(w.sub(t4p, w.add(t1, t2), "t4"), "t4", True)
    t5 = w.carry1(w.sub(t5p, w.add(t0, t2), "t5"), "t5", True)
    y3 = w.scale(t5, f.b3, "|3b| t5")
    # round 3: t3 t1m | t4 y3 | y3 3t0 | z3 t4
    u3 = w.by_row(t3, t4, y3, z3)
    v3 = w.by_row(t1m, y3, t00, t4)
    p1, p2, p3, p4 = w.exchange(w.mul(u3, [0, 1, 2, 3], v3))
    if not f.b3_negative:
        nx = w.carry1(w.sub(p1, p2, "X3"), "X3", True)   # t3 t1m - t4 y3
        ny = w.carry1(w.add(mm, p3), "Y3", True)         # t1m z3 + y3 3t0
    else:
        nx = w.carry1(w.add(p1, p2), "X3", True)         # y3 = -|..|: t3 t1m + t4 |y3|
        ny = w.carry1(w.sub(mm, p3, "Y3"), "Y3", True)
    nz = w.carry1(w.add(p4, nn), "Z3", True)
    return w.by_row(nx, ny, nz, nz)


def renorm(w, st):
    """every coordinate times the Montgomery one of the engine's form... not needed: a last product
    by Rw mod p brings V to ~1 and the limbs to < 2^LB + 2 with a zero top lane"""
    f = w.f
    one = w.from_rows([f.Rw % f.p] * 4)
    return w.mul(st, [0, 1, 2, 3], one)


#------------------------------------------------------------------------------------------------
# reference arithmetic (plain integers, projective RCB15 on values mod p)
#------------------------------------------------------------------------------------------------
def ref_b3(f):
    return (-f.b3 if f.b3_negative else f.b3) % f.p


def ref_add(f, P1, P2):
    p, b3 = f.p, ref_b3(f)
    X1, Y1, Z1 = P1
    X2, Y2, Z2 = P2
    t0, t1, t2 = X1 * X2 % p, Y1 * Y2 % p, Z1 * Z2 % p
    t3 = ((X1 + Y1) * (X2 + Y2) - t0 - t1) % p
    t4 = ((Y1 + Z1) * (Y2 + Z2) - t1 - t2) % p
    y3 = b3 * (((X1 + Z1) * (X2 + Z2) - t0 - t2) % p) % p
    t00 = 3 * t0 % p
    u = b3 * t2 % p
    z3, t1m = (t1 + u) % p, (t1 - u) % p
    return ((t3 * t1m - t4 * y3) % p, (t1m * z3 + y3 * t00) % p, (z3 * t4 + t00 * t3) % p)


def ref_dbl(f, P1):
    return ref_add(f, P1, P1)


def same_point(f, A, B):
    p = f.p
    return all((A[i] * B[j] - A[j] * B[i]) % p == 0 for i in range(3) for j in range(3))


def sqrt_mod(a, p, rng):
    """Tonelli-Shanks (grumpkin's base field is 1 mod 4 with 2-adicity 28); None for a non-residue"""
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    if p % 4 == 3:
        return pow(a, (p + 1) // 4, p)
    q, e = p - 1, 0
    while q % 2 == 0:
        q //= 2
        e += 1
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, t, r = e, pow(z, q, p), pow(a, q, p), pow(a, (q + 1) // 2, p)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % p
            i += 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c = i, b * b % p
        t, r = t * c % p, r * b % p
    return r


def random_point(f, rng):
    """a point of y^2 = x^3 + b in projective coordinates with a random Z"""
    p = f.p
    b = ref_b3(f) * pow(3, -1, p) % p
    while True:
        x = rng.randrange(p)
        rhs = (x * x * x + b) % p
        y = sqrt_mod(rhs, p, rng)
        if y is not None:
            assert y * y % p == rhs
            zs = rng.randrange(1, p)
            return (x * zs % p, y * zs % p, zs)


#------------------------------------------------------------------------------------------------
# checks
#------------------------------------------------------------------------------------------------
def check_field_products(name, rounds=60):
    f = FIELDS[name]
    w = Wave(f)
    rng = random.Random(7)
    rinv = pow(f.Rw, -1, f.p)
    worst = 0
    for it in range(rounds):
        if it % 3 == 0:    # carried elements
            a = [rng.randrange(f.p) for _ in range(4)]
            b = [rng.randrange(f.p) for _ in range(4)]
            u, v = w.from_rows(a), w.from_rows(b)
        else:              # limbs at the edge of what the formulas hand over: < 2^(LB+1) + 2^9 each
            top = (2 << f.LB) + 512
            bnd = [top if j < f.N else (8 if j < f.NW else 0) for j in J]
            u = V([rng.choice([b, rng.randrange(b + 1)]) for b in bnd], bnd)
            v = V([rng.choice([b, rng.randrange(b + 1)]) for b in bnd], bnd)
            if it % 3 == 2:  # a subtraction's output: limbs up to 6 2^LB in the lanes the bias covers
                top6 = 6 << f.LB
                u = V([top6 if j < f.N else (8 if j < f.NW else 0) for j in J], [top6 if j < f.N else (8 if j < f.NW else 0) for j in J])
                u = w.carry1(u, "edge")
            a = [w.value(u, r) for r in range(4)]
            b = [w.value(v, r) for r in range(4)]
        z = w.mul(u, [0, 1, 2, 3], v)
        for r in range(4):
            got = w.value(z, r)
            assert got % f.p == a[r] * b[r] * rinv % f.p, (name, it, r)
            assert got < a[r] * b[r] // f.Rw + f.p + (f.p >> 10)
            worst = max(worst, max(z.val[r * 16 + j] for j in range(f.NW)))
        assert max(z.bnd[l] for l in range(L) if J[l] < f.NW) <= f.mask + 3
    return worst / (f.mask + 1)


def check_point_chain(name, doublings=40, seed=3):
    f = FIELDS[name]
    w = Wave(f)
    rng = random.Random(seed)
    P1 = random_point(f, rng)
    P2 = random_point(f, rng)
    # the engine's form: a R29 mod p, carried limbs; the wave reads the same limbs as (a R29 / Rw) Rw:
    # every coordinate of a point scaled by the same factor, which a projective point does not notice
    def to_wave(P):
        return w.from_rows([c * f.R29 % f.p for c in P] + [0])
    def to_ref(st):
        rinv = pow(f.Rw, -1, f.p)
        return tuple(w.value(st, r) * rinv % f.p for r in range(3))
    st, q = to_wave(P1), to_wave(P2)
    ref, refq = P1, P2
    worst_limb = 0
    for k in range(doublings):
        st = dbl(w, st)
        ref = ref_dbl(f, ref)
        assert same_point(f, to_ref(st), ref), (name, "dbl", k)
        if k % 8 == 7:
            st = add(w, st, q)
            ref = ref_add(f, ref, refq)
            assert same_point(f, to_ref(st), ref), (name, "add", k)
        worst_limb = max(worst_limb, max(st.bnd[l] for l in range(L) if J[l] < f.NW and ROW[l] < 3))
    # special cases of the complete formulas: identity operands, P + P, P - P
    ident = w.from_rows([0, f.R29 % f.p, 0, 0])
    assert same_point(f, to_ref(add(w, st, ident)), ref)
    assert same_point(f, to_ref(add(w, ident, q)), refq)
    assert same_point(f, to_ref(add(w, q, q)), ref_dbl(f, refq))
    neg = to_wave((P2[0], (-P2[1]) % f.p, P2[2]))
    zero = to_ref(add(w, q, neg))
    assert zero[0] == 0 and zero[2] == 0 and zero[1] != 0
    assert same_point(f, to_ref(dbl(w, ident)), (0, 1, 0))
    out = renorm(w, st)
    for r in range(3):
        assert w.value(out, r) < f.p + (f.p >> 10)
        assert out.val[r * 16 + f.NW - 1] == 0
    assert same_point(f, to_ref(out), ref)
    return worst_limb / (f.mask + 1)


def main():
    for name in FIELDS:
        print(name, "product limbs / 2^LB <=", check_field_products(name))
        print(name, "state limbs / 2^LB <=", check_point_chain(name))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Executable model of blitzar_amd/csrc/curve/ed16_wave.h: GF(2^255-19) arithmetic with ONE field
element spread over the 16 lanes of a DPP row (16-bit limbs) and a point over the 4 rows of a
wavefront.  Every quantity carries a per-lane upper bound that is pushed through the same
(monotone) operations as the value, so one run proves that no 24-bit multiplier operand, 32-bit
limb or 64-bit column sum of the kernel overflows; the values are checked against plain
big-integer edwards25519 arithmetic.  tests/test_ed16_wave_model.py runs it on the CPU.

    python tools/models/ed16_wave_model.py
"""
import random
P = 2**255 - 19
L = 64
J = [l & 15 for l in range(L)]
ROW = [l >> 4 for l in range(L)]
M38 = [38 if j == 0 else 1 for j in J]

class V:
    def __init__(s, val, bnd): s.val, s.bnd = list(val), list(bnd)
def const(xs): return V(xs, xs)
def ror1(a): return V([a.val[(l & 48) | ((l - 1) & 15)] for l in range(L)], [a.bnd[(l & 48) | ((l - 1) & 15)] for l in range(L)])
def bcast(a, i): return V([a.val[(l & 48) | i] for l in range(L)], [a.bnd[(l & 48) | i] for l in range(L)])
def chk(a, bits, what):
    assert max(a.bnd) < 2**bits, (what, max(a.bnd).bit_length())
    assert all(v <= b for v, b in zip(a.val, a.bnd)), what
    return a
def add(a, b, bits=32): return chk(V([x + y for x, y in zip(a.val, b.val)], [x + y for x, y in zip(a.bnd, b.bnd)]), bits, 'add')
def mulc(a, c, bits=32): return chk(V([x * y for x, y in zip(a.val, c)], [x * y for x, y in zip(a.bnd, c)]), bits, 'mulc')
def mul24(a, c):  # v_mul_u32_u24
    chk(a, 24, 'mul24 in'); return mulc(a, c, 32)
def sub(a, b, bias):  # a + bias - b, bias per-lane constants >= bnd(b)
    assert all(bi >= bb for bi, bb in zip(bias, b.bnd)), 'bias too small: %s' % max(bb - bi for bi, bb in zip(bias, b.bnd))
    return chk(V([x + bi - y for x, y, bi in zip(a.val, b.val, bias)], [x + bi for x, bi in zip(a.bnd, bias)]), 32, 'sub')
def andc(a, m): return V([x & m for x in a.val], [min(b, m) for b in a.bnd])
def shr(a, k): return V([x >> k for x in a.val], [b >> k for b in a.bnd])
def mad(a, b, c, bits):  # a*b + c
    return chk(V([x * y + z for x, y, z in zip(a.val, b.val, c.val)], [x * y + z for x, y, z in zip(a.bnd, b.bnd, c.bnd)]), bits, 'mad')

def ror(a, i):  # DPP row_ror:i -- lane j of a row receives lane j - i
    return V([a.val[(l & 48) | ((l - i) & 15)] for l in range(L)], [a.bnd[(l & 48) | ((l - i) & 15)] for l in range(L)])
def fmul(u, v):
    """column j of u * v mod p in lane j: sum_i u_i * v'_(j-i), v' = v rotated by i lanes with the
    wrapped lanes (j < i) multiplied by 38 -- ed16w::rotations + mul_columns"""
    chk(u, 32, 'u'); chk(v, 24, 'v')
    acc = V([0] * L, [0] * L)
    for i in range(16):
        vs = v if i == 0 else mul24(ror(v, i), [38 if j < i else 1 for j in J])
        acc = mad(bcast(u, i), vs, acc, 64)
    chk(acc, 48, 'acc')
    lo, hi = andc(acc, 0xffff), chk(shr(acc, 16), 32, 'hi')
    x = mad(ror1(hi), const(M38), lo, 64)
    lo2, hi2 = andc(x, 0xffff), chk(shr(x, 16), 24, 'hi2')
    y = add(lo2, mul24(ror1(hi2), M38))
    lo3, hi3 = andc(y, 0xffff), shr(y, 16)
    z = add(lo3, mul24(ror1(hi3), M38))
    return z

def value(a, row): return sum(a.val[row * 16 + j] << (16 * j) for j in range(16)) % P
def limbs_of(x, n=16): return [(x >> (16 * j)) & 0xffff for j in range(n)]
def multiple_of_p(k):  # limb-wise k*p (no carries): every limb ~ k * 2^16 except the top ~ k * 2^15
    pl = limbs_of(P)
    return [k * pl[j] for j in J]

def from_rows(xs):  # 4 field values -> distributed
    vals = []
    for r in range(4): vals += limbs_of(xs[r] % 2**256)
    return V(vals, [0xffff] * L)

def check_field_products():
    random.seed(1)
    R = None
    for it in range(200):
        a = [random.randrange(2**256) for _ in range(4)]; b = [random.randrange(2**256) for _ in range(4)]
        u, v = from_rows(a), from_rows(b)
        # inflate bounds: u < 2^19.3, v < 2^18.7
        u.bnd = [int(2**19.3)] * L; v.bnd = [int(2**18.7)] * L
        if it % 2:
            u.val = [random.randrange(b + 1) for b in u.bnd]; v.val = [random.randrange(b + 1) for b in v.bnd]
            a = [value(u, r) for r in range(4)]; b = [value(v, r) for r in range(4)]
        z = fmul(u, v)
        for r in range(4): assert value(z, r) == a[r] * b[r] % P
        R = z.bnd
    import math
    assert max(R) < 2**16 + 64  # the bound ed16_wave.h states for product limbs
    return math.log2(R[0]), math.log2(max(R[1:16]))

# ---- point arithmetic on the distributed state (rows X, Y, Z, T) ----
D = (-121665 * pow(121666, P - 2, P)) % P
def xch(a): return [V([a.val[(r * 16) | j] for j in J], [a.bnd[(r * 16) | j] for j in J]) for r in range(4)]
def select(a, b, c, d):
    src = [a, b, c, d]
    return V([src[ROW[l]].val[l] for l in range(L)], [src[ROW[l]].bnd[l] for l in range(L)])
B3, B5 = multiple_of_p(3), multiple_of_p(5)
def dbl(acc):
    X, Y, Z, T = xch(acc)
    A, B, ZZ, M = xch(fmul(select(X, Y, Z, X), select(X, Y, Z, Y)))
    Hp = add(A, B)
    E = add(M, M)
    G = sub(B, A, B3)
    Fp = sub(add(A, add(ZZ, ZZ)), B, B3)
    return fmul(select(E, G, Fp, E), select(Fp, Hp, G, Hp))
def add_cached(acc, cw):
    X, Y, Z, T = xch(acc)
    u = select(add(Y, X), sub(Y, X, B3), Z, T)
    a, b, zz, c = xch(fmul(u, cw))
    d = add(zz, zz)
    ez, et, ex, ey = add(d, c), sub(d, c, B3), sub(a, b, B3), add(a, b)
    return fmul(select(ex, ey, ez, ex), select(et, ez, et, ey))

# reference
def e_add(p, q):
    x1, y1, z1, t1 = p; x2, y2, z2, t2 = q
    a = (y1 - x1) * (y2 - x2) % P; b = (y1 + x1) * (y2 + x2) % P; c = t1 * 2 * D * t2 % P; d = z1 * 2 * z2 % P
    e, f, g, h = b - a, d - c, d + c, b + a
    return (e * f % P, g * h % P, f * g % P, e * h % P)
def affine(p): zi = pow(p[2], P - 2, P); return (p[0] * zi % P, p[1] * zi % P)
def cached_rows(q):
    x, y, z, t = q
    return from_rows([(y + x) % P, (y - x) % P, z % P, t * 2 * D % P])

def check_point_chain():
    by = 4 * pow(5, P - 2, P) % P
    bx = 15112221349535400772501151409588531511454012693041857206046113283949847762202
    Bp = (bx, by, 1, bx * by % P)
    ident = (0, 1, 1, 0)
    acc_ref = ident
    acc = from_rows([0, 1, 1, 0])
    q = Bp
    for it in range(6):
        acc = add_cached(acc, cached_rows(q)); acc_ref = e_add(acc_ref, q)
        for k in range(16):
            acc = dbl(acc); acc_ref = e_add(acc_ref, acc_ref)
        got = tuple(value(acc, r) for r in range(4))
        assert affine(got) == affine(acc_ref), it
        assert got[0] * got[1] % P == got[2] * got[3] % P
        q = e_add(q, e_add(Bp, Bp))
    import math
    return math.log2(max(acc.bnd))


if __name__ == '__main__':
    print('field products ok; limb bounds (bits): limb 0 %.3f, others %.3f' % check_field_products())
    print('6 x (add, 16 doublings) ok; state limb bound %.4f bits' % check_point_chain())

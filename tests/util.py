"""Shared builders for parity tests (inputs only; expected values always come from the oracle)."""
import numpy as np

from oracle import ref_oracle


def mt_bytes(seed, n):
    """byte stream of std::mt19937{seed} through libstdc++'s uniform_int_distribution<uint8_t>
    (the reference benchmarks' input recipe); see tools/baseline_workloads.py"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "tools"))
    import baseline_workloads
    return baseline_workloads.mt19937_bytes(n, seed)


def weierstrass_generators(curve_id, n, distinct_seeds=64):
    """n affine generators in C-ABI layout: the first `distinct_seeds` come from the reference's
    generate_random_element(rng{i+1, i+2}); the rest continue g_i = g_{i-1} + g_0 (the recipe of
    SURVEY 8(d) for big generator sets).  Index 5 (if present) is the identity."""
    _, nl, stride, _ = ref_oracle.CURVES[curve_id]
    out = np.zeros((n, stride), dtype=np.uint8)
    k = min(n, distinct_seeds)
    for i in range(k):
        out[i] = ref_oracle.random_affine(curve_id, i + 1, i + 2)
    if n > k:
        one = ref_oracle.identity_affine(curve_id)[8 * nl:16 * nl].view(np.uint64)

        def p2(a):
            return np.concatenate([a[:16 * nl].view(np.uint64), one])
        g0 = p2(out[0])
        acc = p2(out[k - 1])
        for i in range(k, n):
            acc = ref_oracle.add_projective(curve_id, acc, g0)
            out[i] = ref_oracle.to_affine(curve_id, acc)
            acc = p2(out[i])
    if n > 5:
        out[5] = ref_oracle.identity_affine(curve_id)
    return out


def weierstrass_generators_big(curve_id, n, distinct_seeds=1024, threads=None):
    """the same recipe as weierstrass_generators (and the same bytes), for sets of 2^18 .. 2^22
    points: the chain runs inside the oracle library (ref_<curve>_generator_chain), cut into
    segments on host threads -- segment j starts at g_{k-1} + (j L) g_0, formed with the reference's
    own double / add; affine coordinates do not depend on how a point was reached"""
    import concurrent.futures
    import os
    _, nl, stride, _ = ref_oracle.CURVES[curve_id]
    out = np.zeros((n, stride), dtype=np.uint8)
    k = min(n, distinct_seeds)
    for i in range(k):
        out[i] = ref_oracle.random_affine(curve_id, i + 1, i + 2)
    if n > k:
        one = ref_oracle.identity_affine(curve_id)[8 * nl:16 * nl].view(np.uint64)

        def p2(a):
            return np.concatenate([a[:16 * nl].view(np.uint64), one])
        g0, last = p2(out[0]), p2(out[k - 1])

        def multiple(m):  # m * g0, m >= 1
            acc = None
            for bit in bin(m)[2:]:
                if acc is not None:
                    acc = ref_oracle.double_projective(curve_id, acc)
                if bit == "1":
                    acc = g0 if acc is None else ref_oracle.add_projective(curve_id, acc, g0)
            return acc
        threads = threads or min(32, os.cpu_count() or 1)
        rest = n - k
        seg = (rest + threads - 1) // threads
        jobs = []
        for j in range(0, rest, seg):
            start = last if j == 0 else ref_oracle.add_projective(curve_id, last, multiple(j))
            jobs.append((k + j, start, min(seg, rest - j)))
        with concurrent.futures.ThreadPoolExecutor(max_workers=threads) as ex:
            for (at, _, count), (piece, _) in zip(jobs, ex.map(
                    lambda job: ref_oracle.generator_chain(curve_id, job[1], g0, job[2]), jobs)):
                out[at:at + count] = piece
    if n > 5:
        out[5] = ref_oracle.identity_affine(curve_id)
    return out


def generators_for(curve_id, n):
    if curve_id == 0:
        return ref_oracle.ristretto_generators(n)
    return weierstrass_generators(curve_id, n)


def api_generators(curve_id, gens):
    """oracle-side generator array -> uint8 C-ABI view for blitzar_amd.api"""
    return np.ascontiguousarray(gens).view(np.uint8).reshape(gens.shape[0], -1)


def mixed_columns(rng, n, include_signed=True):
    """columns covering the reference exerciser's cases (sxt/multiexp/test/multiexponentiation.cc:
    42-451): every width 1..32, unequal lengths, empty, zeros, all-ones, signed extremes."""
    cols = []
    for nb in (1, 2, 3, 4, 5, 8, 13, 16, 24, 31, 32):
        m = int(rng.integers(0, n + 1))
        cols.append((rng.integers(0, 256, (m, nb), dtype=np.uint8), False))
    if include_signed:
        for nb in (1, 2, 4, 8, 16):
            m = int(rng.integers(1, n + 1))
            cols.append((rng.integers(0, 256, (m, nb), dtype=np.uint8), True))
        cols.append((np.array([-128, 127, -1, 0, 1][:n], dtype=np.int8), True))
        cols.append((np.array([-2**63, 2**63 - 1, -1][:n], dtype=np.int64), True))
    cols.append((np.full((n, 32), 0xff, dtype=np.uint8), False))      # 2^256 - 1 everywhere
    cols.append((np.zeros((n, 8), dtype=np.uint8), False))            # all-zero sequence
    cols.append((np.zeros((0, 4), dtype=np.uint8), False))            # empty sequence
    cols.append((np.array([0, 1, 2, 3][:n], dtype=np.uint64), False))
    cols.append((np.array([2**64 - 1] * min(n, 3), dtype=np.uint64), False))
    return cols

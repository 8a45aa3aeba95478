"""Synthetic inputs of BASELINE.json's five configs, shared by bench.py and
tests/test_gpu_baseline_configs.py.  Input generation only -- expected values always come from the
oracle module the caller passes in (tests/ and bench.py's cpu_baseline leg; this file itself
imports nothing from oracle/).

  * byte streams: `std::mt19937{0}` through libstdc++'s uniform_int_distribution<uint8_t>
    (benchmark/multi_commitment/benchmark.m.cc:141-156): Lemire's method on a 32-bit engine keeps
    the top 8 bits of every draw.  numpy's legacy MT19937 is the same engine (2^21 .. 2^27 draws for
    configs 1-3: <= 4 s).  Configs 4 and 5 need 2^33 and 3.3e9 draws of that ONE serial stream (215 s
    / 83 s through numpy): tools/mt19937/mtstream.cc produces the same bytes on all host threads at
    once, thread j starting from the generator state after j L draws (polynomial jump-ahead), and
    is used whenever it has been built (__graft_entry__.build()).
  * generator sets for the three Weierstrass curves: SURVEY 8(d)'s recipe, g_0 = the reference's
    generate_random_element(rng{1, 2}), g_i = g_{i-1} + g_0, built by the reference's own code on
    host threads (reference_generators; round 3 built the same points with the product's
    bzamd_generator_multiples_device, which is now only cross-checked against them).  The set is
    g_i = (i + 1) g_0, so  sum_i a_i g_i = (sum_i a_i (i + 1) mod r) g_0, one
    scalar multiplication with the reference's own curve operations: a full-size parity check of
    EVERY output that costs milliseconds per output.
"""
import ctypes

import numpy as np

# group orders (curve ids of cbindings/blitzar_api.h:28-31)
ORDER = {
    1: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    2: 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    3: 21888242871839275222246405745257275088696311157297823662689037894645226208583,
}


_MTSTREAM = None


def mtstream_lib():
    """tools/mt19937/_build/libmtstream.so (None when it has not been built)"""
    global _MTSTREAM
    if _MTSTREAM is None:
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mt19937", "_build",
                            "libmtstream.so")
        _MTSTREAM = False
        if os.path.exists(path):
            lib = ctypes.CDLL(path)
            lib.mt19937_fill.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32,
                                         ctypes.c_uint64, ctypes.c_uint, ctypes.c_uint]
            lib.mt19937_fill.restype = ctypes.c_int
            _MTSTREAM = lib
    return _MTSTREAM or None


def mt19937_bytes(count, seed=0, boolean=False, skip=0, threads=0, force_numpy=False):
    """draws [skip, skip + count) of uniform_int_distribution<uint8_t>{0, 255} (or {0, 1}) on
    std::mt19937{seed}"""
    lib = None if force_numpy else mtstream_lib()
    if lib is not None:
        out = np.empty(count, dtype=np.uint8)
        lib.mt19937_fill(out.ctypes.data, count, seed, skip, threads, 31 if boolean else 24)
        return out
    assert skip == 0, "a stream offset needs tools/mt19937 (python __graft_entry__.py)"
    bg = np.random.MT19937()
    bg._legacy_seeding(seed)
    out = np.empty(count, dtype=np.uint8)
    chunk = 1 << 24
    for lo in range(0, count, chunk):
        m = min(chunk, count - lo)
        raw = bg.random_raw(m)  # the engine's 32-bit outputs, in uint64
        out[lo:lo + m] = (raw >> (31 if boolean else 24)).astype(np.uint8)
    return out


def mt19937_scalars(columns, n, nbytes, top_mask=0xff, seed=0, first_column=0):
    """[columns, n, nbytes] little-endian scalars, column-major fill of one stream (column c occupies
    bytes [c n nbytes, (c + 1) n nbytes), multi_commitment/benchmark.m.cc:141-156); the top byte of
    every scalar is masked with `top_mask` (0x0f: uniform 252-bit values).  `first_column`: the
    columns [first_column, first_column + columns) of that stream (a rank's shard)"""
    s = mt19937_bytes(columns * n * nbytes, seed,
                      skip=first_column * n * nbytes).reshape(columns, n, nbytes)
    if top_mask != 0xff:
        s[:, :, nbytes - 1] &= top_mask
    return s


def vp(t):
    return ctypes.c_void_p(t.data_ptr())


def dlog_generators(lib, oracle, cid, n, dev, stream, check_device_kernel=True):
    """(base affine bytes, device tensor [n, stride]) with generators[i] = (i + 1) * base: SURVEY
    8(d)'s recipe (g_0 = generate_random_element(rng{1, 2}), g_i = g_{i-1} + g_0), built on the host
    by the reference's own code (reference_generators) and uploaded.  The product's own generator
    kernel (bzamd_generator_multiples_device, a convenience for callers without the reference) must
    produce the same bytes: checked here at full size."""
    import torch
    from blitzar_amd import api
    base, host = reference_generators(oracle, cid, n)
    gens = torch.from_numpy(host).to(dev)
    if check_device_kernel:
        d_base = torch.from_numpy(base.copy()).to(dev)
        mine = torch.empty((n, api.CURVE_LAYOUT[cid][0]), dtype=torch.uint8, device=dev)
        lib.bzamd_generator_multiples_device(cid, vp(mine), vp(d_base), n, stream)
        torch.cuda.synchronize()
        assert torch.equal(mine, gens), "bzamd_generator_multiples_device differs from the reference chain"
        del mine
    torch.cuda.synchronize()
    return base, gens


def reference_generators(oracle, cid, n, threads=None):
    """SURVEY 8(d)'s generator recipe for configs 3-5, built by the REFERENCE's own code (oracle/_ref:
    generate_random_element, add, to_element_affine): g_0 = generate_random_element(rng{1, 2}),
    g_i = g_{i-1} + g_0.  Returns (g_0 affine bytes, host array [n, stride] in C-ABI layout); the set
    is g_i = (i + 1) g_0, so every commitment still has a closed form.  The chain runs inside the
    oracle library on host threads, segment j starting from (j L) g_0 formed by doubling."""
    import concurrent.futures
    import os
    _, nl, stride, _ = oracle.CURVES[cid]
    base = oracle.random_affine(cid, 1, 2)
    g0 = oracle.affine_to_projective(cid, base)[0]
    out = np.zeros((n, stride), dtype=np.uint8)
    out[0] = base
    threads = threads or min(64, os.cpu_count() or 1)
    rest = n - 1
    if rest > 0:
        seg = (rest + threads - 1) // threads
        jobs = []
        for j in range(0, rest, seg):
            # the segment's elements are g_{1 + j}, ...: start = (1 + j) g_0
            jobs.append((1 + j, scalar_multiple(oracle, cid, g0, 1 + j), min(seg, rest - j)))
        with concurrent.futures.ThreadPoolExecutor(max_workers=threads) as ex:
            for (at, _, count), (piece, _) in zip(jobs, ex.map(
                    lambda job: oracle.generator_chain(cid, job[1], g0, job[2]), jobs)):
                out[at:at + count] = piece
    return base, out


def weighted_byte_sums(rows_u8, first_row=0, chunk_rows=1 << 14):
    """sum_i (first_row + i + 1) * rows[i, k] for every byte column k of a DEVICE uint8 [n, B]
    tensor, as a list of python ints (exact: 2^8 * 2^23 * 2^23 < 2^63 per column)"""
    import torch
    n, width = rows_u8.shape
    assert first_row + n <= 1 << 23
    total = torch.zeros(width, dtype=torch.int64, device=rows_u8.device)
    # bound the int64 temporaries to ~1 GiB
    step = max(1, min(n, (1 << 27) // max(width, 1)))
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        idx = torch.arange(first_row + lo + 1, first_row + hi + 1, dtype=torch.int64,
                           device=rows_u8.device)
        total += (rows_u8[lo:hi].to(torch.int64) * idx[:, None]).sum(dim=0)
    return [int(v) for v in total.cpu().tolist()]


def weighted_scalar_sum(byte_sums, first_byte, nbytes):
    """sum_i (i + 1) * a_i for the little-endian field at bytes [first_byte, first_byte + nbytes)"""
    return sum(byte_sums[first_byte + k] << (8 * k) for k in range(nbytes))


def scalar_multiple(oracle, cid, base_projective, k):
    """k * base with the reference's own curve operations (double-and-add), None for k = 0"""
    acc = None
    for bit in range(k.bit_length() - 1, -1, -1):
        if acc is not None:
            acc = oracle.double_projective(cid, acc)
        if (k >> bit) & 1:
            acc = base_projective if acc is None else oracle.add_projective(cid, acc,
                                                                             base_projective)
    return acc


def expected_canonical(oracle, cid, base_affine, weighted_sum):
    """canonical bytes of (weighted_sum mod r) * base, as the reference encodes them"""
    s = weighted_sum % ORDER[cid]
    if s == 0:
        p = oracle.affine_to_projective(cid, oracle.identity_affine(cid))[0]
    else:
        p = scalar_multiple(oracle, cid, oracle.affine_to_projective(cid, base_affine)[0], s)
    return np.ascontiguousarray(oracle.canonical(cid, p)).view(np.uint8).reshape(-1)


def config5_bit_table(outputs):
    """BASELINE configs[4]: bit_table[i] = {8, 32, 256}[i mod 3]"""
    return np.array([(8, 32, 256)[i % 3] for i in range(outputs)], dtype=np.uint32)

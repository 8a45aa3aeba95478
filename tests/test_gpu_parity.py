"""-m gpu: the HIP path through the C ABI against the reference oracle, bit-exact.

Every expected value comes from oracle/_ref (the reference's own CPU backend) or from the golden
vectors of rust/tests/src/main.rs:28-47; the product is only ever called through
libblitzar_amd.so.  `bzamd_kernel_launch_count` proves kernels actually ran.
"""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

GOLDEN_DATA = [[2000, 7500, 5000, 1500], [5000, 0, 400000, 10], [7000, 7500, 405000, 1510]]
GOLDEN = [
    [4, 105, 58, 131, 59, 69, 150, 106, 120, 137, 32, 225, 175, 244, 82, 115, 216, 180, 206, 150,
     21, 250, 240, 98, 251, 192, 146, 244, 54, 169, 199, 97],
    [2, 254, 178, 195, 198, 238, 44, 156, 24, 29, 88, 196, 37, 63, 157, 50, 236, 159, 61, 49, 153,
     181, 79, 126, 55, 188, 67, 1, 228, 248, 72, 51],
    [30, 237, 163, 234, 252, 111, 45, 133, 235, 227, 21, 117, 229, 188, 88, 149, 240, 109, 205, 90,
     6, 130, 199, 152, 5, 221, 57, 231, 168, 9, 141, 122],
]


def test_rust_golden_vectors(gpu_backend):
    api = gpu_backend
    before = api.load().bzamd_kernel_launch_count()
    cols = [(np.array(d, dtype=np.uint32), False) for d in GOLDEN_DATA]
    out = api.compute_pedersen_commitments(0, cols)
    assert out.tolist() == GOLDEN
    assert api.load().bzamd_kernel_launch_count() > before


def test_builtin_generators_limb_exact(gpu_backend, oracle):
    api = gpu_backend
    # served from the init-time cache (100), straddling it, and fully beyond it
    for n, off in ((7, 0), (100, 0), (50, 80), (33, 1000), (0, 5)):
        assert np.array_equal(api.get_generators(n, off), oracle.ristretto_generators(n, off))
    assert api.load().sxt_ristretto255_get_generators(None, 3, 0) == 1


def test_one_commit_limb_exact(gpu_backend, oracle):
    api = gpu_backend
    for n in (0, 1, 2, 57, 99, 100, 101, 130):
        assert np.array_equal(api.get_one_commit(n), oracle.one_commit(n))


@pytest.mark.parametrize("curve_id", [0, 1, 2, 3])
def test_mixed_columns_match_oracle(gpu_backend, oracle, curve_id):
    api = gpu_backend
    rng = np.random.default_rng(100 + curve_id)
    n = 97
    gens = util.generators_for(curve_id, n)
    cols = util.mixed_columns(rng, n)
    got = api.compute_pedersen_commitments(curve_id, cols, generators=util.api_generators(curve_id, gens))
    want = oracle.commit(curve_id, cols, gens)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, f"columns {bad.tolist()} differ"


@pytest.mark.parametrize("curve_id,n", [(0, 1 << 12), (0, 20000), (1, 3000), (2, 5000), (3, 1 << 12)])
def test_long_columns_match_oracle(gpu_backend, oracle, curve_id, n):
    api = gpu_backend
    rng = np.random.default_rng(7 * n + curve_id)
    gens = util.generators_for(curve_id, n)
    full = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    masked = full.copy()
    masked[:, 31] &= 0x0f
    cols = [(full, False), (masked, False), (rng.integers(0, 256, (n, 1), dtype=np.uint8), False),
            (rng.integers(0, 256, (n - 17, 4), dtype=np.uint8), False),
            (rng.integers(0, 256, (n, 8), dtype=np.uint8), True),
            (np.ones((n, 1), dtype=np.uint8), False)]
    got = api.compute_pedersen_commitments(curve_id, cols, generators=util.api_generators(curve_id, gens))
    want = oracle.commit(curve_id, cols, gens)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, f"columns {bad.tolist()} differ"


def test_builtin_generators_with_offset(gpu_backend, oracle):
    api = gpu_backend
    rng = np.random.default_rng(3)
    for n, off in ((60, 10), (300, 0), (128, 1 << 20)):
        cols = [(rng.integers(0, 256, (n, 32), dtype=np.uint8), False),
                (rng.integers(0, 256, (n // 2, 2), dtype=np.uint8), True)]
        got = api.compute_pedersen_commitments(0, cols, offset_generators=off)
        want = oracle.commit(0, cols, oracle.ristretto_generators(n, off))
        assert np.array_equal(got, want)


def test_null_and_empty_inputs(gpu_backend):
    api = gpu_backend
    lib = api.load()
    # num_sequences == 0 returns without touching anything (cbindings/pedersen.cc:77-78)
    lib.sxt_curve25519_compute_pedersen_commitments(None, 0, None, 0)
    # zero-length sequences commit to the identity encoding
    out = api.compute_pedersen_commitments(0, [(np.zeros((0, 4), np.uint8), False)])
    assert out.tolist() == [[0] * 32]


#--------------------------------------------------------------------------------------------------
# committed golden fixtures (generated from the reference by tests/golden/make_golden.py)
#--------------------------------------------------------------------------------------------------
import os  # noqa: E402

from tests.golden import make_golden  # noqa: E402

GOLDEN_NPZ = np.load(os.path.join(os.path.dirname(__file__), "golden", "msm_golden.npz"))


@pytest.mark.parametrize("curve_id", [0, 1, 2, 3])
def test_golden_commitments(gpu_backend, curve_id):
    cols = make_golden.golden_columns(1000 + curve_id, 48)
    got = gpu_backend.compute_pedersen_commitments(
        curve_id, cols, generators=GOLDEN_NPZ[f"curve{curve_id}_generators"])
    assert np.array_equal(got, GOLDEN_NPZ[f"curve{curve_id}_commitments"])


def _canon(oracle, cid, res):
    words = res.view(np.uint64).reshape(res.shape[0], -1)
    return np.stack([oracle.canonical(cid, p).view(np.uint8).reshape(-1) for p in words])


@pytest.mark.parametrize("curve_id", [0, 1, 2, 3])
def test_fixed_base_golden_and_file_roundtrip(gpu_backend, oracle, curve_id, tmp_path):
    proj = GOLDEN_NPZ[f"curve{curve_id}_fixed_projective_generators"]
    before = gpu_backend.load().bzamd_kernel_launch_count()
    h = gpu_backend.MultiexpHandle(curve_id, proj)
    res = h.packed_multiexponentiation(GOLDEN_NPZ["fixed_bit_table"], proj.shape[0],
                                       GOLDEN_NPZ[f"curve{curve_id}_fixed_scalars"])
    assert gpu_backend.load().bzamd_kernel_launch_count() > before
    assert np.array_equal(_canon(oracle, curve_id, res),
                          GOLDEN_NPZ[f"curve{curve_id}_fixed_canonical"])
    path = str(tmp_path / "t.bin")
    h.write_to_file(path)
    h.close()
    h2 = gpu_backend.MultiexpHandle(curve_id, filename=path)
    res = h2.packed_multiexponentiation(GOLDEN_NPZ["fixed_bit_table"], proj.shape[0],
                                        GOLDEN_NPZ[f"curve{curve_id}_fixed_scalars"])
    assert np.array_equal(_canon(oracle, curve_id, res),
                          GOLDEN_NPZ[f"curve{curve_id}_fixed_canonical"])
    h2.close()


@pytest.mark.parametrize("curve_id", [0, 1, 2, 3])
def test_fixed_base_plain_packed_vlen_match_oracle(gpu_backend, oracle, curve_id):
    from oracle import fixed_base
    rng = np.random.default_rng(900 + curve_id)
    n = 300
    gens = util.generators_for(curve_id, n)
    proj = gens if curve_id == 0 else oracle.affine_to_projective(curve_id, gens)
    h = gpu_backend.MultiexpHandle(curve_id, proj)
    # mixed 8 / 32 / 256-bit outputs (BASELINE configs[4] shape), checked against the
    # variable-base reference backend on the unpacked columns
    bt = [8, 32, 256] * 4
    s = rng.integers(0, 256, (n, (sum(bt) + 7) // 8), dtype=np.uint8)
    got = _canon(oracle, curve_id, h.packed_multiexponentiation(bt, n, s))
    want = oracle.commit(curve_id, fixed_base.unpack_columns(bt, n, s), gens)
    assert np.array_equal(got[:, :want.shape[1]], want)
    bt, lengths = [4, 12, 1, 64], [0, 5, 5, n]
    s = rng.integers(0, 256, (n, (sum(bt) + 7) // 8), dtype=np.uint8)
    got = _canon(oracle, curve_id, h.vlen_multiexponentiation(bt, lengths, s))
    want = oracle.commit(curve_id, fixed_base.unpack_columns(bt, n, s, lengths), gens)
    assert np.array_equal(got[:, :want.shape[1]], want)
    s = rng.integers(0, 256, (n, 6), dtype=np.uint8)
    got = _canon(oracle, curve_id, h.multiexponentiation(2, 3, n, s))
    want = oracle.commit(curve_id, fixed_base.unpack_columns([16] * 3, n, s), gens)
    assert np.array_equal(got[:, :want.shape[1]], want)
    h.close()


#--------------------------------------------------------------------------------------------------
# BASELINE sizes through size-independent properties (the oracle needs ~80 s for one 2^20 column)
#--------------------------------------------------------------------------------------------------
def test_full_size_properties_curve25519_n2_20(gpu_backend):
    """n = 2^20 (BASELINE configs[1]):
      * all-ones column == compress(one_commit(n))  (prefix sum of the built-in generators),
      * linearity: commit(a) + commit(b) == commit(a + b) with 252-bit a, b,
      * a column and its two halves (second half against offset generators) agree,
      * skewed data (every scalar equal) == scalar * one_commit."""
    from tests import hooks
    api = gpu_backend
    n = 1 << 20
    rng = np.random.default_rng(2020)
    ones = np.ones((n, 1), np.uint8)
    a = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    a[:, 31] &= 0x0f
    b[:, 31] &= 0x0f
    # a + b as 256-bit little-endian integers (fits: both < 2^252)
    wa = a.view("<u8").astype(object)
    wb = b.view("<u8").astype(object)
    carry = np.zeros(n, dtype=object)
    s = np.zeros((n, 4), dtype=np.uint64)
    for k in range(4):
        t = wa[:, k] + wb[:, k] + carry
        s[:, k] = (t & 0xFFFFFFFFFFFFFFFF).astype(np.uint64)
        carry = t >> 64
    ab = s.view(np.uint8).reshape(n, 32)
    same = np.tile(a[:1], (n, 1))
    out = api.compute_pedersen_commitments(0, [(ones, False), (a, False), (b, False), (ab, False),
                                               (same, False)])
    one_commit = api.get_one_commit(n)
    assert np.array_equal(out[0], hooks.ristretto_encode(one_commit))
    pa, pb = hooks.ristretto_decode(out[1]), hooks.ristretto_decode(out[2])
    assert np.array_equal(hooks.ristretto_encode(hooks.ed_add(pa, pb)), out[3])
    lo = api.compute_pedersen_commitments(0, [(a[:n // 2], False)])
    hi = api.compute_pedersen_commitments(0, [(a[n // 2:], False)], offset_generators=n // 2)
    both = hooks.ed_add(hooks.ristretto_decode(lo[0]), hooks.ristretto_decode(hi[0]))
    assert np.array_equal(hooks.ristretto_encode(both), out[1])
    # scalar * one_commit by double-and-add on the host hooks
    k = int.from_bytes(a[0].tobytes(), "little")
    acc = api.get_one_commit(0)  # identity
    for bit in range(k.bit_length() - 1, -1, -1):
        acc = hooks.ed_dbl(acc)
        if (k >> bit) & 1:
            acc = hooks.ed_add(acc, one_commit)
    assert np.array_equal(hooks.ristretto_encode(acc), out[4])


def test_long_column_properties_n2_23(gpu_backend):
    """n = 2^23 rows: beyond the sizes the oracle reaches, and a different regime of the two-pass
    sort (1024 bucket groups of ~8192 records each, all streamed instead of staged in LDS).
    Size-independent properties: the all-ones column is the prefix sum of the generators,
    commit(a) + commit(b) == commit(a + b) for 128-bit columns, a column equals the sum of its
    halves."""
    from tests import hooks
    api = gpu_backend
    n = 1 << 23
    rng = np.random.default_rng(23)
    a = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    b = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    wa, wb = a.view("<u8"), b.view("<u8")
    lo = wa[:, 0] + wb[:, 0]
    carry = (lo < wa[:, 0]).astype(np.uint64)
    hi = wa[:, 1] + wb[:, 1]
    carry2 = (hi < wa[:, 1]).astype(np.uint64)
    hi2 = hi + carry
    carry2 |= (hi2 < hi).astype(np.uint64)
    ab = np.zeros((n, 3), dtype=np.uint64)            # 129-bit sums in 24-byte scalars
    ab[:, 0], ab[:, 1], ab[:, 2] = lo, hi2, carry2
    ab = ab.view(np.uint8).reshape(n, 24)
    out = api.compute_pedersen_commitments(0, [(np.ones((n, 1), np.uint8), False), (a, False),
                                               (b, False), (ab, False)])
    assert np.array_equal(out[0], hooks.ristretto_encode(api.get_one_commit(n)))
    pa, pb = hooks.ristretto_decode(out[1]), hooks.ristretto_decode(out[2])
    assert np.array_equal(hooks.ristretto_encode(hooks.ed_add(pa, pb)), out[3])
    first = api.compute_pedersen_commitments(0, [(a[:n // 2], False)])
    second = api.compute_pedersen_commitments(0, [(a[n // 2:], False)], offset_generators=n // 2)
    both = hooks.ed_add(hooks.ristretto_decode(first[0]), hooks.ristretto_decode(second[0]))
    assert np.array_equal(hooks.ristretto_encode(both), out[1])


def test_row_sharded_fold_on_device(gpu_backend, oracle):
    """the single-GPU half of the row-sharded path: projective partials of two row ranges folded
    and canonicalised == the unsharded commitment"""
    import ctypes
    api = gpu_backend
    rng = np.random.default_rng(11)
    for curve_id in (0, 2):
        n = 5000
        gens = util.generators_for(curve_id, n)
        g = util.api_generators(curve_id, gens)
        cols = [(rng.integers(0, 256, (n, 32), dtype=np.uint8), False),
                (rng.integers(0, 256, (n, 8), dtype=np.uint8), True)]
        parts = []
        for lo, hi in ((0, 1700), (1700, n)):
            parts.append(api.msm_projective(curve_id, [(c[lo:hi], s) for c, s in cols], g[lo:hi]))
        got = api.fold_encode(curve_id, np.stack(parts))
        assert np.array_equal(got, oracle.commit(curve_id, cols, gens))
        _ = ctypes


def test_many_short_columns(gpu_backend, oracle):
    """the reference's bucket_method2 regime (sxt/multiexp/bucket_method2/multiexponentiation.h:48-121:
    256 <= n <= 4096, many outputs): the planner gives such launches 32 or fewer entries per
    accumulation lane, tasks of 256 buckets and k_reduce blocks of ONE wavefront (64 lanes x 4
    buckets, scan and tree over the lanes in use); one launch also mixes a few much shorter columns
    in, whose tasks use a fraction of the lanes"""
    api = gpu_backend
    rng = np.random.default_rng(4096)
    # (70 columns and more: k_horner in one-wavefront blocks too)
    for curve_id, n, columns in ((0, 4096, 12), (2, 4096, 8), (3, 1024, 9), (1, 2048, 6), (0, 512, 70),
                                 (2, 300, 66)):
        gens = util.generators_for(curve_id, n)
        g = util.api_generators(curve_id, gens)
        cols = [(rng.integers(0, 256, (n, 32), dtype=np.uint8), False) for _ in range(columns)]
        cols += [(rng.integers(0, 256, (300, 32), dtype=np.uint8), False),
                 (rng.integers(0, 256, (n - 1, 16), dtype=np.uint8), False),
                 (rng.integers(0, 256, (1, 32), dtype=np.uint8), False)]
        got = api.compute_pedersen_commitments(curve_id, cols, generators=g)
        assert np.array_equal(got, oracle.commit(curve_id, cols, gens)), curve_id


def test_skewed_and_batched_columns(gpu_backend, oracle):
    """digit distributions that put most entries in one bucket (equal scalars, two values, all
    ones, sparse) and many-column jobs split into several batches by the engine; results never
    depend on the tuning knobs"""
    api = gpu_backend
    lib = api.load()
    rng = np.random.default_rng(77)
    n = 9000
    gens = util.generators_for(0, n)
    g = util.api_generators(0, gens)
    one = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    two = rng.integers(0, 256, (2, 32), dtype=np.uint8)
    sparse = np.zeros((n, 32), np.uint8)
    sparse[rng.integers(0, n, 40)] = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    cols = [(np.tile(one, (n, 1)), False), (two[rng.integers(0, 2, n)], False),
            (np.full((n, 32), 0xff, np.uint8), False), (sparse, False),
            (np.ones((n, 1), np.uint8), False), (rng.integers(0, 2, (n, 1), dtype=np.uint8), False)]
    small = np.zeros((n, 32), np.uint8)           # 40-bit values in 32-byte scalars: the top
    small[:, :5] = rng.integers(0, 256, (n, 5))    # windows are empty and the chain skips them
    cols.append((small, False))
    cols += [(rng.integers(0, 256, (n - 7 * k, 1 + (5 * k) % 32), dtype=np.uint8), False)
             for k in range(30)]
    want = oracle.commit(0, cols, gens)
    assert np.array_equal(api.compute_pedersen_commitments(0, cols, generators=g), want)
    try:
        lib.bzamd_set_tuning(11, 40, 0)          # narrower windows, ~3 columns per batch
        assert np.array_equal(api.compute_pedersen_commitments(0, cols, generators=g), want)
        lib.bzamd_set_tuning(16, 32768, 1 << 20)  # 1 MiB of workspace: one column per batch
        assert np.array_equal(api.compute_pedersen_commitments(0, cols, generators=g), want)
        # bucket reduction geometry: 2 .. 64 buckets per lane (one partial per 512 .. 16384 buckets)
        lib.bzamd_set_tuning(16, 32768, 64 << 30)
        # and 8 .. 128 sorted entries per accumulation lane
        for acc_log2, red_log2 in ((0, 1), (3, 4), (6, 6), (7, 0)):
            lib.bzamd_set_segments(acc_log2, red_log2)
            assert np.array_equal(api.compute_pedersen_commitments(0, cols, generators=g), want)
    finally:
        lib.bzamd_set_tuning(16, 32768, 64 << 30)
        lib.bzamd_set_segments(0, 0)


def test_skewed_long_column(gpu_backend):
    """2^18 equal 252-bit scalars: every window has one bucket spanning 8192 segments"""
    from tests import hooks
    api = gpu_backend
    n = 1 << 18
    rng = np.random.default_rng(5)
    s = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    s[0, 31] &= 0x0f
    out = api.compute_pedersen_commitments(0, [(np.tile(s, (n, 1)), False)])
    k = int.from_bytes(s[0].tobytes(), "little")
    one_commit = api.get_one_commit(n)
    acc = api.get_one_commit(0)
    for bit in range(k.bit_length() - 1, -1, -1):
        acc = hooks.ed_dbl(acc)
        if (k >> bit) & 1:
            acc = hooks.ed_add(acc, one_commit)
    assert np.array_equal(hooks.ristretto_encode(acc), out[0])


def test_skewed_long_columns_match_oracle(gpu_backend, oracle):
    """2^17 rows of the shapes table columns have -- all ones, booleans, two distinct 128-bit
    values, 10-bit integers, a mostly-zero column: bucket groups far beyond what one workgroup
    sorts, split into cooperating chunks (k_group_sort_big), and heavy buckets folded by whole
    workgroups in k_reduce"""
    api = gpu_backend
    rng = np.random.default_rng(11)
    n = 1 << 17
    gens = util.generators_for(0, n)
    g = util.api_generators(0, gens)
    two = rng.integers(0, 256, (2, 16), dtype=np.uint8)
    small = np.zeros((n, 8), np.uint8)
    small[:, :2] = rng.integers(0, 256, (n, 2))
    small[:, 1] &= 0x03
    sparse = np.zeros((n, 16), np.uint8)
    hits = rng.integers(0, n, n // 10)
    sparse[hits] = rng.integers(0, 256, (hits.size, 16), dtype=np.uint8)
    cols = [(np.ones((n, 1), np.uint8), False), (rng.integers(0, 2, (n, 1), dtype=np.uint8), False),
            (two[rng.integers(0, 2, n)], False), (small, False), (sparse, False),
            (np.full((n, 2), 0xff, np.uint8), True)]
    want = oracle.commit(0, cols, gens)
    assert np.array_equal(api.compute_pedersen_commitments(0, cols, generators=g), want)


def test_device_resident_entry_points(gpu_backend, oracle):
    """include/blitzar_amd.h: operands already in HBM (torch tensors only provide the memory),
    resident generator sets reused across calls, caller-provided stream"""
    import ctypes
    import torch
    api = gpu_backend
    lib = api.load()
    dev = torch.device("cuda", 0)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(21)
    for curve_id in (0, 2):
        n = 3000
        gens = util.generators_for(curve_id, n)
        g_host = np.ascontiguousarray(util.api_generators(curve_id, gens))
        cols = [(rng.integers(0, 256, (n, 32), dtype=np.uint8), False),
                (rng.integers(0, 256, (n - 5, 4), dtype=np.uint8), True)]
        want = oracle.commit(curve_id, cols, gens)
        d_cols = [torch.from_numpy(c.copy()).to(dev) for c, _ in cols]
        desc = (api.sxt_sequence_descriptor * 2)()
        for i, ((c, s), d) in enumerate(zip(cols, d_cols)):
            desc[i] = api.sxt_sequence_descriptor(c.shape[1], c.shape[0], d.data_ptr(), int(s))
        d_gens = torch.from_numpy(g_host.copy()).to(dev)
        out = torch.zeros((2, want.shape[1]), dtype=torch.uint8, device=dev)
        lib.bzamd_msm_device(curve_id, ctypes.c_void_p(out.data_ptr()), 2, desc,
                             ctypes.c_void_p(d_gens.data_ptr()), stream)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
        for maker in ("host", "device"):
            if maker == "host":
                h = lib.bzamd_generators_new_host(curve_id, g_host.ctypes.data_as(ctypes.c_void_p), n)
            else:
                h = lib.bzamd_generators_new_device(curve_id, ctypes.c_void_p(d_gens.data_ptr()), n,
                                                    stream)
            out.zero_()
            for _ in range(2):  # reused across calls
                lib.bzamd_msm_device_resident(ctypes.c_void_p(out.data_ptr()), 2, desc, h, stream)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), want)
            lib.bzamd_generators_free(h)


@pytest.mark.parametrize("side_stream", [True, False])
def test_pipelined_calls(gpu_backend, oracle, side_stream):
    """bzamd_pipeline_next / bzamd_pipeline_flush (include/blitzar_amd.h): a sequence of calls whose
    stages overlap on the engine's own streams (front of call k + 1 beside the accumulation of call k
    beside the tails of call k - 1).  Calls with the same descriptors (the fast path: nothing is
    re-uploaded), with different ones, on two curves and through the resident entry point; every
    output buffer is read after TWO further calls were enqueued or after a flush, on the caller's
    stream; operands are overwritten right behind a call (they are consumed in stream order).  On
    a stream of the caller's own (no implicit synchronisation hides an ordering bug) and on the
    NULL stream."""
    import ctypes
    import torch
    api = gpu_backend
    lib = api.load()
    dev = torch.device("cuda", 0)
    torch_stream = torch.cuda.Stream(device=dev) if side_stream else torch.cuda.current_stream()
    with torch.cuda.stream(torch_stream):
        _pipelined_calls(api, lib, oracle, dev, ctypes, torch)
    torch.cuda.synchronize()


def _pipelined_calls(api, lib, oracle, dev, ctypes, torch):
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(33)
    jobs = []
    for curve_id, n in ((0, 40000), (2, 2500)):
        gens = util.generators_for(curve_id, n)
        g_host = np.ascontiguousarray(util.api_generators(curve_id, gens))
        d_gens = torch.from_numpy(g_host.copy()).to(dev)
        for rows, nbytes in ((n, 32), (n - 11, 32), (n, 5), (n, 32)):
            col = rng.integers(0, 256, (rows, nbytes), dtype=np.uint8)
            want = oracle.commit(curve_id, [(col, False)], gens)
            d_col = torch.from_numpy(col.copy()).to(dev)
            desc = (api.sxt_sequence_descriptor * 1)()
            desc[0] = api.sxt_sequence_descriptor(nbytes, rows, d_col.data_ptr(), 0)
            jobs.append((curve_id, d_gens, d_col, desc, want))
    torch.cuda.current_stream().synchronize()
    # every job three times in a row (identical descriptors), jobs interleaved (changing ones),
    # and two jobs of one shape over different buffers alternating (the overlapping path with a
    # descriptor upload per call)
    order = ([j for j in range(len(jobs)) for _ in range(3)] + list(range(len(jobs))) * 2 +
             [0, 3] * 4 + [4, 7] * 3)
    outs, copies = [], []
    for j in order:
        curve_id, d_gens, _, desc, want = jobs[j]
        out = torch.zeros((1, want.shape[1]), dtype=torch.uint8, device=dev)
        lib.bzamd_pipeline_next()
        lib.bzamd_msm_device(curve_id, ctypes.c_void_p(out.data_ptr()), 1, desc,
                             ctypes.c_void_p(d_gens.data_ptr()), stream)
        outs.append((j, out))
        if len(outs) >= 3:  # two calls later a result is complete on the stream: copy it out
            copies.append((outs[-3][0], outs[-3][1].clone()))
    lib.bzamd_pipeline_flush(stream)
    for j, out in outs[-2:]:
        copies.append((j, out.clone()))
    torch.cuda.synchronize()
    assert len(copies) == len(order)
    for j, got in copies:
        assert np.array_equal(got.cpu().numpy(), jobs[j][4]), f"pipelined job {j}"
    # operands are consumed in stream order: a scratch column and scratch generators refilled right
    # behind every call, alternating between two jobs of one shape
    curve_id = jobs[0][0]
    scratch_col = torch.empty_like(jobs[0][2])
    scratch_gens = torch.empty_like(jobs[0][1])
    sdesc = (api.sxt_sequence_descriptor * 1)()
    sdesc[0] = api.sxt_sequence_descriptor(32, 40000, scratch_col.data_ptr(), 0)
    results = []
    for i in range(12):
        j = 0 if i % 2 == 0 else 3
        scratch_col.copy_(jobs[j][2])
        scratch_gens.copy_(jobs[j][1])
        out = torch.zeros((1, 32), dtype=torch.uint8, device=dev)
        lib.bzamd_pipeline_next()
        lib.bzamd_msm_device(curve_id, ctypes.c_void_p(out.data_ptr()), 1, sdesc,
                             ctypes.c_void_p(scratch_gens.data_ptr()), stream)
        scratch_col.zero_()   # behind the call in stream order: must not reach its front
        scratch_gens.zero_()
        results.append((j, out))
    lib.bzamd_pipeline_flush(stream)
    torch.cuda.synchronize()
    for j, out in results:
        assert np.array_equal(out.cpu().numpy(), jobs[j][4]), "operands overwritten behind a call"
    # calls whose columns are all empty keep their place in a sequence (identities, no tasks), and
    # the two-calls rule holds across them; a flush from ANOTHER stream completes everything there
    empty = (api.sxt_sequence_descriptor * 2)()
    empty[0] = api.sxt_sequence_descriptor(32, 0, None, 0)
    empty[1] = api.sxt_sequence_descriptor(4, 0, None, 1)
    identity = oracle.commit(0, [(np.zeros((0, 32), np.uint8), False)] * 2,
                             oracle.ristretto_generators(1))
    curve_id, d_gens, _, desc, want = jobs[0]
    seq = []
    for i in range(6):
        lib.bzamd_pipeline_next()
        if i % 2 == 0:
            out = torch.zeros((1, 32), dtype=torch.uint8, device=dev)
            lib.bzamd_msm_device(curve_id, ctypes.c_void_p(out.data_ptr()), 1, desc,
                                 ctypes.c_void_p(d_gens.data_ptr()), stream)
            seq.append((want, out))
        else:
            out = torch.ones((2, 32), dtype=torch.uint8, device=dev)
            lib.bzamd_msm_device(curve_id, ctypes.c_void_p(out.data_ptr()), 2, empty,
                                 ctypes.c_void_p(d_gens.data_ptr()), stream)
            seq.append((identity, out))
    other = torch.cuda.Stream(device=dev)
    lib.bzamd_pipeline_flush(ctypes.c_void_p(other.cuda_stream))
    with torch.cuda.stream(other):
        snaps = [(w, o.clone()) for w, o in seq]
    other.synchronize()
    for w, got in snaps:
        assert np.array_equal(got.cpu().numpy(), w), "empty calls inside a pipelined sequence"
    torch.cuda.synchronize()
    # a plain call after a pipelined one needs no flush of its own
    curve_id, d_gens, _, desc, want = jobs[0]
    h = lib.bzamd_generators_new_device(curve_id, ctypes.c_void_p(d_gens.data_ptr()), 40000, stream)
    a = torch.zeros((1, 32), dtype=torch.uint8, device=dev)
    b = torch.zeros((1, 32), dtype=torch.uint8, device=dev)
    lib.bzamd_pipeline_next()
    lib.bzamd_msm_device_resident(ctypes.c_void_p(a.data_ptr()), 1, desc, h, stream)
    lib.bzamd_msm_device(curve_id, ctypes.c_void_p(b.data_ptr()), 1, desc,
                         ctypes.c_void_p(d_gens.data_ptr()), stream)
    got_a, got_b = a.clone(), b.clone()
    torch.cuda.synchronize()
    assert np.array_equal(got_a.cpu().numpy(), want) and np.array_equal(got_b.cpu().numpy(), want)
    lib.bzamd_generators_free(h)
    # a request is consumed by the next DEVICE entry point only: a blocking call in between reads
    # its own results right away and must not be deferred
    lib.bzamd_pipeline_next()
    host_cols = [(rng.integers(0, 256, (3000, 32), dtype=np.uint8), False)]
    host_gens = util.generators_for(0, 3000)
    assert np.array_equal(
        api.compute_pedersen_commitments(0, host_cols, generators=util.api_generators(0, host_gens)),
        oracle.commit(0, host_cols, host_gens))
    lib.bzamd_pipeline_flush(stream)
    # a random mix: deferred and plain calls, one and five columns, flushes now and then; a
    # deferred result is read once two further calls, a plain call or a flush were enqueued
    curve_id, d_gens = jobs[0][0], jobs[0][1]
    five = (api.sxt_sequence_descriptor * 5)()
    for c in range(5):
        five[c] = jobs[c % 4][3][0]
    want5 = np.concatenate([jobs[c % 4][4] for c in range(5)])
    pending = []  # [want, out, calls enqueued since]
    mixed = []
    for _ in range(60):
        kind = int(rng.integers(0, 4))
        if kind == 3:
            out = torch.zeros((5, 32), dtype=torch.uint8, device=dev)
            deferred = bool(rng.integers(0, 2))
            if deferred:
                lib.bzamd_pipeline_next()
            lib.bzamd_msm_device(curve_id, ctypes.c_void_p(out.data_ptr()), 5, five,
                                 ctypes.c_void_p(d_gens.data_ptr()), stream)
            want_k = want5
        else:
            j = int(rng.integers(0, 4))
            out = torch.zeros((1, 32), dtype=torch.uint8, device=dev)
            deferred = kind != 0
            if deferred:
                lib.bzamd_pipeline_next()
            lib.bzamd_msm_device(curve_id, ctypes.c_void_p(out.data_ptr()), 1, jobs[j][3],
                                 ctypes.c_void_p(d_gens.data_ptr()), stream)
            want_k = jobs[j][4]
        for item in pending:
            item[2] += 1
        done = [it for it in pending if it[2] >= 2 or not deferred]
        pending = [it for it in pending if not (it[2] >= 2 or not deferred)]
        for w, o, _ in done:  # complete on the stream by now
            mixed.append((w, o.clone()))
        if deferred:
            pending.append([want_k, out, 0])
        else:
            mixed.append((want_k, out.clone()))
        if rng.integers(0, 5) == 0:
            lib.bzamd_pipeline_flush(stream)
            for w, o, _ in pending:
                mixed.append((w, o.clone()))
            pending = []
    lib.bzamd_pipeline_flush(stream)
    for w, o, _ in pending:
        mixed.append((w, o.clone()))
    torch.cuda.synchronize()
    for w, got in mixed:
        assert np.array_equal(got.cpu().numpy(), w)


@pytest.mark.parametrize("curve_id", [0, 1, 2, 3])
def test_random_sweep_matches_oracle(gpu_backend, oracle, curve_id):
    """the randomized part of the reference exerciser (sxt/multiexp/test/multiexponentiation.cc:
    290-451) through the HIP path: 1-10 sequences, length 0-100, 1-32 byte scalars, alternating
    signedness"""
    api = gpu_backend
    rng = np.random.default_rng(8000 + curve_id)
    gens = util.generators_for(curve_id, 100)
    g = util.api_generators(curve_id, gens)
    for _ in range(6):
        cols = []
        for s in range(int(rng.integers(1, 11))):
            signed = bool(s % 2)
            nb = int(rng.choice([1, 2, 4, 8, 16])) if signed else int(rng.integers(1, 33))
            cols.append((rng.integers(0, 256, (int(rng.integers(0, 101)), nb), dtype=np.uint8),
                         signed))
        got = api.compute_pedersen_commitments(curve_id, cols, generators=g)
        assert np.array_equal(got, oracle.commit(curve_id, cols, gens))


@pytest.mark.parametrize("curve_id", [1, 2, 3])
def test_repeated_and_cancelling_generators(gpu_backend, oracle, curve_id):
    """degenerate inputs for the complete addition formulas: every generator the same point
    (bucket sums are repeated doublings), and pairs that cancel to the identity"""
    api = gpu_backend
    rng = np.random.default_rng(8100 + curve_id)
    n = 600
    base = util.generators_for(curve_id, 8)
    gens = np.tile(base[1], (n, 1))
    g = util.api_generators(curve_id, gens)
    a = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    same = np.tile(a[:1], (n, 1))
    # x and -x on the same point: signed 8-byte column with values v, -v, v, -v ...
    v = rng.integers(1, 2**62, n // 2, dtype=np.int64)
    pm = np.empty(n, np.int64)
    pm[0::2], pm[1::2] = v, -v
    cols = [(a, False), (same, False), (pm, True), (np.ones((n, 1), np.uint8), False)]
    got = api.compute_pedersen_commitments(curve_id, cols, generators=g)
    want = oracle.commit(curve_id, cols, gens)
    assert np.array_equal(got, want)
    assert np.array_equal(got[2], oracle.commit(curve_id, [(np.zeros((0, 1), np.uint8), False)],
                                                gens)[0])  # cancels to the identity encoding


@pytest.mark.parametrize("curve_id,width,n", [(0, 4, 11), (0, 16, 20), (0, 11, 30), (1, 8, 9),
                                              (2, 16, 17), (3, 1, 5), (3, 13, 14)])
def test_partition_table_built_on_device(gpu_backend, oracle, curve_id, width, n, tmp_path,
                                         monkeypatch):
    """sxt_multiexp_handle_write_to_file on the GPU backend builds the 2^w subset sums of every
    window on the device (fixed/partition_table_device.h: half tables + one addition per entry,
    inversions shared two levels deep); the file must equal the reference's own
    compute_partition_table output byte for byte, identity padding and sentinels included"""
    from oracle import fixed_base
    monkeypatch.setenv("BLITZAR_PARTITION_WINDOW_WIDTH", str(width))
    gens = util.generators_for(curve_id, n)  # index 5 is the identity for the Weierstrass curves
    proj = gens if curve_id == 0 else oracle.affine_to_projective(curve_id, gens)
    before = gpu_backend.load().bzamd_kernel_launch_count()
    h = gpu_backend.MultiexpHandle(curve_id, proj)
    path = str(tmp_path / "table.bin")
    h.write_to_file(path)
    h.close()
    assert gpu_backend.load().bzamd_kernel_launch_count() >= before + 4
    with open(path, "rb") as fh:
        data = fh.read()
    want = fixed_base.PartitionTable(curve_id, proj, width).file_bytes()
    assert len(data) == len(want)
    assert data == want


#--------------------------------------------------------------------------------------------------
# sequences longer than one pass of the engine; mid-size columns in the c = 16 regime; the synthetic
# generator sets of the full-size checks; multi-chunk blocking calls
#--------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("curve_id", [0, 2])
def test_rows_in_several_passes(gpu_backend, oracle, curve_id):
    """a sequence longer than bzamd_set_max_rows_per_pass runs as row ranges whose projective
    partials are folded (the ABI's n is a u64, sxt/multiexp/base/exponent_sequence.h:25-42): the
    blocking entry points with caller and built-in generators, the device entry point and the
    resident one -- 5003 rows in passes of at most 1000 against the reference"""
    import ctypes
    import torch
    api = gpu_backend
    lib = api.load()
    rng = np.random.default_rng(4100 + curve_id)
    n = 5003
    gens = util.generators_for(curve_id, n)
    g_api = util.api_generators(curve_id, gens)
    cols = [(rng.integers(0, 256, (n, 32), dtype=np.uint8), False),
            (rng.integers(0, 256, (n - 1001, 7), dtype=np.uint8), False),
            (rng.integers(0, 256, (999, 4), dtype=np.uint8), True),
            (np.zeros((0, 8), np.uint8), False),
            (rng.integers(0, 256, (n, 1), dtype=np.uint8), False)]
    want = oracle.commit(curve_id, cols, gens)
    lib.bzamd_set_max_rows_per_pass(1000)
    try:
        assert np.array_equal(api.compute_pedersen_commitments(curve_id, cols, generators=g_api), want)
        if curve_id == 0:
            off = 37
            want_off = oracle.commit(0, cols, oracle.ristretto_generators(n, off))
            assert np.array_equal(api.compute_pedersen_commitments(0, cols, offset_generators=off),
                                  want_off)
        dev = torch.device("cuda", 0)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        d_gens = torch.from_numpy(np.ascontiguousarray(g_api).copy()).to(dev)
        keep = [torch.from_numpy(np.ascontiguousarray(c).view(np.uint8).reshape(len(c), c.shape[1])
                                 .copy()).to(dev) for c, _ in cols]
        desc = (api.sxt_sequence_descriptor * len(cols))()
        for i, (c, signed) in enumerate(cols):
            desc[i] = api.sxt_sequence_descriptor(c.shape[1], len(c),
                                                  keep[i].data_ptr() if len(c) else None,
                                                  1 if signed else 0)
        out = torch.zeros((len(cols), want.shape[1]), dtype=torch.uint8, device=dev)
        lib.bzamd_pipeline_next()  # ignored by a call of several passes
        lib.bzamd_msm_device(curve_id, ctypes.c_void_p(out.data_ptr()), len(cols), desc,
                             ctypes.c_void_p(d_gens.data_ptr()), stream)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
        h = lib.bzamd_generators_new_device(curve_id, ctypes.c_void_p(d_gens.data_ptr()), n, stream)
        out.zero_()
        lib.bzamd_msm_device_resident(ctypes.c_void_p(out.data_ptr()), len(cols), desc, h, stream)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
        lib.bzamd_generators_free(h)
    finally:
        lib.bzamd_set_max_rows_per_pass(1 << 28)


@pytest.mark.parametrize("curve_id", [0, 1, 3])
@pytest.mark.parametrize("chunks", [2, 3, 7])
def test_blocking_call_as_a_row_pipeline(gpu_backend, oracle, curve_id, chunks):
    """a blocking call with host operands cut into row chunks (chunk k commits to projective partials
    while chunk k + 1 uploads, one fold at the end): forced on a small input, caller and built-in
    generators (inside and straddling the init-time cache), columns of unequal length, signed and
    empty ones -- the same bytes as the reference"""
    api = gpu_backend
    lib = api.load()
    rng = np.random.default_rng(5200 + 10 * curve_id + chunks)
    n = 4001
    gens = util.generators_for(curve_id, n)
    g_api = util.api_generators(curve_id, gens)
    cols = [(rng.integers(0, 256, (n, 32), dtype=np.uint8), False),
            (rng.integers(0, 256, (n - 1500, 9), dtype=np.uint8), False),
            (rng.integers(0, 256, (700, 4), dtype=np.uint8), True),
            (np.zeros((0, 8), np.uint8), False),
            (rng.integers(0, 256, (n, 2), dtype=np.uint8), True)]
    want = oracle.commit(curve_id, cols, gens)
    lib.bzamd_set_row_pipeline_chunks(chunks)
    try:
        before = lib.bzamd_kernel_launch_count()
        got = api.compute_pedersen_commitments(curve_id, cols, generators=g_api)
        assert lib.bzamd_kernel_launch_count() - before >= 6 * chunks  # every chunk ran the engine
        assert np.array_equal(got, want)
        # with caller generators the first four columns are cut into the chunks and the others follow
        # whole on the addends the chunks left: the longest column last (the lead columns end before
        # the generators do: their later chunks are empty), and six columns (two whole ones)
        back = [cols[2], cols[3], cols[1], cols[4], cols[0]]
        assert np.array_equal(api.compute_pedersen_commitments(curve_id, back, generators=g_api),
                              want[[2, 3, 1, 4, 0]])
        six = cols + [(rng.integers(0, 256, (n - 7, 32), dtype=np.uint8), False)]
        assert np.array_equal(api.compute_pedersen_commitments(curve_id, six, generators=g_api),
                              oracle.commit(curve_id, six, gens))
        if curve_id == 0:
            # the session's backend caches 100 built-in generators: rows 0..99 resident, the rest of
            # a longer column derived on the fly (that shape is not pipelined; it must still agree)
            short = [(c[:90], sgn) for c, sgn in cols]
            for off in (0, 7):
                want_off = oracle.commit(0, short, oracle.ristretto_generators(90, off))
                assert np.array_equal(api.compute_pedersen_commitments(0, short, offset_generators=off),
                                      want_off)
            want_far = oracle.commit(0, cols, oracle.ristretto_generators(n, 50))
            assert np.array_equal(api.compute_pedersen_commitments(0, cols, offset_generators=50),
                                  want_far)
    finally:
        lib.bzamd_set_row_pipeline_chunks(0)


def test_sequence_of_2_31_plus_5_rows(gpu_backend, oracle):
    """n = 2^31 + 5 (a 2 GiB column of bytes): beyond what a 31-bit row index holds, nine passes of
    the engine against built-in generators derived on the fly.  The column is zero except for rows
    at the pass boundaries and past 2^31, so the reference can produce the expected commitment from
    those rows' generators alone (compute_base_element at the same indices)."""
    api = gpu_backend
    n = (1 << 31) + 5
    col = np.zeros((n, 1), dtype=np.uint8)
    rows = [0, 1, (1 << 28) - 1, 1 << 28, (1 << 30) + 12345, (1 << 31) - 1, 1 << 31, (1 << 31) + 4]
    rng = np.random.default_rng(31)
    vals = rng.integers(1, 256, len(rows), dtype=np.uint8)
    col[rows, 0] = vals
    before = api.load().bzamd_kernel_launch_count()
    got = api.compute_pedersen_commitments(0, [(col, False)])
    assert api.load().bzamd_kernel_launch_count() - before >= 9 * 10
    gens = np.concatenate([oracle.ristretto_generators(1, r) for r in rows])
    want = oracle.commit(0, [(vals.reshape(-1, 1), False)], gens)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("curve_id", [1, 2, 3])
def test_mid_size_column_with_16_bit_windows(gpu_backend, oracle, curve_id):
    """a 2^16-row column on ORACLE-made generators (1024 outputs of the reference's
    generate_random_element, then the chain g_i = g_{i-1} + g_0 built with the reference's curve
    code) with the window width pinned to 16: stored digit -32768, 2^15 buckets per window -- the
    regime of BASELINE configs 3-5, compared with the reference MSM itself"""
    api = gpu_backend
    lib = api.load()
    n = 1 << 16
    rng = np.random.default_rng(6100 + curve_id)
    gens = util.weierstrass_generators(curve_id, n, distinct_seeds=1024)
    col = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    col[:, 31] &= 0x0f
    want = oracle.commit(curve_id, [(col, False)], gens)
    lib.bzamd_set_window_bits(16)
    try:
        got = api.compute_pedersen_commitments(curve_id, [(col, False)],
                                               generators=util.api_generators(curve_id, gens))
    finally:
        lib.bzamd_set_window_bits(0)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("curve_id", [1, 2, 3])
def test_generator_multiples_match_reference_additions(gpu_backend, oracle, curve_id):
    """bzamd_generator_multiples_device (the synthetic generator sets g_i = (i + 1) G of the
    full-size checks of configs 3-5) against the chain G, G + G, ... built with the reference's own
    addition, for the first 300 rows"""
    import ctypes
    import torch
    lib = gpu_backend.load()
    dev = torch.device("cuda", 0)
    _, nl, stride, _ = oracle.CURVES[curve_id]
    base = oracle.random_affine(curve_id, 1, 2)
    n = 300
    d_base = torch.from_numpy(np.ascontiguousarray(base).view(np.uint8).reshape(-1).copy()).to(dev)
    out = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    lib.bzamd_generator_multiples_device(curve_id, ctypes.c_void_p(out.data_ptr()),
                                         ctypes.c_void_p(d_base.data_ptr()), n, None)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    g0 = oracle.affine_to_projective(curve_id, base.reshape(1, -1))[0]
    acc = g0
    for i in range(n):
        if i:
            acc = oracle.add_projective(curve_id, acc, g0)
        want = np.ascontiguousarray(oracle.to_affine(curve_id, acc)).view(np.uint8).reshape(-1)
        assert np.array_equal(got[i, :16 * nl], want[:16 * nl]) and got[i, 16 * nl] == 0, f"row {i}"


def test_blocking_call_of_two_chunks_bn254(gpu_backend, oracle):
    """host buffers are uploaded in chunks of 48 MiB of scalars while the engine works on the
    previous chunk, and the chunks of a call run in throughput mode among themselves: two 2^20-row
    bn254 columns of 32-byte elements and a short one = two chunks, against the reference (24-bit
    values: the reference's cost follows the bit width; 16 distinct generators repeated)"""
    api = gpu_backend
    n = 1 << 20
    rng = np.random.default_rng(254)
    gens = np.tile(util.weierstrass_generators(2, 16, distinct_seeds=16), (n // 16, 1))
    cols = []
    for rows in (n, n, 1000):
        c = np.zeros((rows, 32), dtype=np.uint8)
        c[:, :3] = rng.integers(0, 256, (rows, 3), dtype=np.uint8)
        cols.append((c, False))
    want = oracle.commit(2, cols, gens)
    got = api.compute_pedersen_commitments(2, cols, generators=util.api_generators(2, gens))
    assert np.array_equal(got, want)


_FORCED_KNOB_SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from blitzar_amd import api
data = np.load(sys.argv[2])
assert api.init(api.SXT_GPU_BACKEND, 0) == 0
lib = api.load()
lib.bzamd_set_window_bits(16)  # 32768 buckets per window: 16 k_reduce blocks per task
out = {}
for cid in (0, 1, 2, 3):
    cols = [(data[f"s{cid}_{k}"], False) for k in range(2)]
    out[f"c{cid}"] = api.compute_pedersen_commitments(cid, cols, generators=data[f"g{cid}"])
np.savez(sys.argv[3], **out)
"""


@pytest.mark.parametrize("knobs", [{"BLITZAR_AMD_COMPACT_REDUCE": "1"}, {"BLITZAR_AMD_COMPACT_REDUCE": "0"}])
def test_bucket_reduction_variants_forced(gpu_backend, oracle, tmp_path, knobs):
    """k_reduce_compact (what the engine picks on its own only on boxes with slow instruction fetch)
    and the inlined k_reduce, each forced in a process of its own, on columns whose windows span 16
    reduce blocks (the suffix scan's block offset -- one multiple per workgroup on the lane-spread
    form -- is only reached from the second block on): all four curves against the reference"""
    import subprocess
    import sys
    n = 1 << 13
    rng = np.random.default_rng(77)
    arrays, want = {}, {}
    for cid in (0, 1, 2, 3):
        gens = util.generators_for(cid, n)
        arrays[f"g{cid}"] = util.api_generators(cid, gens)
        cols = []
        for k in range(2):
            s = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            s[:, 31] &= 0x0f
            arrays[f"s{cid}_{k}"] = s
            cols.append((s, False))
        want[cid] = oracle.commit(cid, cols, gens)
    src, dst = tmp_path / "in.npz", tmp_path / "out.npz"
    np.savez(src, **arrays)
    env = dict(os.environ, **knobs)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _FORCED_KNOB_SCRIPT, root, str(src), str(dst)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(dst)
    for cid in (0, 1, 2, 3):
        assert np.array_equal(got[f"c{cid}"], want[cid]), f"curve {cid} under {knobs}"

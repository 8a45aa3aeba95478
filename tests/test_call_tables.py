"""Per-call window tables: a call of many columns over the SAME caller generators -- the reference's
bucket_method2 regime (sxt/multiexp/bucket_method2/multiexponentiation.h:48-121, sum.h:41-72,
reduce.h:50-78: 256 <= n <= 4096, hundreds of outputs) -- builds the 2^(c w) multiples of its
generators inside the call and runs every column as ONE task with one bucket set.  The planner's
choice on the host (msm/plan.h, choose_call_table), and on the GPU the commitments against the
reference CPU backend with tables of every width the regime uses, on all four curves."""
import numpy as np
import pytest

from tests import hooks, util


def test_choice_follows_the_cost_model():
    # 1024 x 4096 x 256-bit (missing 3): separate 9-bit windows cost
    # 29 x (4096 + 3.5 x 256 + 800) per column; one merged task at c = 12..13 about 42 % less
    (stride, windows, bits), (separate, merged, build) = hooks.choose_call_table(
        [4096] * 1024, [256] * 1024, [0] * 1024)
    assert stride == 4096 and bits in (12, 13) and windows == -(-257 // bits)
    assert separate == pytest.approx(1024 * 29 * (4096 + 3.5 * 256 + 800))
    assert merged == pytest.approx(1024 * (windows * 4096 * 1.03 + 3.5 * 2 ** (bits - 1) + 800))
    assert merged + build < 0.65 * separate and build < 0.05 * separate
    # 1024 rows: narrower table; the Z = 1 discount of curve25519 does not change the width much
    (stride, windows, bits), _ = hooks.choose_call_table([1024] * 1024, [256] * 1024, [0] * 1024,
                                                         addend_size=128, entry_cost=0.88)
    assert stride == 1024 and 9 <= bits <= 11
    # a handful of columns, long columns, signed columns, narrow columns: no table
    assert hooks.choose_call_table([4096] * 4, [256] * 4, [0] * 4)[0][1] == 0
    assert hooks.choose_call_table([1 << 20] * 256, [256] * 256, [0] * 256)[0][1] == 0
    assert hooks.choose_call_table([4096] * 512, [128] * 512, [1] * 512)[0][1] == 0
    assert hooks.choose_call_table([4096] * 512, [8] * 512, [0] * 512)[0][1] == 0
    # mixed call: the 256-bit columns merge, the table covers the longest column, rows padded to 8
    ns = [4093] * 600 + [4093] * 100 + [700] * 50
    bw = [256] * 600 + [8] * 100 + [256] * 50
    (stride, windows, bits), _ = hooks.choose_call_table(ns, bw, [0] * len(ns))
    assert stride == 4096 and windows == -(-257 // bits) and bits >= 11
    # forced width (tests, A/B runs): whatever the model says
    (stride, windows, bits), _ = hooks.choose_call_table([300] * 3, [256] * 3, [0] * 3, force_bits=10)
    assert (stride, windows, bits) == (304, 26, 10)


def test_plan_with_a_narrow_table_merges_what_it_can():
    """make_msm_plan over a 12-bit table: 256-bit columns of at least half the stride become one task
    of 22 x stride virtual rows and 2048 buckets; short, narrow and signed columns keep separate
    windows over slice 0"""
    ns = [4096, 4000, 1000, 4096, 4096, 0]
    bw = [256, 256, 256, 8, 64, 256]
    sg = [0, 0, 0, 0, 1, 0]
    per, totals = hooks.plan_tables(ns, bw, sg, 4096, 22, force=True, bits=12)
    assert list(per[0][:4]) == [12, 22, 1, 4096] and per[0][4] == 21 * 4096 + 4096
    assert list(per[1][:4]) == [12, 22, 1, 4096] and per[1][4] == 21 * 4096 + 4000
    assert per[2][2] == per[2][1] > 1 and per[2][3] == 0     # too short for the slices
    assert per[3][3] == 0 and per[4][3] == 0 and per[5][1] == 0
    assert totals[6] == 0                                    # 16-bit digits


@pytest.mark.gpu
@pytest.mark.parametrize("curve_id,bits", [(0, 10), (0, 13), (1, 12), (2, 12), (2, 14), (3, 10), (3, 13),
                                           (1, 14), (0, 6), (2, 16)])
def test_forced_call_tables_match_the_reference(gpu_backend, oracle, curve_id, bits):
    """tables of width `bits` forced for every call with caller generators: full-width, short,
    narrow, signed, ragged and empty columns in one call, and generator 5 = the identity on the
    Weierstrass curves; bit-exact against the reference CPU backend"""
    api = gpu_backend
    lib = api.load()
    rng = np.random.default_rng(100 * curve_id + bits)
    n = 777
    gens = util.generators_for(curve_id, n)
    g = util.api_generators(curve_id, gens)
    cols = [(rng.integers(0, 256, (n, 32), dtype=np.uint8), False) for _ in range(5)]
    cols += [(rng.integers(0, 256, (n - 3, 32), dtype=np.uint8), False),
             (rng.integers(0, 256, (n // 2 + 1, 31), dtype=np.uint8), False),
             (rng.integers(0, 256, (100, 32), dtype=np.uint8), False),     # too short: separate windows
             (rng.integers(0, 256, (n, 1), dtype=np.uint8), False),
             (rng.integers(0, 256, (n, 16), dtype=np.uint8), True),
             (np.full((n, 32), 0xff, np.uint8), False),
             (np.zeros((n, 32), np.uint8), False),
             (np.zeros((0, 32), np.uint8), False),
             (rng.integers(0, 256, (1, 32), dtype=np.uint8), False)]
    want = oracle.commit(curve_id, cols, gens)
    before = lib.bzamd_set_call_tables(bits)
    try:
        got = api.compute_pedersen_commitments(curve_id, cols, generators=g)
        assert lib.bzamd_set_call_tables(-1) == before + 1, "no table was built"
    finally:
        lib.bzamd_set_call_tables(0)
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_model_chosen_call_tables(gpu_backend, oracle):
    """the default: 96 columns x 1024 rows take a table (chosen by the model), the same call with the
    tables switched off gives the same bytes, and a call of three columns builds none"""
    api = gpu_backend
    lib = api.load()
    rng = np.random.default_rng(96)
    for curve_id in (0, 2):
        n = 1024
        gens = util.generators_for(curve_id, n)
        g = util.api_generators(curve_id, gens)
        cols = [(rng.integers(0, 256, (n, 32), dtype=np.uint8), False) for _ in range(96)]
        cols.append((rng.integers(0, 256, (n, 4), dtype=np.uint8), True))
        built = lib.bzamd_set_call_tables(-1)
        got = api.compute_pedersen_commitments(curve_id, cols, generators=g)
        assert lib.bzamd_set_call_tables(-1) == built + 1
        lib.bzamd_set_call_tables(1)
        try:
            plain = api.compute_pedersen_commitments(curve_id, cols, generators=g)
            assert lib.bzamd_set_call_tables(-1) == built + 1
        finally:
            lib.bzamd_set_call_tables(0)
        assert np.array_equal(got, plain)
        # (the oracle on a sample of the columns: 97 x 1024 rows would take a minute)
        pick = [0, 17, 95, 96]
        assert np.array_equal(got[pick], oracle.commit(curve_id, [cols[i] for i in pick], gens))
        api.compute_pedersen_commitments(curve_id, cols[:3], generators=g)
        assert lib.bzamd_set_call_tables(-1) == built + 1


@pytest.mark.gpu
def test_call_tables_in_a_sequence_and_on_device_pointers(gpu_backend, oracle):
    """bzamd_msm_device with caller generators in device memory, three calls back to back on one
    stream with DIFFERENT generator sets of different sizes (the table block is rebuilt per call and
    grows), with and without the side-stream overlap of the build"""
    import ctypes

    import torch
    api = gpu_backend
    lib = api.load()
    rng = np.random.default_rng(5)
    lib.bzamd_set_call_tables(11)
    try:
        results, wants = [], []
        keep = []
        for n in (300, 1500, 640):
            gens = util.generators_for(0, n)
            g = torch.from_numpy(np.ascontiguousarray(util.api_generators(0, gens))).cuda()
            cols = [rng.integers(0, 256, (n, 32), dtype=np.uint8) for _ in range(4)]
            d_cols = [torch.from_numpy(c).cuda() for c in cols]
            desc = (api.sxt_sequence_descriptor * 4)()
            for i, d in enumerate(d_cols):
                desc[i] = api.sxt_sequence_descriptor(32, n, d.data_ptr(), 0)
            out = torch.zeros((4, 32), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            lib.bzamd_msm_device(0, ctypes.c_void_p(out.data_ptr()), 4, desc,
                                 ctypes.c_void_p(g.data_ptr()), None)
            keep.append((g, d_cols, desc))
            results.append(out)
            wants.append(oracle.commit(0, [(c, False) for c in cols], gens))
        torch.cuda.synchronize()
        for out, want in zip(results, wants):
            assert np.array_equal(out.cpu().numpy(), want)
    finally:
        lib.bzamd_set_call_tables(0)

"""tools/mt19937 (input generation for bench.py / the full-size tests): the std::mt19937 byte stream
of the reference benchmarks (benchmark/multi_commitment/benchmark.m.cc:141-156) produced on several
host threads by polynomial jump-ahead must be the serial stream, byte for byte."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import baseline_workloads as wl  # noqa: E402


@pytest.fixture(scope="module")
def mt():
    if wl.mtstream_lib() is None:
        pytest.skip("tools/mt19937/_build/libmtstream.so not built (python __graft_entry__.py)")
    return wl


def test_known_first_outputs():
    # std::mt19937{5489}'s first outputs are 3499211612, 581869302, 3890346734 (the C++ standard
    # pins the 10000th of the default seed; these are the well-known first three): top bytes
    got = wl.mt19937_bytes(3, seed=5489, force_numpy=True)
    assert got.tolist() == [3499211612 >> 24, 581869302 >> 24, 3890346734 >> 24]


def test_threads_and_offsets_reproduce_the_serial_stream(mt):
    n = (1 << 22) + 4321
    serial = mt.mt19937_bytes(n, 0, force_numpy=True)
    assert np.array_equal(mt.mt19937_bytes(n, 0, threads=1), serial)
    assert np.array_equal(mt.mt19937_bytes(n, 0, threads=4), serial)  # pieces of >= 2^20 draws
    assert np.array_equal(mt.mt19937_bytes(1 << 21, 0, skip=1234567, threads=2),
                          serial[1234567:1234567 + (1 << 21)])
    assert np.array_equal(mt.mt19937_bytes(1000, 0, skip=n - 1000), serial[-1000:])
    other = mt.mt19937_bytes(1 << 21, 9, force_numpy=True)
    assert np.array_equal(mt.mt19937_bytes(1 << 21, 9, threads=2), other)
    assert np.array_equal(mt.mt19937_bytes(1 << 21, 9, boolean=True, threads=2),
                          mt.mt19937_bytes(1 << 21, 9, boolean=True, force_numpy=True))


def test_column_shards_are_pieces_of_one_stream(mt):
    whole = mt.mt19937_scalars(6, 1 << 16, 32, top_mask=0x0f)
    part = mt.mt19937_scalars(2, 1 << 16, 32, top_mask=0x0f, first_column=3)
    assert np.array_equal(part, whole[3:5])

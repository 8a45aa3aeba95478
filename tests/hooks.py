"""ctypes view of tests/native/_build/libbz_hooks.so: the product's header-only field / curve /
recoding / planning code compiled for the host, so CPU tests can compare it limb-for-limb with
the reference oracle.  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "hooks.cpp")
LIB = os.path.join(ROOT, "tests", "native", "_build", "libbz_hooks.so")

_lib = None


def _newest_header():
    newest = os.path.getmtime(SRC)
    for d, _, fs in os.walk(os.path.join(ROOT, "blitzar_amd", "csrc")):
        for f in fs:
            if f.endswith(".h"):
                newest = max(newest, os.path.getmtime(os.path.join(d, f)))
    return newest


def lib():
    global _lib
    if _lib is None:
        have_cxx = subprocess.run(["which", "g++"], capture_output=True).returncode == 0
        if have_cxx and (not os.path.exists(LIB) or os.path.getmtime(LIB) < _newest_header()):
            os.makedirs(os.path.dirname(LIB), exist_ok=True)
            subprocess.run(["g++", "-O2", "-std=c++20", "-fPIC", "-shared", "-I" + ROOT, SRC, "-o",
                            LIB], check=True)
        _lib = ctypes.CDLL(LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype=np.uint64):
    return np.ascontiguousarray(a, dtype=dtype)


def f51(op, *args):
    out = np.zeros(5, np.uint64)
    getattr(lib(), f"bz_f51_{op}")(_p(out), *[_p(_c(a)) for a in args])
    return out


def ed_base_elements(first, n):
    out = np.zeros((n, 20), np.uint64)
    lib().bz_ed_base_elements(_p(out), ctypes.c_uint64(first), ctypes.c_uint64(n))
    return out


def ed_add(a, b):
    out = np.zeros(20, np.uint64)
    lib().bz_ed_add(_p(out), _p(_c(a)), _p(_c(b)))
    return out


def ed_sub(a, b):
    out = np.zeros(20, np.uint64)
    lib().bz_ed_sub_cached(_p(out), _p(_c(a)), _p(_c(b)))
    return out


def ed_dbl(a, k=None):
    out = np.zeros(20, np.uint64)
    if k is None:
        lib().bz_ed_dbl(_p(out), _p(_c(a)))
    else:
        lib().bz_ed_dbl_n(_p(out), _p(_c(a)), ctypes.c_int(k))
    return out


def ed_neg(a):
    out = np.zeros(20, np.uint64)
    lib().bz_ed_neg(_p(out), _p(_c(a)))
    return out


def ristretto_encode(p):
    out = np.zeros(32, np.uint8)
    lib().bz_ristretto_encode(_p(out), _p(_c(p)))
    return out


def ristretto_decode(s):
    out = np.zeros(20, np.uint64)
    rc = lib().bz_ristretto_decode(_p(out), _p(_c(s, np.uint8)))
    assert rc == 0, "not a canonical ristretto255 encoding"
    return out


PFX = {1: "bls12_381", 2: "bn254", 3: "grumpkin"}
LIMBS = {1: 6, 2: 4, 3: 4}


def sw_field_mul(cid, f, g):
    out = np.zeros(LIMBS[cid], np.uint64)
    getattr(lib(), f"bz_{PFX[cid]}_field_mul")(_p(out), _p(_c(f)), _p(_c(g)))
    return out


def sw_field_addsub(cid, f, g):
    s = np.zeros(LIMBS[cid], np.uint64)
    d = np.zeros(LIMBS[cid], np.uint64)
    getattr(lib(), f"bz_{PFX[cid]}_field_addsub")(_p(s), _p(d), _p(_c(f)), _p(_c(g)))
    return s, d


def sw_add(cid, a, b):
    out = np.zeros(3 * LIMBS[cid], np.uint64)
    getattr(lib(), f"bz_{PFX[cid]}_add")(_p(out), _p(_c(a)), _p(_c(b)))
    return out


def sw_add_mixed(cid, a, b_affine_xy):
    out = np.zeros(3 * LIMBS[cid], np.uint64)
    getattr(lib(), f"bz_{PFX[cid]}_add_mixed")(_p(out), _p(_c(a)), _p(_c(b_affine_xy)))
    return out


def sw_dbl(cid, a):
    out = np.zeros(3 * LIMBS[cid], np.uint64)
    getattr(lib(), f"bz_{PFX[cid]}_dbl")(_p(out), _p(_c(a)))
    return out


def sw_to_affine(cid, a):
    out = np.zeros(2 * LIMBS[cid], np.uint64)
    inf = getattr(lib(), f"bz_{PFX[cid]}_to_affine")(_p(out), _p(_c(a)))
    return out, bool(inf)


def bls_compress(a):
    out = np.zeros(48, np.uint8)
    lib().bz_bls12_381_compress(_p(out), _p(_c(a)))
    return out


def recode(row_bytes, bit_offset, bit_width, is_signed, window_bits, num_windows):
    row = np.zeros(len(row_bytes) + 40, np.uint8)  # the recoder may read a few bytes past the field
    row[:len(row_bytes)] = row_bytes
    digits = np.zeros(num_windows, np.int32)
    lib().bz_recode(_p(digits), _p(row), ctypes.c_uint32(bit_offset), ctypes.c_uint32(bit_width),
                    ctypes.c_int(1 if is_signed else 0), ctypes.c_uint32(window_bits),
                    ctypes.c_uint32(num_windows))
    return digits


def recode_words32(row_bytes, skew, bit_offset, bit_width, is_signed, window_bits, num_windows):
    row = np.zeros(len(row_bytes) + 48, np.uint8)
    row[:len(row_bytes)] = row_bytes
    digits = np.zeros(num_windows, np.int32)
    lib().bz_recode_words32(_p(digits), _p(row), ctypes.c_uint32(skew), ctypes.c_uint32(bit_offset),
                            ctypes.c_uint32(bit_width), ctypes.c_int(1 if is_signed else 0),
                            ctypes.c_uint32(window_bits), ctypes.c_uint32(num_windows))
    return digits


def plan(ns, bit_widths, signed, max_window_bits=16, in_sequence=False):
    k = len(ns)
    per = np.zeros((k, 6), np.uint32)
    totals = np.zeros(8, np.uint64)
    lib().bz_plan(_p(per), _p(totals), _p(_c(ns)), _p(_c(bit_widths, np.uint32)),
                  _p(_c(signed, np.int32)), ctypes.c_uint32(k), ctypes.c_uint32(max_window_bits),
                  ctypes.c_int(1 if in_sequence else 0))
    return per, totals


def plan_tables(ns, bit_widths, signed, stride, windows, force=False, table_penalty=1.0, bits=16):
    k = len(ns)
    per = np.zeros((k, 6), np.uint32)
    totals = np.zeros(9, np.uint64)
    lib().bz_plan_tables(_p(per), _p(totals), _p(_c(ns)), _p(_c(bit_widths, np.uint32)),
                         _p(_c(signed, np.int32)), ctypes.c_uint32(k), ctypes.c_uint64(stride),
                         ctypes.c_uint32(windows), ctypes.c_int(1 if force else 0),
                         ctypes.c_double(table_penalty), ctypes.c_uint32(bits))
    return per, totals


def choose_call_table(ns, bit_widths, signed, addend_size=64, entry_cost=1.0, force_bits=0):
    """plan.h choose_call_table -> (stride, windows, bits), (separate, merged, build) costs"""
    out = np.zeros(3, np.uint64)
    costs = np.zeros(3, np.float64)
    fn = lib().bz_choose_call_table
    fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_uint32, ctypes.c_uint64, ctypes.c_double, ctypes.c_uint32]
    fn.restype = None
    fn(_p(out), _p(costs), _p(_c(ns)), _p(_c(bit_widths, np.uint32)), _p(_c(signed, np.int32)),
       len(ns), addend_size, entry_cost, force_bits)
    return tuple(int(v) for v in out), tuple(float(v) for v in costs)


def packed_ranges(offsets, strides, bit_offsets, bit_widths):
    """column ranges k_recode_packed would use: list of (first_column, num_columns, base, span)"""
    k = len(offsets)
    out = np.zeros((k, 4), np.uint32)
    n = lib().bz_packed_ranges(_p(out), _p(_c(offsets)), _p(_c(strides)),
                               _p(_c(bit_offsets, np.uint32)), _p(_c(bit_widths, np.uint32)),
                               ctypes.c_uint32(k))
    return [tuple(int(v) for v in row) for row in out[:n]]


def f29_from_fe51(f):
    out = np.zeros(9, np.uint32)
    lib().bz_f29_from_fe51(_p(out), _p(_c(f)))
    return out


def f29_to_int(f):
    w = np.zeros(4, np.uint64)
    lib().bz_f29_to_words(_p(w), _p(_c(f, np.uint32)))
    return int.from_bytes(w.tobytes(), "little")


def f29(op, *args):
    out = np.zeros(9, np.uint32)
    getattr(lib(), f"bz_f29_{op}")(_p(out), *[_p(_c(a, np.uint32)) for a in args])
    return out


def ed29_add(a, b, negate=False):
    out = np.zeros(20, np.uint64)
    lib().bz_ed29_add(_p(out), _p(_c(a)), _p(_c(b)), ctypes.c_int(1 if negate else 0))
    return out


def ed29_add_gathered(a, b, negate=False):
    out = np.zeros(20, np.uint64)
    lib().bz_ed29_add_gathered(_p(out), _p(_c(a)), _p(_c(b)), ctypes.c_int(1 if negate else 0))
    return out


def ed29_dbl_n(a, k):
    out = np.zeros(20, np.uint64)
    lib().bz_ed29_dbl_n(_p(out), _p(_c(a)), ctypes.c_int(k))
    return out


def ed29_chain(points, negate, niels=False):
    out = np.zeros(20, np.uint64)
    pts = _c(points)
    neg = _c(negate, np.int32)
    fn = lib().bz_ed29_chain_niels if niels else lib().bz_ed29_chain
    fn(_p(out), _p(pts), _p(neg), ctypes.c_int(pts.shape[0]))
    return out


def sw29_field(cid, op, *args):
    out = np.zeros(LIMBS[cid], np.uint64)
    getattr(lib(), f"bz_{PFX[cid]}_29_field_{op}")(_p(out), *[_p(_c(a)) for a in args])
    return out


def sw29_add(cid, a, b):
    out = np.zeros(3 * LIMBS[cid], np.uint64)
    getattr(lib(), f"bz_{PFX[cid]}_29_add")(_p(out), _p(_c(a)), _p(_c(b)))
    return out


def sw29_dbl_n(cid, a, k):
    out = np.zeros(3 * LIMBS[cid], np.uint64)
    getattr(lib(), f"bz_{PFX[cid]}_29_dbl_n")(_p(out), _p(_c(a)), ctypes.c_int(k))
    return out


def ed29_chain_first(points, negate, niels=False):
    """a lane of k_accumulate: first entry loaded (one product), the others added"""
    out = np.zeros(20, np.uint64)
    pts = _c(points)
    neg = _c(negate, np.int32)
    lib().bz_ed29_chain_first(_p(out), _p(pts), _p(neg), ctypes.c_int(pts.shape[0]),
                              ctypes.c_int(1 if niels else 0))
    return out


def sw29_chain_lifted(cid, affine_xy, negate):
    """a lane of k_accumulate: first affine entry lifted (Z = 1), the others added (add_mixed_acc)"""
    out = np.zeros(3 * LIMBS[cid], np.uint64)
    xy = _c(affine_xy)
    neg = _c(negate, np.int32)
    getattr(lib(), f"bz_{PFX[cid]}_29_chain_lifted")(_p(out), _p(xy), _p(neg),
                                                     ctypes.c_int(xy.shape[0]))
    return out


def sw29_chain(cid, start, affine_xy, negate):
    out = np.zeros(3 * LIMBS[cid], np.uint64)
    xy = _c(affine_xy)
    neg = _c(negate, np.int32)
    getattr(lib(), f"bz_{PFX[cid]}_29_chain")(_p(out), _p(_c(start)), _p(xy), _p(neg),
                                              ctypes.c_int(xy.shape[0]))
    return out

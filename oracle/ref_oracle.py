"""TEST INFRASTRUCTURE ONLY: ctypes view of oracle/_ref/libblitzar_ref.so.

That library is the reference's *own* CPU backend code (mtxcrv::compute_multiexponentiation +
host canonicalisers), compiled from /root/reference by oracle/ref/build_ref.py.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(blitzar_amd/) never does.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libblitzar_ref.so")

# prefix, limbs per field element, affine stride, commitment bytes
CURVES = {
    0: ("c25519", 5, 160, 32),
    1: ("bls12_381", 6, 104, 48),
    2: ("bn254", 4, 72, 72),
    3: ("grumpkin", 4, 72, 72),
}


class _Desc(ctypes.Structure):
    _fields_ = [("nbytes", ctypes.c_uint8), ("n", ctypes.c_uint64), ("data", ctypes.c_void_p),
                ("is_signed", ctypes.c_int)]


def available():
    return os.path.exists(LIB_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError(f"{LIB_PATH} missing: run `python oracle/ref/build_ref.py` where "
                               "/root/reference is mounted")
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _descs(columns):
    cols = list(columns)
    ds = (_Desc * max(1, len(cols)))()
    keep = []
    for i, (arr, is_signed) in enumerate(cols):
        arr = np.ascontiguousarray(arr)
        nbytes = arr.dtype.itemsize if arr.ndim == 1 else arr.shape[1]
        keep.append(arr)
        ds[i] = _Desc(nbytes, arr.shape[0], arr.ctypes.data if arr.shape[0] else None,
                      1 if is_signed else 0)
    return ds, keep


def ristretto_generators(n, first=0):
    out = np.zeros((n, 20), dtype=np.uint64)
    if n:
        lib().ref_c25519_base_elements(_p(out), ctypes.c_uint64(first), ctypes.c_uint64(n))
    return out


def one_commit(n):
    out = np.zeros(20, dtype=np.uint64)
    lib().ref_c25519_one_commit(_p(out), ctypes.c_uint64(n))
    return out


def commit(curve_id, columns, generators):
    """== cpu_backend::compute_commitments: canonical encodings, one row per column.
    `generators`: curve25519 -> uint64 [n, 20] element_p3; others -> uint8 [n, stride] affine."""
    pfx, _, _, osz = CURVES[curve_id]
    ds, keep = _descs(columns)
    out = np.zeros((len(keep), osz), dtype=np.uint8)
    if keep:
        g = np.ascontiguousarray(generators)
        getattr(lib(), f"ref_{pfx}_commit")(_p(out), len(keep), ds, _p(g))
    return out


def msm_projective(curve_id, columns, generators):
    pfx, nl, _, _ = CURVES[curve_id]
    ds, keep = _descs(columns)
    words = 20 if curve_id == 0 else 3 * nl
    out = np.zeros((len(keep), words), dtype=np.uint64)
    g = np.ascontiguousarray(generators)
    fn = "ref_c25519_msm_p3" if curve_id == 0 else f"ref_{pfx}_msm_p2"
    getattr(lib(), fn)(_p(out), len(keep), ds, _p(g))
    return out


def random_affine(curve_id, seed1, seed2):
    """cn1rn/cg1rn/cgkrn::generate_random_element(rng{seed1, seed2}) in C-ABI affine layout."""
    pfx, _, stride, _ = CURVES[curve_id]
    out = np.zeros(stride, dtype=np.uint8)
    getattr(lib(), f"ref_{pfx}_random_affine")(_p(out), ctypes.c_uint64(seed1),
                                               ctypes.c_uint64(seed2))
    return out


def identity_affine(curve_id):
    """{0, R, infinity = 1} in C-ABI affine layout."""
    pfx, nl, stride, _ = CURVES[curve_id]
    ident = np.zeros(3 * nl, dtype=np.uint64)
    ident[nl] = 1  # any (0 : y : 0) is the identity
    out = np.zeros(stride, dtype=np.uint8)
    getattr(lib(), f"ref_{pfx}_to_affine")(_p(out), _p(ident))
    return out


def to_affine(curve_id, p2):
    pfx, _, stride, _ = CURVES[curve_id]
    out = np.zeros(stride, dtype=np.uint8)
    getattr(lib(), f"ref_{pfx}_to_affine")(_p(out), _p(np.ascontiguousarray(p2)))
    return out


def add_projective(curve_id, a, b):
    pfx, nl, _, _ = CURVES[curve_id]
    if curve_id == 0:
        out = np.zeros(20, dtype=np.uint64)
        lib().ref_c25519_add(_p(out), _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)))
        return out
    out = np.zeros(3 * nl, dtype=np.uint64)
    getattr(lib(), f"ref_{pfx}_add_p2")(_p(out), _p(np.ascontiguousarray(a)),
                                         _p(np.ascontiguousarray(b)))
    return out


def double_projective(curve_id, a):
    pfx, nl, _, _ = CURVES[curve_id]
    if curve_id == 0:
        out = np.zeros(20, dtype=np.uint64)
        lib().ref_c25519_double(_p(out), _p(np.ascontiguousarray(a)))
        return out
    out = np.zeros(3 * nl, dtype=np.uint64)
    getattr(lib(), f"ref_{pfx}_double_p2")(_p(out), _p(np.ascontiguousarray(a)))
    return out


def affine_to_projective(curve_id, affine_bytes):
    """C-ABI affine generator(s) -> projective element_p2 words (Z = R, identity = (0, R, 0))"""
    _, nl, stride, _ = CURVES[curve_id]
    a = np.ascontiguousarray(affine_bytes, dtype=np.uint8).reshape(-1, stride)
    one = identity_affine(curve_id)[8 * nl:16 * nl].view(np.uint64)
    out = np.zeros((a.shape[0], 3 * nl), dtype=np.uint64)
    for i in range(a.shape[0]):
        if a[i, 16 * nl]:
            out[i, nl:2 * nl] = one
        else:
            out[i, :2 * nl] = a[i, :16 * nl].view(np.uint64)
            out[i, 2 * nl:] = one
    return out


def ristretto_compress(p3):
    out = np.zeros(32, dtype=np.uint8)
    lib().ref_c25519_compress(_p(out), _p(np.ascontiguousarray(p3)))
    return out


def bls_compress(p2):
    out = np.zeros(48, dtype=np.uint8)
    lib().ref_bls12_381_compress(_p(out), _p(np.ascontiguousarray(p2)))
    return out


def canonical(curve_id, projective):
    """canonical bytes of a projective element (what the parity harness compares)."""
    if curve_id == 0:
        return ristretto_compress(projective)
    if curve_id == 1:
        return bls_compress(projective)
    return to_affine(curve_id, projective)


def partition_table(curve_id, window_width, generators_projective):
    """reference compute_partition_table: (n / w) * 2^w compact elements as raw bytes."""
    pfx, nl, _, _ = CURVES[curve_id]
    g = np.ascontiguousarray(generators_projective)
    n = g.shape[0]
    assert n % window_width == 0
    entry = 120 if curve_id == 0 else 16 * nl
    out = np.zeros(((n // window_width) << window_width) * entry, dtype=np.uint8)
    getattr(lib(), f"ref_{pfx}_partition_table")(_p(out), ctypes.c_uint(window_width), _p(g),
                                                 ctypes.c_uint(n))
    return out

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


def generator_chain(curve_id, start_p2, step_p2, count):
    """affine(start + (i + 1) step) for i < count, C-ABI affine layout (ref_driver.cc: the
    reference's add + to_element_affine in a loop; releases the GIL, so segments of a long chain
    can run on host threads)"""
    pfx, _, stride, _ = CURVES[curve_id]
    out = np.zeros((count, stride), dtype=np.uint8)
    start = np.ascontiguousarray(start_p2, dtype=np.uint64).copy()
    step = np.ascontiguousarray(step_p2, dtype=np.uint64)
    getattr(lib(), f"ref_{pfx}_generator_chain")(_p(out), _p(start), _p(step),
                                                  ctypes.c_uint64(count))
    return out, start


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


#--------------------------------------------------------------------------------------------------
# fixed-base MSM (oracle/ref/ref_fixed_base.cc: the reference's own host path -- its partition-table
# accessor, mtxpp2::multiexponentiate, partition_product and reduce_products -- compiled in place)
#--------------------------------------------------------------------------------------------------
_PROJ_WORDS = {0: 20, 1: 18, 2: 12, 3: 12}


class FixedHandle:
    """what cpu_backend keeps behind a sxt_multiexp_handle (cpu_backend.cc:196-211): an
    in_memory_partition_table_accessor over the reference's own table"""

    def __init__(self, curve_id, generators_projective=None, window_width=16, filename=None):
        self.curve_id = curve_id
        L = lib()
        L.ref_fixed_handle_new.restype = ctypes.c_void_p
        L.ref_fixed_handle_from_file.restype = ctypes.c_void_p
        if filename is not None:
            self._h = ctypes.c_void_p(L.ref_fixed_handle_from_file(ctypes.c_uint(curve_id),
                                                                    filename.encode()))
            return
        g = np.ascontiguousarray(generators_projective, dtype=np.uint64).reshape(
            -1, _PROJ_WORDS[curve_id])
        self.n = g.shape[0]
        self._h = ctypes.c_void_p(L.ref_fixed_handle_new(ctypes.c_uint(curve_id), _p(g),
                                                          ctypes.c_uint(self.n),
                                                          ctypes.c_uint(window_width)))

    def write_to_file(self, filename):
        lib().ref_fixed_handle_write(self._h, filename.encode())

    def close(self):
        if self._h is not None:
            lib().ref_fixed_handle_free(self._h)
            self._h = None

    def _out(self, num_outputs):
        return np.zeros((num_outputs, _PROJ_WORDS[self.curve_id]), dtype=np.uint64)

    @staticmethod
    def _fn(name, gpu_flow):
        # gpu_flow: the reference's GPU control flow (async_multiexponentiate + scheduler) executed
        # on the host stand-ins instead of its host loop -- see ref_fixed_base.cc
        return getattr(lib(), name + ("_async" if gpu_flow else ""))

    def multiexponentiation(self, element_num_bytes, num_outputs, n, scalars, gpu_flow=False):
        s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1)
        assert s.size == element_num_bytes * num_outputs * n
        res = self._out(num_outputs)
        self._fn("ref_fixed_multiexponentiation", gpu_flow)(_p(res), self._h, ctypes.c_uint(element_num_bytes),
                                            ctypes.c_uint(num_outputs), ctypes.c_uint(n), _p(s))
        return res

    def packed_multiexponentiation(self, bit_table, n, scalars, gpu_flow=False):
        bt = np.ascontiguousarray(bit_table, dtype=np.uint32)
        s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1)
        assert s.size == (int(bt.sum()) + 7) // 8 * n
        res = self._out(bt.size)
        self._fn("ref_fixed_packed_multiexponentiation", gpu_flow)(
            _p(res), self._h, _p(bt), ctypes.c_uint(bt.size), ctypes.c_uint(n), _p(s))
        return res

    def vlen_multiexponentiation(self, bit_table, lengths, scalars, gpu_flow=False):
        bt = np.ascontiguousarray(bit_table, dtype=np.uint32)
        ln = np.ascontiguousarray(lengths, dtype=np.uint32)
        s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1)
        assert s.size >= (int(bt.sum()) + 7) // 8 * int(ln.max(initial=0))
        res = self._out(bt.size)
        self._fn("ref_fixed_vlen_multiexponentiation", gpu_flow)(
            _p(res), self._h, _p(bt), _p(ln), ctypes.c_uint(bt.size), _p(s))
        return res


    # BLITZAR_DUMP_DIR recordings through the reference's own writer / reader
    # (multiexponentiation_serialization.h:70-151; ref_fixed_base.cc)
    def write_dump(self, directory, bit_table, scalars, lengths=None, n=None):
        """what gpu_backend.cc:286-301 / :317-332 records BEFORE the computation (result.bin -- the raw
        result elements, :298-300 -- is the caller's to add)"""
        bt = np.ascontiguousarray(bit_table, dtype=np.uint32)
        s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1)
        if lengths is None:
            lib().ref_write_packed_multiexponentiation(directory.encode(), self._h, _p(bt),
                                                       ctypes.c_uint(bt.size), ctypes.c_uint(n), _p(s))
        else:
            ln = np.ascontiguousarray(lengths, dtype=np.uint32)
            lib().ref_write_vlen_multiexponentiation(directory.encode(), self._h, _p(bt), _p(ln),
                                                     ctypes.c_uint(bt.size), _p(s))

    @classmethod
    def read_dump(cls, curve_id, directory, vlen=False):
        """-> (handle over the accessor the reference rebuilt from generators.bin + window_width.bin,
        output_bit_table, output_lengths or None, scalars)"""
        L = lib()
        L.ref_read_multiexponentiation.restype = ctypes.c_void_p
        sizes = np.zeros(3, np.uint64)
        probe = ctypes.c_void_p(L.ref_read_multiexponentiation(
            directory.encode(), ctypes.c_uint(curve_id), ctypes.c_int(int(vlen)), _p(sizes), None,
            None, None))
        L.ref_fixed_handle_free(probe)
        bt = np.zeros(int(sizes[0]), np.uint32)
        ln = np.zeros(int(sizes[1]), np.uint32)
        sc = np.zeros(int(sizes[2]), np.uint8)
        self = cls.__new__(cls)
        self.curve_id = curve_id
        self._h = ctypes.c_void_p(L.ref_read_multiexponentiation(
            directory.encode(), ctypes.c_uint(curve_id), ctypes.c_int(int(vlen)), _p(sizes), _p(bt),
            _p(ln) if vlen else None, _p(sc)))
        return self, bt, (ln if vlen else None), sc


#--------------------------------------------------------------------------------------------------
# inner-product argument (oracle/ref/ref_inner_product.cc: the reference's own prover / verifier)
#--------------------------------------------------------------------------------------------------
def transcript_new(label):
    """prft::transcript{label}: the 203-byte Merlin state handed to the C API"""
    out = np.zeros(203, dtype=np.uint8)
    raw = label.encode() if isinstance(label, str) else bytes(label)
    lib().ref_transcript_new(_p(out), ctypes.c_char_p(raw), ctypes.c_uint64(len(raw)))
    return out


def _rounds(n):
    return max(int(n) - 1, 0).bit_length()


def ip_prove(transcript, n, generators_offset, a_vector, b_vector):
    """-> (l_vector [rounds, 32], r_vector [rounds, 32], ap_value [32], transcript after)"""
    t = np.ascontiguousarray(transcript, dtype=np.uint8).copy()
    a = np.ascontiguousarray(a_vector, dtype=np.uint8).reshape(n, 32)
    b = np.ascontiguousarray(b_vector, dtype=np.uint8).reshape(n, 32)
    rounds = _rounds(n)
    l = np.zeros((max(rounds, 1), 32), dtype=np.uint8)
    r = np.zeros((max(rounds, 1), 32), dtype=np.uint8)
    ap = np.zeros(32, dtype=np.uint8)
    lib().ref_ip_prove(_p(l), _p(r), _p(ap), _p(t), ctypes.c_uint64(n),
                       ctypes.c_uint64(generators_offset), _p(a), _p(b))
    return l[:rounds], r[:rounds], ap, t


def ip_verify(transcript, n, generators_offset, b_vector, product, a_commit, l_vector, r_vector,
              ap_value):
    t = np.ascontiguousarray(transcript, dtype=np.uint8).copy()
    b = np.ascontiguousarray(b_vector, dtype=np.uint8).reshape(n, 32)
    lv = np.ascontiguousarray(l_vector, dtype=np.uint8).reshape(-1, 32)
    rv = np.ascontiguousarray(r_vector, dtype=np.uint8).reshape(-1, 32)
    if lv.shape[0] == 0:
        lv = np.zeros((1, 32), np.uint8)
        rv = np.zeros((1, 32), np.uint8)
    lib().ref_ip_verify.restype = ctypes.c_int
    rc = lib().ref_ip_verify(_p(t), ctypes.c_uint64(n), ctypes.c_uint64(generators_offset), _p(b),
                             _p(np.ascontiguousarray(product, dtype=np.uint8)),
                             _p(np.ascontiguousarray(a_commit, dtype=np.uint64)), _p(lv), _p(rv),
                             _p(np.ascontiguousarray(ap_value, dtype=np.uint8)))
    return bool(rc), t


def s25_inner_product(a_vector, b_vector):
    a = np.ascontiguousarray(a_vector, dtype=np.uint8).reshape(-1, 32)
    b = np.ascontiguousarray(b_vector, dtype=np.uint8).reshape(-1, 32)
    out = np.zeros(32, dtype=np.uint8)
    lib().ref_s25_inner_product(_p(out), _p(a), _p(b), ctypes.c_uint64(min(a.shape[0], b.shape[0])))
    return out


#--------------------------------------------------------------------------------------------------
# sumcheck (oracle/ref/ref_sumcheck.cc: the reference's own prover with its cpu driver)
#--------------------------------------------------------------------------------------------------
SUMCHECK_CALLBACK = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_uint)


def sumcheck_product_stride(field_id):
    return int(lib().ref_sumcheck_product_stride(ctypes.c_uint(field_id)))


def prove_sumcheck(field_id, mles, product_table, product_terms, n, round_degree, callback):
    """mles: uint8 [num_mles, n, 32] (column-major n x num_mles); product_table: raw bytes of
    num_products x {element; unsigned}; callback(r_ptr, ctx, polynomial_ptr, length)."""
    m = np.ascontiguousarray(mles, dtype=np.uint8)
    num_mles = m.shape[0]
    table = np.ascontiguousarray(product_table, dtype=np.uint8)
    terms = np.ascontiguousarray(product_terms, dtype=np.uint32)
    num_products = table.size // sumcheck_product_stride(field_id)
    num_variables = max((int(n) - 1).bit_length(), 1)
    polys = np.zeros((num_variables, round_degree + 1, 32), dtype=np.uint8)
    point = np.zeros((num_variables, 32), dtype=np.uint8)
    cb = SUMCHECK_CALLBACK(callback)
    lib().ref_prove_sumcheck(_p(polys), _p(point), ctypes.c_uint(field_id), _p(m), _p(table),
                             _p(terms), ctypes.c_uint(n), ctypes.c_uint(num_mles),
                             ctypes.c_uint(num_products), ctypes.c_uint(terms.size),
                             ctypes.c_uint(round_degree), cb, None)
    return polys, point

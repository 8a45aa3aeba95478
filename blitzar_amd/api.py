"""ctypes binding of libblitzar_amd.so -- the host-side mirror of the reference's C API.

Function names, argument order and error behaviour follow `cbindings/blitzar_api.h` of the
reference (declared for this repo in include/blitzar_api.h); the `bzamd_*` functions are the
device-resident extensions of include/blitzar_amd.h.  This module is plumbing for tests and
bench.py: the product is the C-ABI shared library, not this wrapper.  It never computes anything
itself and raises ImportError-like failures loudly when the HIP library is missing.
"""
import ctypes
import os

import numpy as np

SXT_CPU_BACKEND = 1
SXT_GPU_BACKEND = 2
SXT_CURVE_RISTRETTO255 = 0
SXT_CURVE_BLS_381 = 1
SXT_CURVE_BN_254 = 2
SXT_CURVE_GRUMPKIN = 3

_HERE = os.path.dirname(os.path.abspath(__file__))
# BLITZAR_AMD_LIB selects another build of the same library (kernel-variant A/B runs)
LIB_PATH = os.environ.get("BLITZAR_AMD_LIB") or os.path.join(_HERE, "lib", "libblitzar_amd.so")

# per curve: (C-ABI generator stride, commitment bytes, projective element bytes)
CURVE_LAYOUT = {
    SXT_CURVE_RISTRETTO255: (160, 32, 160),
    SXT_CURVE_BLS_381: (104, 48, 144),
    SXT_CURVE_BN_254: (72, 72, 96),
    SXT_CURVE_GRUMPKIN: (72, 72, 96),
}


class sxt_config(ctypes.Structure):
    _fields_ = [("backend", ctypes.c_int), ("num_precomputed_generators", ctypes.c_uint64)]


class sxt_sequence_descriptor(ctypes.Structure):
    _fields_ = [("element_nbytes", ctypes.c_uint8), ("n", ctypes.c_uint64),
                ("data", ctypes.c_void_p), ("is_signed", ctypes.c_int)]


assert ctypes.sizeof(sxt_sequence_descriptor) == 32
assert ctypes.sizeof(sxt_config) == 16

_lib = None


def load():
    """Load the HIP library; fails loudly (no CPU fallback module exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m blitzar_amd.build` "
            "(or __graft_entry__.build()); there is no fallback implementation")
    lib = ctypes.CDLL(LIB_PATH)
    vp, u32, u64, cu = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint
    lib.sxt_init.argtypes = [ctypes.POINTER(sxt_config)]
    lib.sxt_init.restype = ctypes.c_int
    lib.sxt_curve25519_compute_pedersen_commitments.argtypes = [
        vp, u32, ctypes.POINTER(sxt_sequence_descriptor), u64]
    for name in ("sxt_curve25519_compute_pedersen_commitments_with_generators",
                 "sxt_bls12_381_g1_compute_pedersen_commitments_with_generators",
                 "sxt_bn254_g1_uncompressed_compute_pedersen_commitments_with_generators",
                 "sxt_grumpkin_uncompressed_compute_pedersen_commitments_with_generators"):
        getattr(lib, name).argtypes = [vp, u32, ctypes.POINTER(sxt_sequence_descriptor), vp]
        getattr(lib, name).restype = None
    lib.sxt_ristretto255_get_generators.argtypes = [vp, u64, u64]
    lib.sxt_ristretto255_get_generators.restype = ctypes.c_int
    lib.sxt_curve25519_get_one_commit.argtypes = [vp, u64]
    lib.sxt_curve25519_get_one_commit.restype = ctypes.c_int
    lib.sxt_multiexp_handle_new.argtypes = [cu, vp, cu]
    lib.sxt_multiexp_handle_new.restype = vp
    lib.sxt_multiexp_handle_new_from_file.argtypes = [cu, ctypes.c_char_p]
    lib.sxt_multiexp_handle_new_from_file.restype = vp
    lib.sxt_multiexp_handle_write_to_file.argtypes = [vp, ctypes.c_char_p]
    lib.sxt_multiexp_handle_write_to_file.restype = None
    lib.sxt_multiexp_handle_free.argtypes = [vp]
    lib.sxt_multiexp_handle_free.restype = None
    lib.sxt_fixed_multiexponentiation.argtypes = [vp, vp, cu, cu, cu, vp]
    lib.sxt_fixed_multiexponentiation.restype = None
    lib.sxt_fixed_packed_multiexponentiation.argtypes = [vp, vp, vp, cu, cu, vp]
    lib.sxt_fixed_packed_multiexponentiation.restype = None
    lib.sxt_fixed_vlen_multiexponentiation.argtypes = [vp, vp, vp, vp, cu, vp]
    lib.sxt_fixed_vlen_multiexponentiation.restype = None
    lib.bzamd_version.restype = ctypes.c_char_p
    lib.bzamd_device_count.restype = ctypes.c_int
    lib.bzamd_active_backend.restype = ctypes.c_int
    lib.bzamd_kernel_launch_count.restype = ctypes.c_uint64
    lib.bzamd_concurrent_calls_high_water.restype = ctypes.c_uint32
    lib.bzamd_slow_instruction_fetch.restype = ctypes.c_int
    lib.bzamd_probe_mad_rate.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
    lib.bzamd_probe_mad_rate.restype = ctypes.c_int
    lib.bzamd_set_row_pipeline_chunks.argtypes = [u32]
    lib.bzamd_set_row_pipeline_chunks.restype = None
    lib.bzamd_reset_for_testing.restype = None
    lib.bzamd_set_tuning.argtypes = [u32, u64, u64]
    lib.bzamd_set_tuning.restype = None
    lib.bzamd_set_segments.argtypes = [u32, u32]
    lib.bzamd_set_segments.restype = None
    lib.bzamd_stage_timing_begin.argtypes = [u64]
    lib.bzamd_stage_timing_begin.restype = None
    lib.bzamd_pipeline_next.argtypes = []
    lib.bzamd_pipeline_next.restype = None
    lib.bzamd_pipeline_flush.argtypes = [ctypes.c_void_p]
    lib.bzamd_pipeline_flush.restype = None
    lib.bzamd_stage_timing_begin_masked.argtypes = [u64, u32]
    lib.bzamd_stage_timing_begin_masked.restype = None
    lib.bzamd_stage_timing_begin_sampled.argtypes = [u64, u32, u32]
    lib.bzamd_stage_timing_begin_sampled.restype = None
    lib.bzamd_stage_timing_collect.argtypes = [ctypes.POINTER(ctypes.c_double)]
    lib.bzamd_stage_timing_collect.restype = ctypes.c_uint64
    lib.bzamd_msm_device.argtypes = [cu, vp, u32, ctypes.POINTER(sxt_sequence_descriptor), vp, vp]
    lib.bzamd_msm_device.restype = None
    lib.bzamd_msm_device_projective.argtypes = [cu, vp, u32,
                                                ctypes.POINTER(sxt_sequence_descriptor), vp, vp]
    lib.bzamd_msm_device_projective.restype = None
    lib.bzamd_msm_projective.argtypes = [cu, vp, u32, ctypes.POINTER(sxt_sequence_descriptor), vp]
    lib.bzamd_msm_projective.restype = None
    lib.bzamd_fold_encode.argtypes = [cu, vp, vp, u32, u32]
    lib.bzamd_fold_encode.restype = None
    lib.bzamd_fold_encode_device.argtypes = [cu, vp, vp, u32, u32, vp]
    lib.bzamd_fold_encode_device.restype = None
    lib.bzamd_generators_new_device.argtypes = [cu, vp, u64, vp]
    lib.bzamd_generators_new_device.restype = vp
    lib.bzamd_generators_new_host.argtypes = [cu, vp, u64]
    lib.bzamd_generators_new_host.restype = vp
    lib.bzamd_generators_free.argtypes = [vp]
    lib.bzamd_generators_free.restype = None
    lib.bzamd_msm_device_resident.argtypes = [vp, u32, ctypes.POINTER(sxt_sequence_descriptor),
                                              vp, vp]
    lib.bzamd_msm_device_resident.restype = None
    lib.bzamd_generator_multiples_device.argtypes = [cu, vp, vp, u64, vp]
    lib.bzamd_generator_multiples_device.restype = None
    lib.bzamd_ristretto255_generators_device.argtypes = [vp, u64, u64, vp]
    lib.bzamd_ristretto255_generators_device.restype = None
    lib.bzamd_fixed_packed_multiexponentiation_device.argtypes = [vp, vp, vp, vp, cu, cu, vp, vp]
    lib.bzamd_fixed_packed_multiexponentiation_device.restype = None
    lib.sxt_curve25519_prove_inner_product.argtypes = [vp, vp, vp, vp, u64, u64, vp, vp]
    lib.sxt_curve25519_prove_inner_product.restype = None
    lib.sxt_curve25519_verify_inner_product.argtypes = [vp, u64, u64, vp, vp, vp, vp, vp, vp]
    lib.sxt_curve25519_verify_inner_product.restype = ctypes.c_int
    lib.bzamd_transcript_init.argtypes = [vp, ctypes.c_char_p, u64]
    lib.bzamd_transcript_init.restype = None
    lib.bzamd_num_devices.restype = ctypes.c_int
    lib.bzamd_set_window_bits.argtypes = [u32]
    lib.bzamd_set_window_bits.restype = None
    lib.bzamd_generator_cache_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64)] * 2
    lib.bzamd_generator_cache_stats.restype = None
    lib.bzamd_set_call_tables.argtypes = [ctypes.c_int]
    lib.bzamd_set_call_tables.restype = ctypes.c_uint64
    lib.bzamd_set_max_rows_per_pass.argtypes = [u64]
    lib.bzamd_set_max_rows_per_pass.restype = None
    lib.bzamd_device_id.argtypes = [ctypes.c_int]
    lib.bzamd_device_id.restype = ctypes.c_int
    lib.bzamd_multi_device_columns_per_device.argtypes = [u32]
    lib.bzamd_multi_device_columns_per_device.restype = u32
    lib.bzamd_multi_device_exchange.restype = ctypes.c_char_p
    lib.bzamd_msm_multi_device.argtypes = [cu, ctypes.POINTER(vp), u32,
                                           ctypes.POINTER(sxt_sequence_descriptor),
                                           ctypes.POINTER(vp)]
    lib.bzamd_msm_multi_device.restype = None
    lib.bzamd_set_shard_min_bytes.argtypes = [u64]
    lib.bzamd_set_shard_min_bytes.restype = None
    lib.bzamd_accumulate_form.restype = ctypes.c_int
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def init(backend, num_precomputed_generators=0):
    cfg = sxt_config(backend, num_precomputed_generators)
    return load().sxt_init(ctypes.byref(cfg))


def reset_for_testing():
    load().bzamd_reset_for_testing()


def make_descriptors(columns):
    """columns: iterable of (numpy array of shape [n, nbytes] or 1-D integer array, is_signed).

    Returns (ctypes array, keep-alive list).  1-D integer arrays use their itemsize as
    element_nbytes (little-endian host assumed)."""
    cols = list(columns)
    descs = (sxt_sequence_descriptor * max(1, len(cols)))()
    keep = []
    for i, (arr, is_signed) in enumerate(cols):
        arr = np.ascontiguousarray(arr)
        if arr.ndim == 1:
            nbytes, n = arr.dtype.itemsize, arr.shape[0]
        else:
            assert arr.dtype == np.uint8
            n, nbytes = arr.shape
        keep.append(arr)
        descs[i] = sxt_sequence_descriptor(nbytes, n, arr.ctypes.data if n > 0 else None,
                                           1 if is_signed else 0)
    return descs, keep


def compute_pedersen_commitments(curve_id, columns, generators=None, offset_generators=0):
    """Drop-in Pedersen call with host buffers.  `generators`: uint8 array in the C-ABI layout of
    the curve, or None for the built-in ristretto generators."""
    lib = load()
    descs, keep = make_descriptors(columns)
    num = len(keep)
    out = np.zeros((num, CURVE_LAYOUT[curve_id][1]), dtype=np.uint8)
    if curve_id == SXT_CURVE_RISTRETTO255 and generators is None:
        lib.sxt_curve25519_compute_pedersen_commitments(_ptr(out), num, descs, offset_generators)
        return out
    fn = {
        SXT_CURVE_RISTRETTO255: lib.sxt_curve25519_compute_pedersen_commitments_with_generators,
        SXT_CURVE_BLS_381: lib.sxt_bls12_381_g1_compute_pedersen_commitments_with_generators,
        SXT_CURVE_BN_254:
            lib.sxt_bn254_g1_uncompressed_compute_pedersen_commitments_with_generators,
        SXT_CURVE_GRUMPKIN:
            lib.sxt_grumpkin_uncompressed_compute_pedersen_commitments_with_generators,
    }[curve_id]
    gens = np.ascontiguousarray(generators)
    fn(_ptr(out), num, descs, _ptr(gens))
    return out


def msm_projective(curve_id, columns, generators):
    """raw projective MSM results (host operands), one element per column"""
    lib = load()
    descs, keep = make_descriptors(columns)
    out = np.zeros((len(keep), CURVE_LAYOUT[curve_id][2]), dtype=np.uint8)
    gens = np.ascontiguousarray(generators)
    lib.bzamd_msm_projective(curve_id, _ptr(out), len(keep), descs, _ptr(gens))
    return out


def fold_encode(curve_id, partials):
    """partials: uint8 [num_partials, num_outputs, projective bytes] -> canonical [num_outputs, .]"""
    p = np.ascontiguousarray(partials, dtype=np.uint8)
    num_partials, num_outputs = p.shape[0], p.shape[1]
    out = np.zeros((num_outputs, CURVE_LAYOUT[curve_id][1]), dtype=np.uint8)
    load().bzamd_fold_encode(curve_id, _ptr(out), _ptr(p), num_partials, num_outputs)
    return out


def get_generators(n, offset=0):
    out = np.zeros((n, 20), dtype=np.uint64)
    rc = load().sxt_ristretto255_get_generators(_ptr(out) if n > 0 else None, n, offset)
    assert rc == 0
    return out


def get_one_commit(n):
    out = np.zeros(20, dtype=np.uint64)
    rc = load().sxt_curve25519_get_one_commit(_ptr(out), n)
    assert rc == 0
    return out


def transcript_new(label):
    """a fresh Merlin transcript (203 bytes) with the application label"""
    raw = label.encode() if isinstance(label, str) else bytes(label)
    out = np.zeros(203, dtype=np.uint8)
    load().bzamd_transcript_init(_ptr(out), raw, len(raw))
    return out


def _rounds(n):
    return max(int(n) - 1, 0).bit_length()


def prove_inner_product(transcript, n, generators_offset, a_vector, b_vector):
    """sxt_curve25519_prove_inner_product -> (l [rounds, 32], r [rounds, 32], ap [32], transcript)"""
    t = np.ascontiguousarray(transcript, dtype=np.uint8).copy()
    a = np.ascontiguousarray(a_vector, dtype=np.uint8).reshape(n, 32)
    b = np.ascontiguousarray(b_vector, dtype=np.uint8).reshape(n, 32)
    rounds = _rounds(n)
    l = np.zeros((max(rounds, 1), 32), dtype=np.uint8)
    r = np.zeros((max(rounds, 1), 32), dtype=np.uint8)
    ap = np.zeros(32, dtype=np.uint8)
    load().sxt_curve25519_prove_inner_product(_ptr(l), _ptr(r), _ptr(ap), _ptr(t), n,
                                              generators_offset, _ptr(a), _ptr(b))
    return l[:rounds], r[:rounds], ap, t


def verify_inner_product(transcript, n, generators_offset, b_vector, product, a_commit, l_vector,
                         r_vector, ap_value):
    """sxt_curve25519_verify_inner_product -> (bool, transcript after)"""
    t = np.ascontiguousarray(transcript, dtype=np.uint8).copy()
    b = np.ascontiguousarray(b_vector, dtype=np.uint8).reshape(n, 32)
    lv = np.ascontiguousarray(l_vector, dtype=np.uint8).reshape(-1, 32)
    rv = np.ascontiguousarray(r_vector, dtype=np.uint8).reshape(-1, 32)
    if lv.shape[0] == 0:
        lv = rv = np.zeros((1, 32), np.uint8)
    rc = load().sxt_curve25519_verify_inner_product(
        _ptr(t), n, generators_offset, _ptr(b), _ptr(np.ascontiguousarray(product, np.uint8)),
        _ptr(np.ascontiguousarray(a_commit, np.uint64)), _ptr(lv), _ptr(rv),
        _ptr(np.ascontiguousarray(ap_value, np.uint8)))
    return bool(rc), t


class sumcheck_descriptor(ctypes.Structure):
    _fields_ = [("mles", ctypes.c_void_p), ("product_table", ctypes.c_void_p),
                ("product_terms", ctypes.c_void_p), ("n", ctypes.c_uint),
                ("num_mles", ctypes.c_uint), ("num_products", ctypes.c_uint),
                ("num_product_terms", ctypes.c_uint), ("round_degree", ctypes.c_uint)]


SUMCHECK_CALLBACK = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_uint)
SUMCHECK_PRODUCT_STRIDE = {0: 36, 1: 40}  # std::pair<FIELD, unsigned> of the reference


def prove_sumcheck(field_id, mles, product_table, product_terms, n, round_degree, callback):
    """sxt_prove_sumcheck.  mles: uint8 [num_mles, n, 32]; product_table: raw bytes of
    num_products x {32-byte multiplier; unsigned length}; callback(r_ptr, ctx, poly_ptr, length).
    -> (polynomials [num_variables, round_degree + 1, 32], evaluation_point [num_variables, 32])"""
    m = np.ascontiguousarray(mles, dtype=np.uint8)
    table = np.ascontiguousarray(product_table, dtype=np.uint8)
    terms = np.ascontiguousarray(product_terms, dtype=np.uint32)
    num_variables = max((int(n) - 1).bit_length(), 1)
    polys = np.zeros((num_variables, round_degree + 1, 32), dtype=np.uint8)
    point = np.zeros((num_variables, 32), dtype=np.uint8)
    d = sumcheck_descriptor(m.ctypes.data, table.ctypes.data, terms.ctypes.data, n, m.shape[0],
                            table.size // SUMCHECK_PRODUCT_STRIDE[field_id], terms.size,
                            round_degree)
    cb = SUMCHECK_CALLBACK(callback)
    fn = load().sxt_prove_sumcheck
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint,
                   ctypes.POINTER(sumcheck_descriptor), SUMCHECK_CALLBACK, ctypes.c_void_p]
    fn.restype = None
    fn(_ptr(polys), _ptr(point), field_id, ctypes.byref(d), cb, None)
    return polys, point


class MultiexpHandle:
    """sxt_multiexp_handle wrapper (fixed generators)."""

    def __init__(self, curve_id, generators_projective=None, filename=None):
        lib = load()
        self.curve_id = curve_id
        if filename is not None:
            self._h = lib.sxt_multiexp_handle_new_from_file(curve_id, filename.encode())
        else:
            g = np.ascontiguousarray(generators_projective)
            psize = CURVE_LAYOUT[curve_id][2]
            assert g.nbytes % psize == 0
            self._h = lib.sxt_multiexp_handle_new(curve_id, _ptr(g), g.nbytes // psize)

    def write_to_file(self, filename):
        load().sxt_multiexp_handle_write_to_file(self._h, filename.encode())

    def _out(self, num_outputs):
        return np.zeros((num_outputs, CURVE_LAYOUT[self.curve_id][2]), dtype=np.uint8)

    def multiexponentiation(self, element_num_bytes, num_outputs, n, scalars):
        s = np.ascontiguousarray(scalars, dtype=np.uint8)
        out = self._out(num_outputs)
        load().sxt_fixed_multiexponentiation(_ptr(out), self._h, element_num_bytes, num_outputs, n,
                                             _ptr(s))
        return out

    def packed_multiexponentiation(self, bit_table, n, scalars):
        bt = np.ascontiguousarray(bit_table, dtype=np.uint32)
        s = np.ascontiguousarray(scalars, dtype=np.uint8)
        out = self._out(len(bt))
        load().sxt_fixed_packed_multiexponentiation(_ptr(out), self._h, _ptr(bt), len(bt), n,
                                                    _ptr(s))
        return out

    def vlen_multiexponentiation(self, bit_table, lengths, scalars):
        bt = np.ascontiguousarray(bit_table, dtype=np.uint32)
        ln = np.ascontiguousarray(lengths, dtype=np.uint32)
        s = np.ascontiguousarray(scalars, dtype=np.uint8)
        out = self._out(len(bt))
        load().sxt_fixed_vlen_multiexponentiation(_ptr(out), self._h, _ptr(bt), _ptr(ln), len(bt),
                                                  _ptr(s))
        return out

    def close(self):
        if self._h:
            load().sxt_multiexp_handle_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

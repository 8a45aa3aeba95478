"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's fixed-base MSM ("pippenger2").

Round 1 could not compile the reference's fixed-base path (partition_product.h, reduce.h,
combine_reduce.h and multiexponentiation.h include sxt/algorithm/iteration/for_each.h -- `__global__`,
`<<<>>>` -- and the accessor pulls the CUDA runtime) and restated its ~60 lines of control flow here
over the parts that did compile: the partition-table builder (sxt/multiexp/pippenger2/
partition_table.h:36-98) and every curve operation, from oracle/_ref.  Since round 4 the reference's
OWN path is compiled in place as well (oracle/ref/ref_fixed_base.cc over host stand-ins for the CUDA
runtime; ref_oracle.FixedHandle: host loop and GPU control flow), and this restatement is kept as a
second opinion that tests/test_oracle.py pins against it.  Each function cites the reference lines it
follows:

  pad_generators        in_memory_partition_table_accessor_utility.h:41-58
  partition_index       partition_product.h:47-67   (compute_partition_index)
  plane_products        partition_product.h:69-94,169-194  (step = ceil(P / 8) bytes per row)
  reduce_output         reduce.h:37-47              (MSB -> LSB: double, add)
  multiexponentiate     multiexponentiation.h:266-288 (CPU path) / cpu_backend.cc fixed overloads
  product_lengths       variable_length_computation.cc:28-62
  vlen multiexponentiate  variable_length_multiexponentiation.h:197-232 (GPU-lambda semantics
                        :101-109; the host loop's product_index mutation at :145-148 is a latent
                        bug that only matters for zero-length outputs)

Pinning: tests/test_oracle.py checks this restatement against the reference's own compiled
fixed-base path (host loop and GPU control flow), against its variable-base CPU backend (oracle/_ref)
on the same scalars unpacked to columns, and against the known answers of
cbindings/fixed_pedersen.t.cc:51-200 expressed as group expressions.

Only tests/ may import this module; the product never does.
"""
import numpy as np

from oracle import ref_oracle

# compact (table entry) sizes in bytes and projective element sizes in u64 words
_COMPACT_BYTES = {0: 120, 1: 96, 2: 64, 3: 64}
_PROJ_WORDS = {0: 20, 1: 18, 2: 12, 3: 12}


def identity_projective(curve_id):
    """T::identity(): curve25519 {0,1,1,0}; Weierstrass {0, R, 0}"""
    if curve_id == 0:
        p = np.zeros(20, dtype=np.uint64)
        p[5] = 1
        p[10] = 1
        return p
    nl = ref_oracle.CURVES[curve_id][1]
    aff = ref_oracle.identity_affine(curve_id)  # {0, R, inf}
    p = np.zeros(3 * nl, dtype=np.uint64)
    p[nl:2 * nl] = aff[8 * nl:16 * nl].view(np.uint64)
    return p


def pad_generators(curve_id, generators, w):
    """identity padding to a multiple of the window width"""
    g = np.ascontiguousarray(generators, dtype=np.uint64).reshape(-1, _PROJ_WORDS[curve_id])
    n = g.shape[0]
    n_pad = (n + w - 1) // w * w
    if n_pad == n:
        return g
    pad = np.tile(identity_projective(curve_id), (n_pad - n, 1))
    return np.concatenate([g, pad])


def expand_compact(curve_id, entry_bytes):
    """compact table entry U -> projective T (the conversion operator of each compact_element)"""
    if curve_id == 0:
        c = entry_bytes.view(np.uint64)  # X, Y, T
        p = np.zeros(20, dtype=np.uint64)
        p[0:5] = c[0:5]
        p[5:10] = c[5:10]
        p[10] = 1
        p[15:20] = c[10:15]
        return p
    nl = ref_oracle.CURVES[curve_id][1]
    c = entry_bytes.view(np.uint64)
    if c[nl - 1] == np.uint64(0xFFFFFFFFFFFFFFFF):  # identity sentinel
        return identity_projective(curve_id)
    p = np.zeros(3 * nl, dtype=np.uint64)
    p[0:nl] = c[0:nl]
    p[nl:2 * nl] = c[nl:2 * nl]
    p[2 * nl:3 * nl] = identity_projective(curve_id)[nl:2 * nl]  # Z = R (one)
    return p


class PartitionTable:
    """in_memory_partition_table_accessor: the table of the reference, built by the reference's
    own compute_partition_table (oracle/_ref)"""

    def __init__(self, curve_id, generators_projective, window_width):
        self.curve_id = curve_id
        self.w = window_width
        g = pad_generators(curve_id, generators_projective, window_width)
        self.n = g.shape[0]
        raw = ref_oracle.partition_table(curve_id, window_width, g)
        self.entries = raw.reshape(-1, _COMPACT_BYTES[curve_id])

    def lookup(self, window, mask):
        return expand_compact(self.curve_id, self.entries[(window << self.w) + mask])

    def file_bytes(self):
        """on-disk format: u32 w, then the raw U array (in_memory_partition_table_accessor.h:
        42-59,98-105)"""
        return np.uint32(self.w).tobytes() + self.entries.tobytes()


def _add(curve_id, a, b):
    return ref_oracle.add_projective(curve_id, a, b)


def _double(curve_id, a):
    return ref_oracle.double_projective(curve_id, a)


def partition_index(scalars, row_stride, byte_index, bit_in_byte, first_row, w, n):
    """bit i of the result = plane bit of row first_row + i (rows >= n contribute 0)"""
    m = 0
    for i in range(w):
        row = first_row + i
        if row >= n:
            break
        byte = int(scalars[row * row_stride + byte_index])
        m |= ((byte >> bit_in_byte) & 1) << i
    return m


def plane_products(table, scalars, num_planes, lengths_per_plane):
    """product[p] = sum over windows of table[window][partition_index]"""
    cid = table.curve_id
    row_stride = (num_planes + 7) // 8
    out = []
    for p in range(num_planes):
        n = lengths_per_plane[p]
        acc = identity_projective(cid)
        num_windows = (n + table.w - 1) // table.w
        for k in range(num_windows):
            m = partition_index(scalars, row_stride, p // 8, p % 8, k * table.w, table.w, n)
            if m:
                acc = _add(cid, acc, table.lookup(k, m))
        out.append(acc)
    return out


def reduce_output(curve_id, planes):
    """res = q_0 + 2 (q_1 + 2 (q_2 + ...)), evaluated from the top plane down"""
    if not planes:
        return identity_projective(curve_id)
    acc = planes[-1]
    for q in reversed(planes[:-1]):
        acc = _double(curve_id, acc)
        acc = _add(curve_id, acc, q)
    return acc


def product_lengths(bit_table, output_lengths):
    """per-plane effective length (variable_length_computation.cc:28-62)"""
    out = []
    for bits, length in zip(bit_table, output_lengths):
        out += [int(length)] * int(bits)
    return out


def multiexponentiate(table, bit_table, n, scalars, output_lengths=None):
    """packed / vlen fixed-base MSM; returns one projective element per output.
    `scalars`: uint8, n rows of ceil(sum(bit_table) / 8) bytes."""
    bit_table = [int(b) for b in bit_table]
    assert all(b > 0 for b in bit_table), "bit widths must be positive"
    if output_lengths is None:
        output_lengths = [n] * len(bit_table)
    else:
        output_lengths = [int(x) for x in output_lengths]
        assert all(a <= b for a, b in zip(output_lengths, output_lengths[1:])), \
            "output lengths must be ascending"
    num_planes = sum(bit_table)
    s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1)
    planes = plane_products(table, s, num_planes, product_lengths(bit_table, output_lengths))
    res, first = [], 0
    for bits, length in zip(bit_table, output_lengths):
        if length == 0:
            res.append(identity_projective(table.curve_id))
        else:
            res.append(reduce_output(table.curve_id, planes[first:first + bits]))
        first += bits
    return np.stack(res)


def multiexponentiate_bytes(table, element_num_bytes, num_outputs, n, scalars):
    """sxt_fixed_multiexponentiation: every output is 8 * element_num_bytes bits wide"""
    return multiexponentiate(table, [8 * element_num_bytes] * num_outputs, n, scalars)


def unpack_columns(bit_table, n, scalars, output_lengths=None):
    """the same packed scalars as independent byte columns for the variable-base oracle (the
    cross-check SURVEY 8(c) prescribes); each column is ceil(bits / 8) bytes wide"""
    bit_table = [int(b) for b in bit_table]
    row_stride = (sum(bit_table) + 7) // 8
    s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, row_stride) if n else \
        np.zeros((0, row_stride), np.uint8)
    bits = np.unpackbits(s, axis=1, bitorder="little")
    cols, first = [], 0
    for k, b in enumerate(bit_table):
        length = n if output_lengths is None else int(output_lengths[k])
        field = bits[:length, first:first + b]
        nbytes = (b + 7) // 8
        padded = np.zeros((length, 8 * nbytes), dtype=np.uint8)
        padded[:, :b] = field
        cols.append((np.packbits(padded, axis=1, bitorder="little"), False))
        first += b
    return cols

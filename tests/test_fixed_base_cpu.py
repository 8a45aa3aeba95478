"""Fixed-base handles (sxt_multiexp_handle_*, sxt_fixed_*multiexponentiation) on the host
backend, through the C ABI, against the fixed-base oracle (oracle/fixed_base.py over oracle/_ref)
and the golden fixtures.  Results are projective and not canonical in the reference either
(cbindings/fixed_pedersen.t.cc:63,133-134 compares with projective ==), so both sides are
canonicalised before the byte comparison."""
import os

import numpy as np
import pytest

from tests import util

GOLDEN = np.load(os.path.join(os.path.dirname(__file__), "golden", "msm_golden.npz"))


def canon(oracle, cid, res):
    words = res.view(np.uint64).reshape(res.shape[0], -1)
    return np.stack([oracle.canonical(cid, p).view(np.uint8).reshape(-1) for p in words])


def projective_generators(oracle, cid, n):
    gens = util.generators_for(cid, n)
    return gens if cid == 0 else oracle.affine_to_projective(cid, gens)


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_golden_packed(cpu_backend, oracle, cid):
    proj = GOLDEN[f"curve{cid}_fixed_projective_generators"]
    h = cpu_backend.MultiexpHandle(cid, proj)
    res = h.packed_multiexponentiation(GOLDEN["fixed_bit_table"], proj.shape[0],
                                       GOLDEN[f"curve{cid}_fixed_scalars"])
    assert np.array_equal(canon(oracle, cid, res), GOLDEN[f"curve{cid}_fixed_canonical"])
    h.close()


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_plain_packed_vlen_match_oracle(cpu_backend, oracle, cid):
    from oracle import fixed_base
    rng = np.random.default_rng(300 + cid)
    n = 37
    proj = projective_generators(oracle, cid, n)
    table = fixed_base.PartitionTable(cid, proj, 4)
    h = cpu_backend.MultiexpHandle(cid, proj)
    # plain: 3 outputs of 2 bytes
    s = rng.integers(0, 256, (n, 6), dtype=np.uint8)
    got = h.multiexponentiation(2, 3, n, s)
    want = fixed_base.multiexponentiate_bytes(table, 2, 3, n, s)
    assert np.array_equal(canon(oracle, cid, got), canon(oracle, cid, want))
    # packed: odd widths, a 256-bit output, fewer rows than generators
    bt = [1, 7, 9, 33, 256, 2]
    m = 29
    s = rng.integers(0, 256, (m, (sum(bt) + 7) // 8), dtype=np.uint8)
    got = h.packed_multiexponentiation(bt, m, s)
    want = fixed_base.multiexponentiate(table, bt, m, s)
    assert np.array_equal(canon(oracle, cid, got), canon(oracle, cid, want))
    # vlen: ascending lengths with a zero-length output
    bt, lengths = [4, 12, 1, 64], [0, 5, 5, 37]
    s = rng.integers(0, 256, (n, (sum(bt) + 7) // 8), dtype=np.uint8)
    got = h.vlen_multiexponentiation(bt, lengths, s)
    want = fixed_base.multiexponentiate(table, bt, n, s, lengths)
    assert np.array_equal(canon(oracle, cid, got), canon(oracle, cid, want))
    h.close()


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_outputs_wider_than_256_bits(cpu_backend, oracle, cid):
    """the reference takes any unsigned bit width per output (cbindings/blitzar_api.h:712,
    pippenger2/multiexponentiation.h:207-288: one bit plane per bit); here such an output is
    computed as 256-bit pieces and folded -- against the bit-plane oracle"""
    _wide_outputs(cpu_backend, oracle, cid, 23)


def _wide_outputs(api, oracle, cid, n):
    from oracle import fixed_base
    rng = np.random.default_rng(5300 + cid)
    proj = projective_generators(oracle, cid, n)
    table = fixed_base.PartitionTable(cid, proj, 4)
    h = api.MultiexpHandle(cid, proj)
    bt = [300, 8, 513, 256, 1]
    s = rng.integers(0, 256, (n, (sum(bt) + 7) // 8), dtype=np.uint8)
    got = h.packed_multiexponentiation(bt, n, s)
    want = fixed_base.multiexponentiate(table, bt, n, s)
    assert np.array_equal(canon(oracle, cid, got), canon(oracle, cid, want))
    lengths = [0, 3, 3, n - 1, n]
    got = h.vlen_multiexponentiation(bt, lengths, s)
    want = fixed_base.multiexponentiate(table, bt, n, s, lengths)
    assert np.array_equal(canon(oracle, cid, got), canon(oracle, cid, want))
    # plain entry point with 40-byte elements
    s = rng.integers(0, 256, (n, 80), dtype=np.uint8)
    got = h.multiexponentiation(40, 2, n, s)
    want = fixed_base.multiexponentiate_bytes(table, 40, 2, n, s)
    assert np.array_equal(canon(oracle, cid, got), canon(oracle, cid, want))
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cid", [0, 1, 3])
def test_outputs_wider_than_256_bits_gpu(gpu_backend, oracle, cid):
    _wide_outputs(gpu_backend, oracle, cid, 23)


def test_known_answers_of_the_reference_tests(cpu_backend, oracle):
    """cbindings/fixed_pedersen.t.cc:51-200"""
    g = oracle.ristretto_generators(3, 7)

    def lin(coeffs):
        return oracle.commit(0, [(np.array(coeffs, dtype=np.uint64), False)], g)[0]

    h = cpu_backend.MultiexpHandle(0, g[:2])
    r = h.multiexponentiation(2, 1, 2, np.array([1, 0, 0, 2], np.uint8))
    assert np.array_equal(canon(oracle, 0, r)[0], lin([1, 512]))
    r = h.packed_multiexponentiation([3, 1], 2, np.array([0b1010, 0b0101], np.uint8))
    assert np.array_equal(canon(oracle, 0, r), np.stack([lin([2, 5]), lin([1, 0])]))
    r = h.vlen_multiexponentiation([3, 1], [1, 2], np.array([0b1011, 0b1101], np.uint8))
    assert np.array_equal(canon(oracle, 0, r), np.stack([lin([3, 0]), lin([1, 1])]))
    h.close()
    h = cpu_backend.MultiexpHandle(0, g)
    r = h.packed_multiexponentiation([8], 3, np.array([1, 1, 1], np.uint8))
    assert np.array_equal(canon(oracle, 0, r)[0], lin([1, 1, 1]))
    h.close()


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_partition_table_file_interop(cpu_backend, oracle, cid, tmp_path, monkeypatch):
    """sxt_multiexp_handle_write_to_file writes the reference's table format byte for byte
    (in_memory_partition_table_accessor.h:42-59,98-105) and _new_from_file reads it back"""
    from oracle import fixed_base
    monkeypatch.setenv("BLITZAR_PARTITION_WINDOW_WIDTH", "4")
    proj = GOLDEN[f"curve{cid}_fixed_projective_generators"]  # 11 generators -> padded to 12
    h = cpu_backend.MultiexpHandle(cid, proj)
    path = str(tmp_path / f"table{cid}.bin")
    h.write_to_file(path)
    h.close()
    with open(path, "rb") as fh:
        data = fh.read()
    want = fixed_base.PartitionTable(cid, proj, 4).file_bytes()
    assert data == want
    h2 = cpu_backend.MultiexpHandle(cid, filename=path)
    n = proj.shape[0]
    res = h2.packed_multiexponentiation(GOLDEN["fixed_bit_table"], n,
                                        GOLDEN[f"curve{cid}_fixed_scalars"])
    assert np.array_equal(canon(oracle, cid, res), GOLDEN[f"curve{cid}_fixed_canonical"])
    h2.close()


def dump_record_and_replay(api, backend_id, oracle, cid, tmp_path, monkeypatch):
    """BLITZAR_DUMP_DIR writes the reference's recording layout
    (multiexponentiation_serialization.h:70-103, gpu_backend.cc:286-301,317-332); the recording
    replays to the same result, and its generators.bin is the reference's compact form"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "tools"))
    import replay_dump
    from oracle import fixed_base
    proj = GOLDEN[f"curve{cid}_fixed_projective_generators"]
    monkeypatch.setenv("BLITZAR_DUMP_DIR", str(tmp_path))
    h = api.MultiexpHandle(cid, proj)
    n = proj.shape[0]
    h.packed_multiexponentiation(GOLDEN["fixed_bit_table"], n, GOLDEN[f"curve{cid}_fixed_scalars"])
    bt, lengths = [4, 12, 1], [0, 5, n]
    s = np.random.default_rng(1).integers(0, 256, (n, 3), dtype=np.uint8)
    h.vlen_multiexponentiation(bt, lengths, s)
    h.close()
    monkeypatch.delenv("BLITZAR_DUMP_DIR")
    dirs = sorted(os.listdir(tmp_path))
    assert [d.rsplit("-", 1)[0] for d in dirs] == ["packed-multiexponentiation",
                                                   "vlen-multiexponentiation"]
    packed = tmp_path / dirs[0]
    assert sorted(os.listdir(packed)) == ["generators.bin", "meta.txt", "output_bit_table.bin",
                                          "result.bin", "scalars.bin", "window_width.bin"]
    assert "output_lengths.bin" in os.listdir(tmp_path / dirs[1])
    # generators.bin == entries 2^i of the reference's partition table (copy_generators)
    table = fixed_base.PartitionTable(cid, proj, 4)
    want = np.stack([table.entries[((i // 4) << 4) + (1 << (i % 4))] for i in range(n)])
    got = np.fromfile(packed / "generators.bin", np.uint8).reshape(n, -1)
    assert np.array_equal(got, want)
    assert np.fromfile(packed / "window_width.bin", np.uint64).tolist() == [16]
    for d in dirs:
        assert replay_dump.replay(str(tmp_path / d), backend_id)[2]
    # the recorded result itself is the right group element (not only self-consistent)
    res = np.fromfile(packed / "result.bin", np.uint8).reshape(len(GOLDEN["fixed_bit_table"]), -1)
    assert np.array_equal(canon(oracle, cid, res), GOLDEN[f"curve{cid}_fixed_canonical"])


def dump_interop_with_the_reference(api, backend_id, oracle, cid, tmp_path, monkeypatch):
    """f2 against the reference's OWN writer and reader (multiexponentiation_serialization.h:70-151,
    compiled in place into the oracle library), both directions, packed and vlen:
      1. a recording the reference wrote (its write_multiexponentiation over its own accessor, result.bin
         = the raw elements its multiexponentiate returned, as gpu_backend.cc:298-300 writes them)
         replays through this library to the same canonical bytes;
      2. a recording THIS library wrote is read by the reference's read_multiexponentiation -- which
         rebuilds its accessor from our generators.bin + window_width.bin -- to the descriptor we were
         called with, and the reference's multiexponentiate over it gives our result.bin (canonical);
      3. the two recordings of one call are byte-identical file by file, meta.txt (the typeid names
         the reference writes) included."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "tools"))
    import replay_dump
    proj = GOLDEN[f"curve{cid}_fixed_projective_generators"]
    n = proj.shape[0]
    width = 4
    bt = np.asarray(GOLDEN["fixed_bit_table"], np.uint32)
    sc = np.asarray(GOLDEN[f"curve{cid}_fixed_scalars"], np.uint8).reshape(-1)
    vbt, vln = np.array([4, 12, 1], np.uint32), np.array([0, 5, n], np.uint32)
    vsc = np.random.default_rng(2).integers(0, 256, (n, 3), dtype=np.uint8).reshape(-1)
    ref = oracle.FixedHandle(cid, proj, width)
    # 1. written by the reference
    theirs = {}
    for kind, args in (("packed", dict(bit_table=bt, scalars=sc, n=n)),
                       ("vlen", dict(bit_table=vbt, scalars=vsc, lengths=vln))):
        d = tmp_path / f"ref-{kind}"
        d.mkdir()
        ref.write_dump(str(d), **args)
        res = (ref.packed_multiexponentiation(bt, n, sc) if kind == "packed"
               else ref.vlen_multiexponentiation(vbt, vln, vsc, gpu_flow=True))
        res.tofile(d / "result.bin")
        theirs[kind] = d
        assert replay_dump.replay(str(d), backend_id) == (cid, len(args["bit_table"]), True)
    # 2. written by this library
    ours_root = tmp_path / "ours"
    ours_root.mkdir()
    monkeypatch.setenv("BLITZAR_DUMP_DIR", str(ours_root))
    monkeypatch.setenv("BLITZAR_PARTITION_WINDOW_WIDTH", str(width))
    h = api.MultiexpHandle(cid, proj)
    h.packed_multiexponentiation(bt, n, sc)
    h.vlen_multiexponentiation(vbt, vln, vsc)
    h.close()
    monkeypatch.delenv("BLITZAR_DUMP_DIR")
    ours = {d.rsplit("-", 2)[0]: ours_root / d for d in sorted(os.listdir(ours_root))}
    for kind in ("packed", "vlen"):
        mine = ours[kind]
        back, r_bt, r_ln, r_sc = oracle.FixedHandle.read_dump(cid, str(mine), vlen=kind == "vlen")
        if kind == "packed":
            assert np.array_equal(r_bt, bt) and r_ln is None and np.array_equal(r_sc, sc)
            again = back.packed_multiexponentiation(r_bt, n, r_sc)
        else:
            assert np.array_equal(r_bt, vbt) and np.array_equal(r_ln, vln) and np.array_equal(r_sc, vsc)
            again = back.vlen_multiexponentiation(r_bt, r_ln, r_sc, gpu_flow=True)
        back.close()
        recorded = np.fromfile(mine / "result.bin", np.uint8).reshape(len(r_bt), -1)
        assert np.array_equal(canon(oracle, cid, again), canon(oracle, cid, recorded))
        # 3. file by file
        for name in sorted(os.listdir(theirs[kind])):
            if name == "result.bin":
                continue  # projective, not canonical: compared above
            assert (mine / name).read_bytes() == (theirs[kind] / name).read_bytes(), (kind, name)
        assert sorted(os.listdir(mine)) == sorted(os.listdir(theirs[kind]))
    ref.close()


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_dump_dir_interop_with_the_reference(cpu_backend, oracle, cid, tmp_path, monkeypatch):
    dump_interop_with_the_reference(cpu_backend, cpu_backend.SXT_CPU_BACKEND, oracle, cid, tmp_path,
                                    monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_dump_dir_interop_with_the_reference_gpu(gpu_backend, oracle, cid, tmp_path, monkeypatch):
    before = gpu_backend.load().bzamd_kernel_launch_count()
    dump_interop_with_the_reference(gpu_backend, gpu_backend.SXT_GPU_BACKEND, oracle, cid, tmp_path,
                                    monkeypatch)
    assert gpu_backend.load().bzamd_kernel_launch_count() > before


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_dump_dir_record_and_replay(cpu_backend, oracle, cid, tmp_path, monkeypatch):
    dump_record_and_replay(cpu_backend, cpu_backend.SXT_CPU_BACKEND, oracle, cid, tmp_path,
                           monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_dump_dir_record_and_replay_gpu(gpu_backend, oracle, cid, tmp_path, monkeypatch):
    """the reference records dumps on its GPU backend only (gpu_backend.cc:286-301,317-332):
    record through the HIP engine, replay through the HIP engine"""
    before = gpu_backend.load().bzamd_kernel_launch_count()
    dump_record_and_replay(gpu_backend, gpu_backend.SXT_GPU_BACKEND, oracle, cid, tmp_path,
                           monkeypatch)
    assert gpu_backend.load().bzamd_kernel_launch_count() > before

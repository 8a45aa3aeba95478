"""Multi-device execution INSIDE the C ABI (one process drives every device, like the reference's
gpu backend: sxt/execution/device/for_each.cc:56-82, gpu_backend.cc:150-193).

There is one GPU per box here, so the sharded code paths are forced the way the reference's own
tests force chunking with split_options (pippenger2/multiexponentiation.t.cc:150-180):
BLITZAR_AMD_FORCE_SHARDS=k makes k logical devices -- host threads on the cpu backend (runs
anywhere), k stream/context/arena sets on the one physical GPU under -m gpu.  Every sharded result
must equal the reference oracle's, whatever the split: columns over devices, rows of a single
long column over devices (projective partials + fold), outputs of a fixed-base call over devices.
Each case runs in a child process: the shard count is read once at sxt_init.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, {root!r})
backend = {backend}
if backend == 2:
    import torch  # the HIP runtime torch ships must be the one the library binds to
from blitzar_amd import api
from tests import util
lib = api.load()
assert api.init(backend, 64) == 0
assert lib.bzamd_num_devices() == {shards}
lib.bzamd_set_shard_min_bytes(0)
rng = np.random.default_rng({seed})
out = {{}}
launches = lib.bzamd_kernel_launch_count()
for cid in {curves}:
    n = {n}
    gens = util.generators_for(cid, n)
    g = util.api_generators(cid, gens)
    many = util.mixed_columns(rng, n)[:11]                       # >= shards columns: column split
    few = [(rng.integers(0, 256, (n, 32), dtype=np.uint8), False),
           (rng.integers(0, 256, (n - 3, 8), dtype=np.uint8), True)]  # < shards: row split
    out[f"many{{cid}}"] = api.compute_pedersen_commitments(cid, many, generators=g).tolist()
    out[f"few{{cid}}"] = api.compute_pedersen_commitments(cid, few, generators=g).tolist()
    if cid == 0:
        out["builtin_few"] = api.compute_pedersen_commitments(0, few, offset_generators=7).tolist()
        out["builtin_many"] = api.compute_pedersen_commitments(0, many, offset_generators=50).tolist()
    proj = gens if cid == 0 else util.ref_oracle.affine_to_projective(cid, gens)
    h = api.MultiexpHandle(cid, proj)
    bt = [8, 32, 256, 5, 1, 64, 13]
    s = rng.integers(0, 256, (n, (sum(bt) + 7) // 8), dtype=np.uint8)
    out[f"packed{{cid}}"] = h.packed_multiexponentiation(bt, n, s).tolist()
    lengths = [0, 5, 5, 9, n // 2, n - 1, n]
    out[f"vlen{{cid}}"] = h.vlen_multiexponentiation(bt, lengths, s).tolist()
    h.close()
out["launches"] = int(lib.bzamd_kernel_launch_count() - launches)
print("RESULT" + json.dumps(out))
"""


def run_child(backend, shards, curves, n, seed):
    env = dict(os.environ, BLITZAR_AMD_FORCE_SHARDS=str(shards))
    env.pop("BLITZAR_BACKEND", None)
    code = CHILD.format(root=ROOT, backend=backend, shards=shards, curves=curves, n=n, seed=seed)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = next(ln for ln in r.stdout.splitlines() if ln.startswith("RESULT"))
    return json.loads(line[len("RESULT"):])


def expected(oracle, curves, n, seed):
    from oracle import fixed_base
    rng = np.random.default_rng(seed)
    want = {}
    for cid in curves:
        gens = util.generators_for(cid, n)
        many = util.mixed_columns(rng, n)[:11]
        few = [(rng.integers(0, 256, (n, 32), dtype=np.uint8), False),
               (rng.integers(0, 256, (n - 3, 8), dtype=np.uint8), True)]
        want[f"many{cid}"] = oracle.commit(cid, many, gens)
        want[f"few{cid}"] = oracle.commit(cid, few, gens)
        if cid == 0:
            want["builtin_few"] = oracle.commit(0, few, oracle.ristretto_generators(n, 7))
            want["builtin_many"] = oracle.commit(0, many, oracle.ristretto_generators(n, 50))
        bt = [8, 32, 256, 5, 1, 64, 13]
        s = rng.integers(0, 256, (n, (sum(bt) + 7) // 8), dtype=np.uint8)
        want[f"packed{cid}"] = oracle.commit(cid, fixed_base.unpack_columns(bt, n, s), gens)
        lengths = [0, 5, 5, 9, n // 2, n - 1, n]
        want[f"vlen{cid}"] = oracle.commit(cid, fixed_base.unpack_columns(bt, n, s, lengths), gens)
    return want


def check(oracle, got, want, curves):
    for key, w in want.items():
        g = np.array(got[key], dtype=np.uint8)
        if key.startswith(("packed", "vlen")):
            cid = int(key[-1])
            words = g.view(np.uint64).reshape(g.shape[0], -1)
            g = np.stack([np.ascontiguousarray(oracle.canonical(cid, p)).view(np.uint8).reshape(-1)
                          for p in words])[:, :w.shape[1]]
        assert np.array_equal(g, w), f"{key} differs"


@pytest.mark.parametrize("shards", [2, 3, 8])
def test_forced_shards_host_backend(oracle, shards):
    curves, n, seed = [0, 2], 4000, 600 + shards
    got = run_child(1, shards, curves, n, seed)
    check(oracle, got, expected(oracle, curves, n, seed), curves)


@pytest.mark.gpu
@pytest.mark.parametrize("shards,curves", [(2, [0, 1]), (3, [2]), (8, [0, 3])])
def test_forced_shards_gpu_backend(oracle, shards, curves):
    """k logical devices on the one physical GPU: per-device threads, streams, contexts and arenas,
    direct D2H of column shards, hipMemcpyPeerAsync of row-split partials into device 0 and the
    fold-and-encode kernel, strided uploads of fixed-base output ranges"""
    n, seed = 6000, 700 + shards
    got = run_child(2, shards, curves, n, seed)
    assert got["launches"] > 10 * shards
    check(oracle, got, expected(oracle, curves, n, seed), curves)


def test_split_by_weight_is_a_partition():
    """the split helper through its observable effect: any shard count gives the single-shard
    answer (ragged: more shards than columns, empty and zero-length columns)"""
    from blitzar_amd import api
    rng = np.random.default_rng(9)
    n = 1500
    cols = [(rng.integers(0, 256, (n, 4), dtype=np.uint8), False),
            (np.zeros((0, 8), np.uint8), False),
            (rng.integers(0, 256, (n // 3, 32), dtype=np.uint8), False)]
    code = (
        "import sys, json, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from blitzar_amd import api\n"
        "assert api.init(api.SXT_CPU_BACKEND, 0) == 0\n"
        "api.load().bzamd_set_shard_min_bytes(0)\n"
        "rng = np.random.default_rng(9)\n"
        f"n = {n}\n"
        "cols = [(rng.integers(0, 256, (n, 4), dtype=np.uint8), False),\n"
        "        (np.zeros((0, 8), np.uint8), False),\n"
        "        (rng.integers(0, 256, (n // 3, 32), dtype=np.uint8), False)]\n"
        "print('RESULT' + json.dumps(api.compute_pedersen_commitments(0, cols).tolist()))\n")
    outs = []
    for shards in (1, 2, 3, 5):
        env = dict(os.environ, BLITZAR_AMD_FORCE_SHARDS=str(shards))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(next(ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")))
    assert len(set(outs)) == 1
    _ = (api, cols)


MD_CHECK = os.path.join(ROOT, "tools", "pipeline_bench", "_build", "multi_device_check")


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [3, 8])
def test_msm_multi_device_native_check(shards):
    """bzamd_msm_multi_device (columns sharded over the devices one process drives, all-gather of the
    commitments) and the sharded blocking entry points against the same work confined to device 0,
    through the native self-check bench.py also runs on a multi-GPU node.  Logical devices on the
    one visible GPU exchange with peer copies (RCCL refuses a communicator with duplicate devices);
    the RCCL exchange itself -- ncclCommInitAll + ncclAllGather with a single rank -- runs in
    test_msm_multi_device_matches_oracle below, on the librccl the test process already maps (a
    native process would page the ROCm install's 570 MB copy in first: up to a minute on a fresh
    box)."""
    if not os.path.exists(MD_CHECK):  # normally built by __graft_entry__.build()
        src_dir = os.path.dirname(os.path.dirname(MD_CHECK))
        lib_dir = os.path.join(ROOT, "blitzar_amd", "lib")
        os.makedirs(os.path.dirname(MD_CHECK), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                        "-I" + src_dir, os.path.join(src_dir, "multi_device_check.cc"),
                        "-L" + lib_dir, "-lblitzar_amd", "-Wl,-rpath," + lib_dir, "-o", MD_CHECK],
                       check=True)
    env = dict(os.environ)
    env.pop("BLITZAR_AMD_NUM_DEVICES", None)
    env["BLITZAR_AMD_FORCE_SHARDS"] = str(shards)
    r = subprocess.run([MD_CHECK, "--log2n", "14", "--columns", "11", "--steps", "2"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["ok"] and line["devices"] == shards
    assert line["exchange"] == "peer-copies", line


@pytest.mark.gpu
def test_msm_multi_device_matches_oracle(gpu_backend, oracle, monkeypatch):
    """the multi-device entry point itself against the reference (one device here: the column
    ranges, the padded send buffer and the RCCL exchange with a single rank -- a backend of ONE
    device skips RCCL altogether unless BLITZAR_AMD_RCCL_SINGLE_DEVICE=1 asks for it)"""
    import ctypes
    import torch
    monkeypatch.setenv("BLITZAR_AMD_RCCL_SINGLE_DEVICE", "1")
    api = gpu_backend
    lib = api.load()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(77)
    for cid, n in ((0, 3000), (2, 1500)):
        gens = util.generators_for(cid, n)
        d_gens = torch.from_numpy(np.ascontiguousarray(util.api_generators(cid, gens)).copy()).to(dev)
        cols = [rng.integers(0, 256, (n - 5 * c, 32 if c % 2 == 0 else 7), dtype=np.uint8)
                for c in range(5)]
        want = oracle.commit(cid, [(c, False) for c in cols], gens)
        d_cols = [torch.from_numpy(c.copy()).to(dev) for c in cols]
        desc = (api.sxt_sequence_descriptor * len(cols))()
        for i, c in enumerate(cols):
            desc[i] = api.sxt_sequence_descriptor(c.shape[1], c.shape[0], d_cols[i].data_ptr(), 0)
        out = torch.zeros((len(cols), want.shape[1]), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        D = lib.bzamd_num_devices()
        outs = (ctypes.c_void_p * D)(*([out.data_ptr()] + [None] * (D - 1)))
        gptr = (ctypes.c_void_p * D)(*([d_gens.data_ptr()] * D))
        lib.bzamd_msm_multi_device(cid, outs, len(cols), desc, gptr)
        assert np.array_equal(out.cpu().numpy(), want)
        assert lib.bzamd_multi_device_exchange() == b"rccl"


PIPELINE_BENCH = os.path.join(ROOT, "tools", "pipeline_bench", "_build", "pipeline_bench")


@pytest.mark.gpu
def test_stream_arrangements_of_the_throughput_mode_agree():
    """the throughput mode against plain calls (knobs are read once per context, hence one native
    process each: tools/pipeline_bench): never forking, the default, the NULL stream as the caller's,
    and both forms of the bucket reduction -- every step of every variant must produce the same
    commitments (the tool checks the steps against each other and prints a hash over them)"""
    if not os.path.exists(PIPELINE_BENCH):  # normally built by __graft_entry__.build()
        src_dir = os.path.dirname(os.path.dirname(PIPELINE_BENCH))
        lib_dir = os.path.join(ROOT, "blitzar_amd", "lib")
        os.makedirs(os.path.dirname(PIPELINE_BENCH), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                        "-I" + src_dir, os.path.join(src_dir, "pipeline_bench.cc"),
                        "-L" + lib_dir, "-lblitzar_amd", "-Wl,-rpath," + lib_dir, "-o", PIPELINE_BENCH],
                       check=True)
    arrangements = [
        ({"BLITZAR_AMD_OVERLAP_TAILS": "0"}, []),
        ({}, []),
        ({}, ["--null-stream"]),
        ({"BLITZAR_AMD_COMPACT_REDUCE": "0"}, []),
        ({"BLITZAR_AMD_COMPACT_REDUCE": "1"}, []),
    ]
    # (2^19 rows x 16 windows: long enough for the throughput mode's own reduce geometry, plan.h)
    # (--skew: two rows in three hold one scalar -- oversized bucket groups, the chunked sort path
    # with its in-launch barrier, the heavy-bucket folding of k_reduce)
    for curve, log2n, columns, shape in ((0, 15, 1, []), (2, 13, 3, []), (0, 19, 1, []),
                                         (0, 19, 1, ["--skew"]), (2, 17, 2, ["--skew"])):
        hashes = set()
        for env_extra, more in arrangements:
            flags = more + shape
            env = {k: v for k, v in os.environ.items() if not k.startswith("BLITZAR_AMD_")}
            env["BLITZAR_AMD_NUM_DEVICES"] = "1"
            env.update(env_extra)
            r = subprocess.run([PIPELINE_BENCH, "--curve", str(curve), "--log2n", str(log2n),
                                "--columns", str(columns), "--steps", "12", "--warmup", "3"] + flags,
                               env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, (env_extra, r.stdout[-2000:], r.stderr[-2000:])
            line = json.loads(r.stdout.strip().splitlines()[-1])
            assert line["outputs_agree"], (env_extra, line)
            hashes.add(line["hash"])
        assert len(hashes) == 1, hashes


CONCURRENT_CHILD = r"""
import json, os, sys, threading
import numpy as np
sys.path.insert(0, {root!r})
backend = {backend}
if backend == 2:
    import torch
from blitzar_amd import api
from tests import util
lib = api.load()
assert api.init(backend, 64) == 0
assert lib.bzamd_num_devices() == 2
lib.bzamd_set_shard_min_bytes(1 << 40)   # every call stays on ONE of the two devices
n = {n}
gens = util.generators_for(0, n)
g = util.api_generators(0, gens)
cols = [[(np.random.default_rng(10 * t + k).integers(0, 256, (n, 32), dtype=np.uint8), False)
         for k in range(3)] for t in range(2)]
results = [[None] * {calls} for _ in range(2)]
start = threading.Barrier(2)
def worker(t):
    start.wait()
    for k in range({calls}):
        results[t][k] = api.compute_pedersen_commitments(0, cols[t], generators=g).tolist()
threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
for th in threads: th.start()
for th in threads: th.join()
print("RESULT" + json.dumps({{"results": results,
                              "high_water": int(lib.bzamd_concurrent_calls_high_water())}}))
"""


def _concurrent_callers(oracle, backend, n, calls):
    env = dict(os.environ, BLITZAR_AMD_FORCE_SHARDS="2")
    env.pop("BLITZAR_BACKEND", None)
    code = CONCURRENT_CHILD.format(root=ROOT, backend=backend, n=n, calls=calls)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(next(ln for ln in r.stdout.splitlines() if ln.startswith("RESULT"))[6:])
    gens = util.generators_for(0, n)
    for t in range(2):
        cols = [(np.random.default_rng(10 * t + k).integers(0, 256, (n, 32), dtype=np.uint8), False)
                for k in range(3)]
        want = oracle.commit(0, cols, gens).tolist()
        assert all(res == want for res in out["results"][t]), f"thread {t}: wrong commitments"
    return out["high_water"]


@pytest.mark.gpu
def test_two_caller_threads_run_side_by_side_on_two_devices(oracle):
    """blocking sxt_* calls take per-device leases (round 3: one process-wide lock): two host
    threads on a backend of two (logical) devices are inside the library at the same time, each on
    its own device, and both get the reference's commitments
    (reference: thread_local per-call state, sxt/execution/schedule/scheduler.cc:65-69)"""
    assert _concurrent_callers(oracle, 2, 1 << 16, 12) >= 2

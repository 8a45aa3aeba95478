"""N > 1 path on CPU: world_size 2 over gloo, host backend (SXT_CPU_BACKEND) behind the same
C ABI.  Column sharding and row sharding must both reproduce the single-process commitments
bit for bit; the GPU run replaces gloo by RCCL and the host backend by the HIP
engine, nothing else."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, curve_id, result_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from blitzar_amd import api, distributed
    from tests import util
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert api.init(api.SXT_CPU_BACKEND, 0) == 0
    rng = np.random.default_rng(77 + curve_id)
    n = 61
    gens = util.api_generators(curve_id, util.generators_for(curve_id, n))
    cols = util.mixed_columns(rng, n)[:9]  # 9 columns over 2 ranks: ragged shards
    by_cols = distributed.commit_columns_sharded(curve_id, cols, gens, dist)
    by_rows = distributed.commit_rows_sharded(curve_id, cols, gens, dist)
    single = api.compute_pedersen_commitments(curve_id, cols, generators=gens)
    np.savez(os.path.join(result_dir, f"rank{rank}.npz"), by_cols=by_cols, by_rows=by_rows,
             single=single)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_is_a_balanced_partition():
    from blitzar_amd import distributed
    for units in (0, 1, 7, 8, 9, 256, 1024, 1000003):
        for world in (1, 2, 3, 4, 8):
            prev = 0
            sizes = []
            for r in range(world):
                b, e = distributed.shard_range(units, r, world)
                assert b == prev and e >= b
                prev = e
                sizes.append(e - b)
            assert prev == units and max(sizes) - min(sizes) <= 1
            assert sizes == distributed.shard_counts(units, world)


@pytest.mark.parametrize("curve_id", [0, 2])
def test_world2_column_and_row_sharding(curve_id, tmp_path, oracle):
    import torch.multiprocessing as mp
    from tests import util
    port = 29000 + (os.getpid() % 2000) + curve_id
    mp.spawn(_worker, args=(2, port, curve_id, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # the reference oracle on the same inputs
    rng = np.random.default_rng(77 + curve_id)
    gens = util.generators_for(curve_id, 61)
    cols = util.mixed_columns(rng, 61)[:9]
    want = oracle.commit(curve_id, cols, gens)
    for r in (r0, r1):
        assert np.array_equal(r["single"], want)
        assert np.array_equal(r["by_cols"], want)
        assert np.array_equal(r["by_rows"], want)

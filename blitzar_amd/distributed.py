"""Multi-GPU sharding of the MSM path: one process per GPU, torch.distributed for the exchange
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference has no communication layer: one host thread hands generator chunks / output
chunks to the least-busy device and partials bounce through pinned host memory
(sxt/execution/device/for_each.cc:56-82, sxt/multiexp/bucket_method/accumulation.h:134-161,
sxt/multiexp/pippenger2/multiexponentiation.h:112-133).  Here the path shards two ways
(SURVEY.md section 8(e)), neither of which needs a data-path collective:

  * columns  -- commitment columns are independent units: rank r commits columns
                [r*S/N, (r+1)*S/N); one all-gather of the canonical encodings (<= 72 B each).
  * rows     -- one long column split by generator rows: every rank produces a projective
                partial, one all-gather of the partials (<= 160 B each), then every rank folds
                them and canonicalises.  RCCL has no user-defined reduction, so all-gather +
                local adds replaces an all-reduce; group addition is exact, so the result is
                bit-identical to the unsharded one.

Payloads are tens of bytes to a few KB: latency-bound, the 7 x 153 GB/s xGMI links never matter.
This module is host plumbing only; all arithmetic happens behind the C ABI.
"""
import numpy as np

from blitzar_amd import api


def shard_range(num_units, rank, world):
    """contiguous, balanced split: the first (num_units % world) ranks get one extra unit"""
    base, extra = divmod(num_units, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_counts(num_units, world):
    return [shard_range(num_units, r, world)[1] - shard_range(num_units, r, world)[0]
            for r in range(world)]


def all_gather_rows(local, counts, dist, device=None):
    """all-gather of per-rank [count_r, width] uint8 blocks -> [sum(counts), width] on every rank.
    Blocks are padded to the largest count so a single fixed-size all_gather_into_tensor serves
    ragged shards (one collective, as cheap as it gets for latency-bound payloads)."""
    import torch
    world = dist.get_world_size()
    width = local.shape[1]
    cap = max(max(counts), 1)
    send = torch.zeros((cap, width), dtype=torch.uint8, device=device or local.device)
    if local.shape[0]:
        send[:local.shape[0]] = local
    recv = torch.empty((world * cap, width), dtype=torch.uint8, device=send.device)
    dist.all_gather_into_tensor(recv, send)
    parts = [recv[r * cap:r * cap + counts[r]] for r in range(world)]
    return torch.cat(parts, dim=0)


def commit_columns_sharded(curve_id, columns, generators, dist):
    """Host-buffer variant used by the CPU (gloo) tests and small jobs: every rank holds the full
    column list, commits its shard through the drop-in Pedersen entry point, and the encodings are
    all-gathered.  Returns uint8 [num_columns, commitment bytes] on every rank."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    cols = list(columns)
    begin, end = shard_range(len(cols), rank, world)
    width = api.CURVE_LAYOUT[curve_id][1]
    if end > begin:
        local = api.compute_pedersen_commitments(curve_id, cols[begin:end], generators=generators)
    else:
        local = np.zeros((0, width), np.uint8)
    out = all_gather_rows(torch.from_numpy(local), shard_counts(len(cols), world), dist)
    return out.cpu().numpy()


def commit_rows_sharded(curve_id, columns, generators, dist):
    """Row sharding: every column is cut into `world` contiguous row ranges; rank r computes
    projective partials over its rows (and the matching generator slice), partials are
    all-gathered and folded + canonicalised locally on every rank."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    cols = [(np.ascontiguousarray(a), s) for a, s in columns]
    longest = max((a.shape[0] for a, _ in cols), default=0)
    begin, end = shard_range(longest, rank, world)
    stride = api.CURVE_LAYOUT[curve_id][0]
    gens = np.ascontiguousarray(generators).view(np.uint8).reshape(-1, stride)
    local_cols = [(a[min(begin, a.shape[0]):min(end, a.shape[0])], s) for a, s in cols]
    psize = api.CURVE_LAYOUT[curve_id][2]
    # an empty generator slice still needs a valid pointer for the ABI
    gslice = gens[begin:end] if end > begin else gens[:1]
    local = api.msm_projective(curve_id, local_cols, gslice)
    gathered = all_gather_rows(torch.from_numpy(local.reshape(1, -1)), [1] * world, dist)
    partials = gathered.cpu().numpy().reshape(world, len(cols), psize)
    return api.fold_encode(curve_id, partials)

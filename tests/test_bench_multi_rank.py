"""bench.py's N > 1 control flow on a one-GPU box (-m gpu): N ranks, all on cuda:0, collectives
through gloo on host copies (--dry-run-one-gpu; never a measurement).  Checks that the compact last
line parses strictly, stays under 4 KB and carries the contract's members, and that the detail file
carries the weak-scaling value, the row-split strong-scaling leg of the headline column (whose commitments
must equal rank 0's commitment of the whole column), the distributed block, and -- with the oracle --
the verified headline commitment of every rank and config 4's 256 columns sharded over the ranks
(at reduced rows: --log2n / --config4-log2n), inside a wall-clock and a device-memory budget."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_dry_run(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "4", "--warmup", "2", "--dry-run-one-gpu", "--no-cpu-baseline",
           "--no-configs", "--detail-file", str(tmp_path / "detail.json")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _records(r.stdout, tmp_path / "detail.json")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    strong = d["strong_scaling_config2"]
    assert strong["scaling"] == "strong" and strong["rows_per_gpu"] == 1 << 19
    assert strong["ms_per_step"] > 0 and "verified" in strong
    assert d["distributed"]["rccl_world_size"] == 2


def _strict(line):
    def refuse(name):
        raise ValueError(f"{name} is not JSON")
    return json.loads(line, parse_constant=refuse)


def _records(stdout, detail_path):
    """the compact record = the LAST stdout line, strict JSON under 4 KB with the contract's members;
    returns the detail file's record (a superset)"""
    lines = stdout.splitlines()
    line = lines[-1]
    assert len(line) < 4096, len(line)
    c = _strict(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "verified"):
        assert k in c, k
    assert "workload" in c["config"]
    with open(detail_path) as fh:
        d = _strict(fh.read())
    assert d["value"] == pytest.approx(c["value"], rel=1e-4) and d["n_gpus"] == c["n_gpus"]
    assert c["strong_scaling_config2"]["rows_per_gpu"] == d["strong_scaling_config2"]["rows_per_gpu"]
    return d


def _dry_run(ranks, port, extra, timeout, tmp_path):
    extra = extra + ["--detail-file", str(tmp_path / "detail.json")]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(ranks), "--dry-run-one-gpu"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    return _records(r.stdout, tmp_path / "detail.json")


@pytest.mark.gpu
def test_two_ranks_dry_run_with_oracle_and_sharded_config4(oracle, tmp_path):
    """the legs the plain dry run drops: every rank verifies its timed commitment against the
    reference CPU backend, and config 4's 256 columns are sharded 128 / 128 with one all-gather, every
    rank checking its shard inside the gathered buffer (2^14 / 2^12 rows: seconds of oracle)"""
    d = _dry_run(2, 29619, ["--steps", "3", "--warmup", "1", "--log2n", "14", "--config4-log2n", "12",
                            "--config-steps", "2"], 900, tmp_path)
    assert d["n_gpus"] == 2 and "verified" in d
    sharded = d["strong_scaling"]
    assert sharded["columns_per_gpu"] == 128 and sharded["ms_per_call"] > 0
    assert "bit-exact" in sharded["verified"]
    assert d["strong_scaling_config2"]["rows_per_gpu"] == 1 << 13
    assert d["distributed"]["rccl_world_size"] == 2


@pytest.mark.gpu
def test_eight_ranks_dry_run_inside_budget(oracle, tmp_path):
    """first contact with --gpus 8 on what one GPU can show: 8 processes, 8 HIP contexts and engine
    workspaces on cuda:0, the rendezvous, the three collectives of the line at world size 8, config 4
    sharded 32 columns per rank -- inside ten minutes of wall clock and, all eight processes together,
    half of the one GPU's memory (measured: ~11.6 GB per process, nearly all of it the HIP runtime's
    own per-process reservations -- scratch for kernels with private stacks on every queue, code objects;
    on a real node every rank has a 288 GB device to itself)"""
    d = _dry_run(8, 29621, ["--steps", "2", "--warmup", "1", "--log2n", "12", "--config4-log2n", "10",
                            "--config-steps", "1"], 600, tmp_path)
    assert d["n_gpus"] == 8 and d["distributed"]["rccl_world_size"] == 8
    assert d["strong_scaling"]["columns_per_gpu"] == 32
    assert d["strong_scaling_config2"]["rows_per_gpu"] == 1 << 9
    assert d["dry_run"]["ranks_on_one_gpu"] == 8
    assert d["dry_run"]["device_bytes_in_use"] < 144 << 30, d["dry_run"]
    assert d["dry_run"]["wall_s_since_start"] < 600

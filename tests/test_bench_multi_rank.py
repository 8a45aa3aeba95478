"""bench.py's N > 1 control flow on a one-GPU box (-m gpu): two ranks, both on cuda:0, collectives
through gloo on host copies (--dry-run-one-gpu; never a measurement).  Checks that the line carries
the weak-scaling value, the row-split strong-scaling leg of the headline column (whose commitments
must equal rank 0's commitment of the whole column) and the distributed block."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_dry_run():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "4", "--warmup", "2", "--dry-run-one-gpu", "--no-cpu-baseline",
           "--no-configs"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    strong = d["strong_scaling_config2"]
    assert strong["scaling"] == "strong" and strong["rows_per_gpu"] == 1 << 19
    assert strong["ms_per_step"] > 0 and "verified" in strong
    assert d["distributed"]["rccl_world_size"] == 2

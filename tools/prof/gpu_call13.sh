#!/bin/bash
set -u
OUT=gpurun_out/r3c13
mkdir -p $OUT
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --dry-run-one-gpu --config-steps 1 > $OUT/dry2.json 2> $OUT/dry2.err
echo "dry rc=$?"; tail -5 $OUT/dry2.err
python - <<'PY'
import json
lines=[l for l in open('gpurun_out/r3c13/dry2.json') if l.startswith('{')]
d=json.loads(lines[-1])
print({k:d.get(k) for k in ('value','n_gpus','ms_per_step','scaling')})
print(json.dumps(d.get('distributed'))[:1500])
print(json.dumps(d.get('strong_scaling'))[:600])
PY

#!/bin/bash
# package power and shader clock while the engine runs (rocm-smi samples beside the native driver):
# a sequence of config-2 calls, then bn254 accumulations (64 columns x 2^20 rows per call)
# usage (repo root, on the GPU box): tools/prof/power_clocks.sh > gpurun_out/power_clocks.txt
echo "---- idle"
rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "power (W)\|sclk\|mclk\|junction"
echo "---- sequence of 6000 config-2 calls (curve25519, 2^20 rows, throughput mode)"
tools/pipeline_bench/_build/pipeline_bench --steps 6000 --warmup 10 > /tmp/pb.json &
PB=$!
sleep 2.5
for i in 1 2 3; do rocm-smi --showpower --showclocks 2>&1 | grep -i "power (W)\|sclk"; sleep 0.7; done
wait $PB
python3 -c "import json; d=json.loads(open('/tmp/pb.json').read().strip().splitlines()[-1]); print('ms per call in sequence', d['ms_per_step'], 'lone', d['lone_ms'])"
echo "---- bn254, 64 columns x 2^20 rows per call"
tools/pipeline_bench/_build/pipeline_bench --curve 2 --columns 64 --log2n 20 --steps 40 --warmup 2 > /tmp/pb2.json &
PB=$!
sleep 3
for i in 1 2 3; do rocm-smi --showpower --showclocks 2>&1 | grep -i "power (W)\|sclk"; sleep 0.7; done
wait $PB

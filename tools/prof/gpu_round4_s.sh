#!/bin/bash
# round 4, nineteenth GPU session: how many lead columns the row pipeline should cut (10 columns,
# caller generators)
set -u
OUT=gpurun_out/r4s
mkdir -p $OUT
for lead in 2 3 4 6 10; do
  BLITZAR_AMD_ROW_PIPELINE_LEAD=$lead tools/pipeline_bench/_build/hostapi_bench --samples 8 --warmup 2 --only 10 caller > $OUT/lead_$lead.json 2>&1
  python - $OUT/lead_$lead.json $lead <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["cases"][0]
print("lead", sys.argv[2], "10 columns caller: mean %.3f min %.3f median %.3f" % (c["ms_mean"], c["ms_min"], c["ms_median"]))
PY
done

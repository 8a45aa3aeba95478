#!/bin/bash
# A/B of engine knobs through the native driver (tools/pipeline_bench), on one box.
# usage: tools/prof/ab_pipeline.sh <out file> <bench args...> -- <spec> [<spec> ...]
#   spec: "K=V,K=V" (environment of the variant; "-" = defaults), optionally "@<library dir>"
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
OUT=$1; shift
ARGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
shift
mkdir -p "$(dirname "$OUT")"
EXE=tools/pipeline_bench/_build/pipeline_bench
for rep in ${AB_REPS:-1 2}; do
for spec in "$@"; do
  envs=${spec%@*}; lib=""
  case "$spec" in *@*) lib=${spec#*@};; esac
  [ "$envs" = "-" ] && envs=""
  echo "== $spec ${ARGS[*]}" >> "$OUT"
  if [ -n "$lib" ]; then
    env $(echo "$envs" | tr ',' ' ') LD_LIBRARY_PATH=$PWD/$lib timeout 120 $EXE "${ARGS[@]}" >> "$OUT" 2>&1
  else
    env $(echo "$envs" | tr ',' ' ') timeout 120 $EXE "${ARGS[@]}" >> "$OUT" 2>&1
  fi
  echo "rc=$?" >> "$OUT"
done
done

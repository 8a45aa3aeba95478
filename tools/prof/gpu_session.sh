#!/bin/bash
# One GPU session on a gpurun box, as a list of steps (replaces the per-session scripts of rounds 3-4):
#   gpurun --timeout 1500 -- 'bash tools/prof/gpu_session.sh <tag> step [step ...]'
# Everything is written under gpurun_out/<tag>/ (merged back by gpurun).  Steps:
#   info        which box: amd-smi static, rocm-smi clocks / power / partitions
#   tests       python -m pytest tests -m gpu -x -q
#   tests:<k>   the same with -k <k>
#   bench       python bench.py (the driver's default line)            -> bench.json
#   bench-quick python bench.py --skip-headline-check                  -> bench_quick.json
#   bench-detail python bench.py --detail (configs 1/3/4/5, trace, host API) -> bench_full*.json
#   profile-headline  tools/prof/run_pmc.sh: the same four passes of the driver's own command
#   profile     tools/prof/run_pmc_configs.sh: rocprofv3 --kernel-trace --stats + three --pmc passes
#               (SQ_*, FETCH_SIZE, WRITE_SIZE; no tracing beside counters) -> gpurun_out/prof_<tag>/
#   ab:<spec-file>   tools/prof/ab_pipeline.sh over the lines of <spec-file> ("<bench args> -- <specs>")
#   grid        the reference's benchmark grid + the short-column / many-column regime -> grid.json
#   cmd:<file>  bash <file> (a one-off, kept under gpurun_out/)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for step in "$@"; do
  echo "==== $step $(date +%T)" | tee -a "$OUT/session.log"
  case "$step" in
    info)
      (amd-smi static --asic --vbios --json; rocm-smi --showclocks --showpower --showcomputepartition \
        --showmemorypartition --showperflevel; nproc) > "$OUT/info.txt" 2>&1 ;;
    tests)
      timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$OUT/gputests.log" 2>&1
      tail -3 "$OUT/gputests.log" | tee -a "$OUT/session.log" ;;
    tests:*)
      timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "${step#tests:}" > "$OUT/gputests_k.log" 2>&1
      tail -3 "$OUT/gputests_k.log" | tee -a "$OUT/session.log" ;;
    bench)
      timeout -k 10 400 python bench.py --detail-file "$OUT/bench_detail.json" > "$OUT/bench.json" 2> "$OUT/bench.err"
      echo "rc=$? $(wc -c < "$OUT/bench.json") bytes" | tee -a "$OUT/session.log" ;;
    bench-detail)
      timeout -k 10 900 python bench.py --detail --detail-file "$OUT/bench_full_detail.json" > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"
      echo "rc=$? $(wc -c < "$OUT/bench_full_detail.json") bytes of detail" | tee -a "$OUT/session.log" ;;
    profile-headline)
      bash tools/prof/run_pmc.sh "${TAG}_headline" > "$OUT/profile_headline.log" 2>&1
      echo "rc=$?" | tee -a "$OUT/session.log" ;;
    bench-quick)
      timeout 600 python bench.py --skip-headline-check > "$OUT/bench_quick.json" 2> "$OUT/bench_quick.err"
      echo "rc=$?" | tee -a "$OUT/session.log" ;;
    profile)
      # rocprofv3 --kernel-trace --stats + three --pmc passes of bench.py with its configs legs
      bash tools/prof/run_pmc_configs.sh "$TAG" > "$OUT/profile.log" 2>&1
      echo "rc=$?" | tee -a "$OUT/session.log"; grep "rc=" "$OUT/profile.log" | tee -a "$OUT/session.log" ;;
    ab:*)
      while IFS= read -r line; do
        [ -z "$line" ] && continue
        case "$line" in \#*) continue;; esac
        # shellcheck disable=SC2086
        bash tools/prof/ab_pipeline.sh "$OUT/ab.log" $line
      done < "${step#ab:}"
      tail -40 "$OUT/ab.log" ;;
    grid)
      timeout -k 10 420 python tools/grid_bench.py --out "$OUT/grid.json" > "$OUT/grid.log" 2>&1
      echo "rc=$?" | tee -a "$OUT/session.log"; tail -5 "$OUT/grid.log" ;;
    cmd:*)
      timeout -k 10 300 bash "${step#cmd:}" "$TAG" > "$OUT/cmd_$(basename "${step#cmd:}").log" 2>&1
      echo "rc=$?" | tee -a "$OUT/session.log" ;;
    *) echo "unknown step $step" | tee -a "$OUT/session.log" ;;
  esac
done
echo "==== done $(date +%T)" | tee -a "$OUT/session.log"

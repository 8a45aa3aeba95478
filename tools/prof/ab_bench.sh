# A/B of two builds on one MI355X box (box-to-box variance is ~10 %): blitzar_amd/lib/ab/base.so (a copy
# of an earlier libblitzar_amd.so) against the current library, alternating, via BLITZAR_AMD_LIB.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in blitzar_amd/lib/ab/base.so blitzar_amd/lib/libblitzar_amd.so; do
  echo "== $lib"
  BLITZAR_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | sed 's/.*"ms_per_step": \([0-9.]*\).*"resident_generators_ms_per_step": \([0-9.]*\), "stage_ms": \({[^}]*}\).*/ms \1 resident \2 \3/'
done
done

# several builds of the library, alternated twice, at both warm-ups: bash scripts/gpu_libs2.sh "a.so b.so"
set -e
cd $GRAFT_REPO_ROOT
LIBS=$1; shift
for w in 5 20; do for rep in 1 2; do for lib in $LIBS; do
  SPH_HIP_LIB=$PWD/$lib timeout -k 10 250 python bench.py --steps 20 --warmup $w --no-cpu-baseline "$@" > gpurun_out/libs.json
  python - <<PY
import json
d=json.load(open("gpurun_out/libs.json"))
print("w=$w $lib", round(d["ms_per_step"],4), d["stage_ms"]["build_grid_ms"], d["stage_ms"]["density_ms"], d["stage_ms"]["force_ms"])
PY
done; done; done

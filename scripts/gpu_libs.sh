# several builds of the library on one workload: bash scripts/gpu_libs.sh "a.so b.so" [bench args]
set -e
cd $GRAFT_REPO_ROOT
LIBS=$1; shift
for lib in $LIBS; do
  SPH_HIP_LIB=$PWD/$lib timeout -k 10 250 python bench.py --steps 20 --warmup 20 --no-cpu-baseline "$@" > gpurun_out/libs.json
  python - <<PY
import json
d=json.load(open("gpurun_out/libs.json"))
print("$lib", round(d["ms_per_step"],4), d["stage_ms"]["build_grid_ms"], d["stage_ms"]["density_ms"], d["stage_ms"]["force_ms"])
PY
done

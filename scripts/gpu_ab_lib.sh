# A/B of two builds of the library on one workload: bash scripts/gpu_ab_lib.sh <other .so> [bench args]
set -e
cd $GRAFT_REPO_ROOT
OTHER=$1; shift
for lib in sph_amd/libsph_hip.so $OTHER sph_amd/libsph_hip.so $OTHER; do
  SPH_HIP_LIB=$PWD/$lib timeout -k 10 250 python bench.py --steps 20 --warmup 20 --no-cpu-baseline "$@" > gpurun_out/ab_lib.json
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_lib.json"))
print("$lib", round(d["ms_per_step"],4), d["stage_ms"])
PY
done

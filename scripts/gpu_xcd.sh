# sweep of the XCD run length (SPH_HIP_XCD_RUN) on one workload: bash scripts/gpu_xcd.sh "16 32 64" [bench args]
set -e
cd $GRAFT_REPO_ROOT
RUNS=$1; shift
for r in $RUNS; do
  SPH_HIP_XCD_RUN=$r timeout -k 10 250 python bench.py --steps 20 --warmup 20 --no-cpu-baseline "$@" > gpurun_out/xcd.json
  python - <<PY
import json
d=json.load(open("gpurun_out/xcd.json"))
print("run $r", round(d["ms_per_step"],4), d["stage_ms"]["density_ms"], d["stage_ms"]["force_ms"])
PY
done

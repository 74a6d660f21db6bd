# A/B of an environment switch on the default bench: bash scripts/gpu_ab.sh VAR
set -e
cd $GRAFT_REPO_ROOT
V=${1:-SPH_HIP_UNIFORM_MASS}
for val in 1 0 1 0; do
  env $V=$val timeout -k 10 200 python bench.py --steps 30 --warmup 20 --no-cpu-baseline $BENCH_ARGS > gpurun_out/ab_$val.json
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$val.json"))
print("$V=$val", round(d["ms_per_step"],4), d["stage_ms"])
PY
done

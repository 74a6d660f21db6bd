# run-to-run spread of the headline number on one box: bench.py with the driver's arguments, REPS times
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in $(seq 1 ${REPS:-6}); do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/rep.json
  python - <<PY
import json
d=json.load(open("gpurun_out/rep.json"))
s=d["stage_ms"]
print("rep $rep ms_per_step", round(d["ms_per_step"],4), "grid", s["build_grid_ms"], "density", s["density_ms"], "force", s["force_ms"], "tail", s["full_step_ms"])
PY
done

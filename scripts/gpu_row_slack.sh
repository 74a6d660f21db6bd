# the force kernel's row gate (SPH_HIP_ROW_SLACK): parity tests at slack 0 and 1, then frame / stage times for every slack
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for s in 0 1; do
  SPH_HIP_ROW_SLACK=$s timeout -k 10 300 python -u -m pytest -x -q --timeout 200 --timeout-method thread -p no:cacheprovider tests/test_force_paths_gpu.py tests/test_parity_gpu.py tests/test_golden_gpu.py > gpurun_out/t_slack$s.log 2>&1 || { tail -20 gpurun_out/t_slack$s.log; exit 1; }
  echo "slack $s: $(tail -1 gpurun_out/t_slack$s.log)"
done
for w in 5 20; do for s in 8 0 1 2 3 4 8; do
  SPH_HIP_ROW_SLACK=$s timeout -k 10 200 python bench.py --steps 20 --warmup $w --no-cpu-baseline > gpurun_out/slack.json
  python - <<PY
import json
d=json.load(open("gpurun_out/slack.json"))
print("w=$w slack=$s ms_per_step", round(d["ms_per_step"],4), "force", d["stage_ms"]["force_ms"], "density", d["stage_ms"]["density_ms"])
PY
done; done

set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2 3; do for w in 5 20; do for s in 8 4 5 6 3; do
  SPH_HIP_ROW_SLACK=$s timeout -k 10 200 python bench.py --steps 20 --warmup $w --no-cpu-baseline > gpurun_out/slack.json
  python - <<PY
import json
d=json.load(open("gpurun_out/slack.json"))
print("rep $rep w=$w slack=$s ms_per_step", round(d["ms_per_step"],4), "force", d["stage_ms"]["force_ms"])
PY
done; done; done

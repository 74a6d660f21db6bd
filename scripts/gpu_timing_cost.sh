# what the per-stage timing costs in frame time: SPH_HIP_TIMING=0 / stride 4 / stride 2 (default) / every frame
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do for mode in "0 2" "1 4" "1 2" "2 1"; do
  set -- $mode
  SPH_HIP_TIMING=$1 SPH_HIP_TIMING_STRIDE=$2 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/tc.json
  python - <<PY
import json
d=json.load(open("gpurun_out/tc.json"))
print("timing=$1 stride=$2 ms_per_step", round(d["ms_per_step"],4))
PY
done; done

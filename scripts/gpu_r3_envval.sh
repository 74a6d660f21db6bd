# A/B of values of one environment variable of the library: bash scripts/gpu_r3_envval.sh VAR "v1 v2 ..." [bench args]
set -e
cd $GRAFT_REPO_ROOT
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do
  for val in $VALS; do
    for w in 5 20; do
      env $VAR=$val timeout -k 10 200 python bench.py --steps 20 --warmup $w --no-cpu-baseline "$@" > gpurun_out/envval.json
      python3 -c "
import json; d=json.load(open('gpurun_out/envval.json')); print('$VAR=$val w=$w rep=$rep', round(d['ms_per_step'],4), {k: v for k,v in d['stage_ms'].items() if v})"
    done
  done
done

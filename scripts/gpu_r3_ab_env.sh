# A/B of several builds of the library with an environment setting in force: bash scripts/gpu_r3_ab_env.sh "VAR=val" "<libs>"
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
  for lib in $2; do
    for w in 5 20; do
      env $1 SPH_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 20 --warmup $w --no-cpu-baseline > gpurun_out/ab_env.json
      python3 -c "
import json; d=json.load(open('gpurun_out/ab_env.json')); print('$1 $(basename $lib .so) w=$w rep=$rep', round(d['ms_per_step'],4), {k: v for k,v in d['stage_ms'].items() if v})"
    done
  done
done

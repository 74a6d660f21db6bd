# round 3: A/B of an environment switch of the library on one box.  bash scripts/gpu_r3_env.sh VAR [tests]
#   runs the core parity tests with the default (switch on), then the bench at both warm-ups with VAR unset and VAR=0, twice
set -e
cd $GRAFT_REPO_ROOT
VAR=$1
mkdir -p gpurun_out
if [ "$2" = "tests" ]; then
  timeout -k 10 900 python -u -m pytest -x -q --timeout 300 --timeout-method thread -p no:cacheprovider tests/test_parity_gpu.py tests/test_force_paths_gpu.py tests/test_golden_gpu.py tests/test_robustness_gpu.py tests/test_random_scenes_gpu.py tests/test_stepwise_gpu.py tests/test_cpp_host_gpu.py -m gpu > gpurun_out/t_quick.log 2>&1 || { tail -60 gpurun_out/t_quick.log; exit 1; }
  tail -2 gpurun_out/t_quick.log
fi
for rep in 1 2; do
  for val in on 0; do
    for w in 5 20; do
      if [ $val = on ]; then
        timeout -k 10 200 python bench.py --steps 20 --warmup $w --no-cpu-baseline > gpurun_out/env_ab.json
      else
        env $VAR=0 timeout -k 10 200 python bench.py --steps 20 --warmup $w --no-cpu-baseline > gpurun_out/env_ab.json
      fi
      python3 -c "
import json; d=json.load(open('gpurun_out/env_ab.json')); print('$VAR=$val w=$w rep=$rep', round(d['ms_per_step'],4), {k: v for k,v in d['stage_ms'].items() if v})"
    done
  done
done
SPH_HIP_LDS_STATS=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep libsph_hip || true

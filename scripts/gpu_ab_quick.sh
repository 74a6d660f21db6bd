# quick loop: core parity tests, then bench at the driver's and the default warm-up (no CPU baseline)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest -x -q --timeout 300 --timeout-method thread -p no:cacheprovider tests/test_parity_gpu.py tests/test_force_paths_gpu.py tests/test_golden_gpu.py tests/test_robustness_gpu.py tests/test_slab_gpu.py -m gpu > gpurun_out/t_quick.log 2>&1 || { tail -40 gpurun_out/t_quick.log; exit 1; }
tail -2 gpurun_out/t_quick.log
for w in 5 20; do
  for rep in 1 2; do
    timeout -k 10 200 python bench.py --steps 20 --warmup $w --no-cpu-baseline $BENCH_ARGS | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('w=$w', round(d['ms_per_step'],4), d['stage_ms'])"
  done
done

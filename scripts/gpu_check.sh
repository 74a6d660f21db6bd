# parity tests, then a short bench (no CPU baseline) -- the usual edit/measure loop
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -15
timeout -k 10 300 python bench.py --steps 20 --warmup 20 --no-cpu-baseline $BENCH_ARGS | tee gpurun_out/bench_last.json

# what the round's end runs, in its order: build check is done on the CPU side; here smoke(), the GPU suite, the bench
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
timeout -k 10 900 python -u -m pytest -x -q --timeout 600 --timeout-method thread -p no:cacheprovider tests -m gpu > gpurun_out/t_all.log 2>&1 && tail -3 gpurun_out/t_all.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/final_bench_w5.json && python - <<'PY'
import json
d=json.load(open("gpurun_out/final_bench_w5.json"))
print("bench w5:", round(d["ms_per_step"],4), "ms/step", round(d["value"]/1e9,3), "G/s", "force frac", round(d["roofline"]["frac"],4), "cpu", round(d["cpu_baseline"]["value"]/1e6,2), "M/s")
PY

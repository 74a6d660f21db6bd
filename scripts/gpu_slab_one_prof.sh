# world-1 slab run of config 5 (8.4 M particles) against the plain engine: wall time, stage times, host time per call,
# and a kernel trace of the slab run (which kernels does slab mode add, what do they cost?)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
N=${N:-8388608}
mkdir -p gpurun_out/slab1
timeout -k 10 200 python bench.py --workload config5 --particles $N --steps 10 --warmup 20 --no-cpu-baseline > gpurun_out/slab1/plain.json && python -c "import json; d=json.load(open('gpurun_out/slab1/plain.json')); print('PLAIN', d['ms_per_step'], d['stage_ms'])" &&
OVERLAP=0 GHOST=12 timeout -k 10 200 python scripts/slab_one.py $N > gpurun_out/slab1/classic.txt 2>&1 && cat gpurun_out/slab1/classic.txt &&
OVERLAP=1 GHOST=12 timeout -k 10 200 python scripts/slab_one.py $N > gpurun_out/slab1/overlap.txt 2>&1 && cat gpurun_out/slab1/overlap.txt &&
OVERLAP=1 GHOST=12 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/slab1/prof -o t -- python3 scripts/slab_one.py $N > gpurun_out/slab1/prof.txt 2>&1 &&
find gpurun_out/slab1/prof -name '*kernel_stats.csv' | head -1 | xargs -r head -30

# hardware-counter passes of the bench command (one --pmc set per run, see MI355X_MICROARCH.md)
# usage: [WARMUP=20 STEPS=5] bash scripts/gpu_pmc.sh <tag> "<counters>" [bench args]
# the summary averages each kernel's counters over the TIMED frames (the last STEPS dispatches)
set -e
TAG=$1; CTRS=$2; shift 2
WARMUP=${WARMUP:-20}; STEPS=${STEPS:-5}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -o pmc -- python3 bench.py --steps $STEPS --warmup $WARMUP --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
python3 scripts/pmc_summary.py $OUT/pmc_counter_collection.csv $STEPS | tee $OUT/summary.txt

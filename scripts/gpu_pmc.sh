# hardware-counter passes of the default bench command (one --pmc set per run, see MI355X_MICROARCH.md)
# usage: bash scripts/gpu_pmc.sh <tag> "<counters>" [bench args]
set -e
TAG=$1; CTRS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -o pmc -- python3 bench.py --steps 5 --warmup 20 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
python3 scripts/pmc_summary.py $OUT/pmc_counter_collection.csv | tee $OUT/summary.txt

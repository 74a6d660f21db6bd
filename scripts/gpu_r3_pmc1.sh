# counter passes of the bench (w5 s20) with an env switch of the library at 1 and at 0: bash scripts/gpu_r3_pmc1.sh VAR "<counters>" [tag]
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
VAR=$1; CTRS=$2; TAG=${3:-sq}
for val in 1 0; do
  OUT=gpurun_out/pmc1_${VAR}_${val}_$TAG
  rm -rf $OUT && mkdir -p $OUT
  export $VAR=$val
  timeout -s KILL 300 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -o pmc -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
  echo "== $VAR=$val"; python3 scripts/pmc_summary.py $OUT/pmc_counter_collection.csv 20 | tee $OUT/summary.txt | grep -A10 density
done

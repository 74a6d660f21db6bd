# round 3: A/B of several builds of the library on one box, the driver's window (--steps 20 --warmup 5) and the defaults.
#   bash scripts/gpu_r3_ab.sh "<lib a> <lib b> ..." [trace]     libs relative to the repo root; "trace" = also a kernel trace per lib
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
LIBS=${1:-sph_amd/libsph_hip.so}
OUT=gpurun_out/r3_ab
mkdir -p $OUT
for rep in 1 2; do
  for lib in $LIBS; do
    tag=$(basename $lib .so)
    for w in 5 20; do
      SPH_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 20 --warmup $w --no-cpu-baseline > $OUT/${tag}_w${w}_r${rep}.json
      python3 - <<PY
import json
d=json.load(open("$OUT/${tag}_w${w}_r${rep}.json"))
print("$tag w=$w rep=$rep", round(d["ms_per_step"],4), {k: round(v,4) for k,v in d["stage_ms"].items()})
PY
    done
  done
done
if [ "$2" = "trace" ]; then
  for lib in $LIBS; do
    tag=$(basename $lib .so)
    SPH_HIP_LIB=$PWD/$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -o trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/trace_$tag.json 2> $OUT/trace_$tag.err
    python3 scripts/trace_tail_stats.py $(find $OUT/trace_$tag -name '*kernel_trace.csv' | head -1) 20 > $OUT/timed_region_$tag.csv
    echo "== $tag"; head -14 $OUT/timed_region_$tag.csv
  done
fi

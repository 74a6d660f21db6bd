# kernel trace (timed frames only) of several builds of the library: bash scripts/gpu_trace_libs.sh "<libs>" [kernel-name filter]
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/trace_libs
rm -rf $OUT && mkdir -p $OUT
for lib in $1; do
  tag=$(basename $lib .so)
  SPH_HIP_LIB=$PWD/$lib timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/$tag -o trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/$tag.json 2> $OUT/$tag.err
  python3 scripts/trace_tail_stats.py $(find $OUT/$tag -name '*kernel_trace.csv' | head -1) 20 > $OUT/$tag.csv
  echo "== $tag"; grep -E "${2:-k_}" $OUT/$tag.csv | cut -d, -f1,4 | cut -c1-80
  rm -rf $OUT/$tag
done

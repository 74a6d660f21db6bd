# rocprofv3 kernel trace of the default bench command; summaries land in gpurun_out/prof_<tag>
set -e
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 bench.py --steps 20 --warmup 20 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
ls -R $OUT | head -30
find $OUT -name '*kernel_stats.csv' | head -1 | xargs -r head -30

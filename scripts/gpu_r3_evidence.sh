# round-3 evidence, one call: counter passes (config 3 at the driver's and the default arguments, config 5's per-GPU block),
# kernel traces of the two bench commands, the bench lines themselves (with the CPU baseline), small configurations
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
WARMUP=5 STEPS=20 TAGP=w5 bash scripts/gpu_r3_pmc.sh
WARMUP=20 STEPS=30 TAGP=w20 bash scripts/gpu_r3_pmc.sh
WARMUP=5 STEPS=20 TAGP=c5w5 bash scripts/gpu_r3_pmc.sh --workload config5
OUT=gpurun_out/r3_final
rm -rf $OUT && mkdir -p $OUT
for cfg in "5 20" "20 30"; do
  set -- $cfg; W=$1; S=$2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_w${W}s${S} -o trace -- python3 bench.py --steps $S --warmup $W --no-cpu-baseline > $OUT/bench_under_rocprof_w${W}s${S}.json 2> $OUT/trace_w${W}s${S}.err
  python3 scripts/trace_tail_stats.py $(find $OUT/trace_w${W}s${S} -name '*kernel_trace.csv' | head -1) $S > $OUT/timed_region_stats_w${W}s${S}.csv
  head -10 $OUT/timed_region_stats_w${W}s${S}.csv
  rm -f $(find $OUT/trace_w${W}s${S} -name '*kernel_trace.csv')
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_w5s20.json && cut -c1-300 $OUT/bench_w5s20.json
timeout -k 10 300 python bench.py > $OUT/bench_default.json && cut -c1-300 $OUT/bench_default.json
for wl in "config2 262144" "config2 65536" "config5 8388608"; do
  set -- $wl
  timeout -k 10 300 python bench.py --workload $1 --particles $2 --steps 30 --warmup 20 --no-cpu-baseline > $OUT/bench_$1_$2.json && python3 -c "import json; d=json.load(open('$OUT/bench_$1_$2.json')); print('$1 $2', round(d['ms_per_step'],4), d['stage_ms'])"
done
timeout -k 10 300 python bench.py --workload config4 --steps 10 --warmup 10 --no-cpu-baseline > $OUT/bench_config4_16777216.json && python3 -c "import json; d=json.load(open('$OUT/bench_config4_16777216.json')); print('config4', round(d['ms_per_step'],4), d['stage_ms'])"

# bench lines of round 2: the driver's arguments, the default ones, the cost of the timing events, and the plain
# `bench.py --gpus 2` (self-launched ranks; on a one-GPU box they share the GPU over gloo: a rehearsal)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="timeout -k 10 300 python bench.py"
true &&
$B --steps 20 --warmup 20 --no-cpu-baseline > gpurun_out/bench_w20.json && cat gpurun_out/bench_w20.json &&
for t in 0 1; do SPH_HIP_TIMING=$t $B --steps 20 --warmup 5 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TIMING=$t', d['ms_per_step'], d['stage_ms'])" ; done &&
true &&
SPH_HIP_TIMING_STRIDE=4 $B --steps 20 --warmup 5 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('STRIDE=4', d['ms_per_step'], d['stage_ms'])" &&
SPH_BENCH_SOLO=0 timeout -k 10 400 python bench.py --gpus 2 --steps 5 --warmup 5 --particles 1048576 > gpurun_out/bench_g2.json 2> gpurun_out/bench_g2.err && cat gpurun_out/bench_g2.json

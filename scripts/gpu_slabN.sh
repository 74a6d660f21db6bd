# W ranks (2..4) sharing the box's one GPU (gloo, host-staged messages): the multi-GPU bench at a narrower geometry.
# Functional rehearsal only; the 8-rank case is the driver's to run, never started from here.
# usage: W=4 ARGS="--workload config4" bash scripts/gpu_slabN.sh
set -e
cd $GRAFT_REPO_ROOT
export SPH_BENCH_BACKEND=gloo SPH_BENCH_SHARE_GPU=1 SPH_BENCH_SOLO=0
W=${W:-4}
if [ "$W" -gt 4 ]; then echo "rehearsals use at most 4 ranks" >&2; exit 2; fi
timeout -k 10 ${T:-500} python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 29621 bench.py --gpus $W --steps ${STEPS:-5} --warmup ${WARMUP:-15} $ARGS > gpurun_out/slabN.json 2> gpurun_out/slabN.err || { tail -30 gpurun_out/slabN.err; exit 1; }
tail -1 gpurun_out/slabN.json

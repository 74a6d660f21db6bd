# two ranks sharing the one GPU of the test box (gloo, host-staged messages): the slab bench with and without the
# overlapped exchange.  Functional rehearsal only -- both ranks compete for the same GPU.
set -e
cd $GRAFT_REPO_ROOT
export SPH_BENCH_BACKEND=gloo SPH_BENCH_SHARE_GPU=1 SPH_BENCH_SOLO=0
N=${N:-2097152}
for ov in 1 0; do
  SPH_BENCH_OVERLAP=$ov timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 10 --particles $N > gpurun_out/slab2_ov$ov.json 2> gpurun_out/slab2_ov$ov.err
  tail -1 gpurun_out/slab2_ov$ov.json
done

# the two command shapes the driver uses for N > 1, at their defaults, on the one-GPU box (auto rehearsal: gloo, ranks
# share the GPU): torchrun with the driver's flags, then the plain self-launching command
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
t0=$(date +%s)
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/drv_torchrun2.json 2> gpurun_out/drv_torchrun2.err
echo "torchrun x2: $(( $(date +%s) - t0 )) s"; tail -c 600 gpurun_out/drv_torchrun2.json; echo
t0=$(date +%s)
timeout -k 10 500 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/drv_self2.json 2> gpurun_out/drv_self2.err
echo "self-launch x2: $(( $(date +%s) - t0 )) s"; tail -c 300 gpurun_out/drv_self2.json; echo

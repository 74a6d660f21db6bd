# two ranks sharing one GPU (gloo): the config-4 strong-scaling workload at a reduced size, rehearsal only
set -e
cd $GRAFT_REPO_ROOT
export SPH_BENCH_BACKEND=gloo SPH_BENCH_SHARE_GPU=1
timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --steps 10 --warmup 10 --workload config4 --particles ${N:-4194304} > gpurun_out/slab2_c4.json 2> gpurun_out/slab2_c4.err
tail -1 gpurun_out/slab2_c4.json

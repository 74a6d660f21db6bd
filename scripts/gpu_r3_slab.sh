# round 3: slab tests on the GPU (new one-message protocol) + the self-launched 2-rank rehearsal of the bench
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest -x -q --timeout 300 --timeout-method thread -p no:cacheprovider tests/test_slab_gpu.py tests/test_statefile_gpu.py tests/test_zz_no_alloc.py -m gpu > gpurun_out/t_slab.log 2>&1 || { tail -60 gpurun_out/t_slab.log; exit 1; }
tail -2 gpurun_out/t_slab.log
timeout -k 10 600 python bench.py --gpus 2 --particles 2097152 --steps 10 --warmup 10 > gpurun_out/r3_g2.json 2> gpurun_out/r3_g2.err || { tail -30 gpurun_out/r3_g2.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r3_g2.json')); c=d['config']
print(d['value'], d['ms_per_step'], {k:c[k] for k in ('halo_violations','every_id_owned_once','host_waits_per_frame_rank0','transfer_rounds_per_exchange_rank0','exchange_ms_per_step_rank0','far_jump_reroutes','ghost_planes_at_end')})"

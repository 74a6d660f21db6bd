# round 3: the driver's N > 1 command at full per-GPU size, self-launched on the one-GPU box (ranks share the GPU, gloo):
#   bash scripts/gpu_r3_rehearse.sh "2 4" [steps] [warmup]
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for g in ${1:-2}; do
  timeout -k 10 900 python bench.py --gpus $g --steps ${2:-20} --warmup ${3:-5} > gpurun_out/r3_rehearse_g$g.json 2> gpurun_out/r3_rehearse_g$g.err || { tail -30 gpurun_out/r3_rehearse_g$g.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3_rehearse_g$g.json')); c=d['config']
print('gpus $g', d['value'], round(d['ms_per_step'],3), {k:c[k] for k in ('halo_violations','halo_violations_warmup','every_id_owned_once','host_waits_per_frame_rank0','transfer_rounds_per_exchange_rank0','far_jump_reroutes','ghost_planes_at_end','band_misses','ghost_to_owned_particles_worst_rank')})"
done

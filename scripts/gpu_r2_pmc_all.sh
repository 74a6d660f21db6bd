# all counter passes behind profiles/traffic.json: config 3 with the driver's arguments and the defaults, config 5's per-GPU block
set -e
WARMUP=5 STEPS=20 TAGP=w5 bash scripts/gpu_r2_pmc.sh > /dev/null
WARMUP=20 STEPS=30 TAGP=w20 bash scripts/gpu_r2_pmc.sh > /dev/null
WARMUP=5 STEPS=20 TAGP=c5w5 bash scripts/gpu_r2_pmc.sh --workload config5 > /dev/null
cd $GRAFT_REPO_ROOT
for t in w5 w20 c5w5; do grep -A9 "k_force_xsph_hits" gpurun_out/pmc_${t}_sq/summary.txt | head -10; done

# round 3: the counter passes of one bench configuration (WARMUP / STEPS), each in a run of its own; the TA pass also
# takes GRBM_GUI_ACTIVE (the kernel's own cycles: ta_busy_frac needs no assumed clock)
set -e
TAGP=${TAGP:-w${WARMUP:-20}}
bash scripts/gpu_pmc.sh ${TAGP}_sq "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS" "$@" > /dev/null &&
bash scripts/gpu_pmc.sh ${TAGP}_fetch "FETCH_SIZE" "$@" > /dev/null &&
bash scripts/gpu_pmc.sh ${TAGP}_write "WRITE_SIZE" "$@" > /dev/null &&
bash scripts/gpu_pmc.sh ${TAGP}_ta "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" "$@" > /dev/null

# per-wave timeline of the two neighbour kernels (debug build with -DSPH_WG_TRACE, built beforehand on the CPU):
#   bash scripts/gpu_wave_trace.sh [frames] [n]
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SPH_HIP_LIB=$PWD/sph_amd/libsph_hip_trace.so timeout -k 10 200 python scripts/wave_trace.py ${1:-25} ${2:-1048576} > gpurun_out/wave_trace_${1:-25}.txt 2>&1
cat gpurun_out/wave_trace_${1:-25}.txt

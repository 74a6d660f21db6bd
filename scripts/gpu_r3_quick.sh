# round 3 edit/measure loop: core parity tests, then A/B of the given libraries.  bash scripts/gpu_r3_quick.sh "<libs>" [trace] [extra test files]
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest -x -q --timeout 300 --timeout-method thread -p no:cacheprovider tests/test_parity_gpu.py tests/test_force_paths_gpu.py tests/test_golden_gpu.py tests/test_robustness_gpu.py tests/test_random_scenes_gpu.py tests/test_cpp_host_gpu.py $3 -m gpu > gpurun_out/t_quick.log 2>&1 || { tail -60 gpurun_out/t_quick.log; exit 1; }
tail -2 gpurun_out/t_quick.log
bash scripts/gpu_r3_ab.sh "$1" $2

# round 2, first GPU call: the no-alloc entry point alone, the new parity / robustness tests, then the whole GPU suite
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PT="python -u -m pytest -x -v --timeout 300 --timeout-method thread -p no:cacheprovider"
true &&
timeout -k 10 420 $PT tests/test_robustness_gpu.py -m gpu -s > gpurun_out/t_robust.log 2>&1 && tail -3 gpurun_out/t_robust.log &&
timeout -k 10 600 $PT tests/test_configs_gpu.py -m gpu > gpurun_out/t_configs.log 2>&1 && tail -3 gpurun_out/t_configs.log &&
timeout -k 10 900 $PT tests -m gpu > gpurun_out/t_all.log 2>&1 && tail -3 gpurun_out/t_all.log

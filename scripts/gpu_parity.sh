set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_parity_gpu.py -m gpu -x -v --timeout 120 --timeout-method thread 2>&1 | tee gpurun_out/parity1.log

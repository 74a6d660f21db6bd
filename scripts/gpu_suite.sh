# the whole GPU suite the way the round's end runs it (one process, -x), log under gpurun_out/
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest -x -v --timeout 300 --timeout-method thread -p no:cacheprovider tests -m gpu > gpurun_out/t_all.log 2>&1 && tail -4 gpurun_out/t_all.log

# prints "ms_per_step + stage_ms" of a short bench run (edit/measure loop helper)
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --steps 20 --warmup 20 --no-cpu-baseline "$@" > gpurun_out/stages.json && python3 -c "
import json;d=json.load(open('gpurun_out/stages.json'));print('STAGES', round(d['ms_per_step'],4), d['stage_ms'])"

#!/bin/bash
# usage: ab_run.sh TAG [ENV=VAL ...]  -> gpurun_out/ab_TAG.json (one bench line, no roofline / cpu baseline / eval)
tag=$1; shift
env "$@" python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-eval > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.log
python - <<PY
import json
d=json.loads(open('gpurun_out/ab_$tag.json').read().strip().splitlines()[-1])
print('$tag', d['ms_per_step'], d.get('losses_last_step'))
PY

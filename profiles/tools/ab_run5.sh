#!/bin/bash
# usage: ab_run5.sh TAG [ENV=VAL ...]  -> config 5 (ResNet-101, 512x512, B=32) bench line without roofline / cpu baseline / eval
tag=$1; shift
env "$@" python bench.py --arch resnet101 --image-size 512 --batch-size 32 --steps 10 --warmup 4 --no-roofline --no-cpu-baseline --no-eval > gpurun_out/ab5_$tag.json 2> gpurun_out/ab5_$tag.log
python - <<PY
import json
d=json.loads(open('gpurun_out/ab5_$tag.json').read().strip().splitlines()[-1])
print('$tag', d['ms_per_step'])
PY

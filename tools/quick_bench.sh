#!/usr/bin/env bash
# tools/quick_bench.sh [ENV=VAL ...] -- one bench run, prints value and per-kernel ms
for kv in "$@"; do export "$kv"; done
python bench.py --steps ${STEPS:-40} --warmup ${WARMUP:-5} --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('views/s', d['value'], 'ms', d['ms_per_step'], 'R', d['config']['num_rendered_mean'])
print('  '.join('%s=%.4f'%(k.replace('_kernel',''),v['ms_per_launch']*v['launches_per_step']) for k,v in d['roofline']['kernels'].items()))
"

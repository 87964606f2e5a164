#!/bin/bash
# A/B of alternate paths inside the product library on one box: tools/ab_mask.sh C3 0 32 0 32   (bench.py --debug-mask values)
CFG=$1; shift
for m in "$@"; do
  python bench.py --config $CFG --steps 30 --warmup 5 --no-cpu-baseline --no-extras --debug-mask $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']
print('mask $m', d['ms_per_step'], d['step_ms']['median'], {n:k[n]['ms'] for n in ('depth_sort','col_scatter','preprocess')})"
done

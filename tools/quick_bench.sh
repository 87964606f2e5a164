#!/bin/bash
# Two short bench runs (no CPU baseline, no extras) printing the step time and the per-kernel table: the A/B harness of a
# GPU session.  usage (GPU box, repo root): bash tools/quick_bench.sh [config] [extra bench args]
CFG=${1:-C3}; shift
for r in 1 2; do
python bench.py --config $CFG --steps 30 --warmup 5 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']
print(d['config']['workload'].split(':')[0], d['ms_per_step'], d['step_ms']['median'], {n:k[n]['ms'] for n in k})"
done

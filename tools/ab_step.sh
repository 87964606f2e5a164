#!/bin/bash
# A/B of library variants by the timed step alone (no per-kernel table): tools/ab_step.sh "C3" base v1 v2 ...
CFGS=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then LIBARG=""; else LIBARG="--library $PWD/gaussian-splatting_cc-comments_amd/libgsr_hip_$v.so"; fi
  for c in $CFGS; do
    for r in 1 2; do
    python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --no-extras $LIBARG 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('$v', d['config']['workload'].split(':')[0], d['ms_per_step'], d['step_ms'])"
    done
  done
done

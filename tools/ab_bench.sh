#!/bin/bash
# A/B of library variants (csrc/Makefile `variant`): one short bench run per variant and configuration, step time and the
# per-kernel table.  usage (GPU box, repo root): bash tools/ab_bench.sh "C3 C5" base v1 v2 ...   (base = the product library)
CFGS=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then LIBARG=""; else LIBARG="--library $PWD/gaussian-splatting_cc-comments_amd/libgsr_hip_$v.so"; fi
  for c in $CFGS; do
    st=30; [ "$c" = C5 ] && st=10
    python bench.py --config $c --steps $st --warmup 5 --no-cpu-baseline --no-extras $LIBARG 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']
print('$v', d['config']['workload'].split(':')[0], d['ms_per_step'], d['step_ms']['median'], {n:k[n]['ms'] for n in k if n not in ('preprocess','preprocess_color','binning')})"
  done
done

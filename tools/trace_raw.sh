#!/bin/bash
# rocprofv3 kernel trace of a short bench run -> every kernel of the process per step (tools/timeline.py --all):
# what runs BETWEEN the steps (the caller's fills, the status zeroing) as well as inside them.
#   usage (GPU box, repo root): bash tools/trace_raw.sh <tag> [extra bench args]
set -u
TAG=${1:?tag}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG.trace
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > "$OUT.log" 2>&1 || { tail -5 "$OUT.log"; exit 1; }
python3 "$ROOT/tools/timeline.py" "$(ls "$OUT"/*/*kernel_trace.csv | head -1)" 10 --all > "$ROOT/gpurun_out/${TAG}_raw.txt"
rm -rf "$OUT"
cat "$ROOT/gpurun_out/${TAG}_raw.txt"

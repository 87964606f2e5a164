#!/bin/bash
# rocprofv3 kernel trace of a short bench run -> the per-step GPU timeline (tools/timeline.py) and the kernel stats csv.
#   usage (GPU box, repo root): bash tools/trace_timeline.sh <tag> [config] [extra bench args]
# writes gpurun_out/<tag>_timeline.txt, gpurun_out/<tag>_kernel_stats.csv
set -u
TAG=${1:?tag}; CFG=${2:-C3}; shift; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG.trace
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 $ROOT/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > "$OUT.log" 2>&1 || { tail -5 "$OUT.log"; exit 1; }
python3 "$ROOT/tools/timeline.py" "$(ls "$OUT"/*/*kernel_trace.csv | head -1)" 10 > "$ROOT/gpurun_out/${TAG}_timeline.txt"
cp "$(ls "$OUT"/*/*kernel_stats.csv | head -1)" "$ROOT/gpurun_out/${TAG}_kernel_stats.csv"
rm -rf "$OUT"
cat "$ROOT/gpurun_out/${TAG}_timeline.txt"

#!/bin/bash
# Profiles one bench configuration on the GPU box: rocprofv3 kernel trace + stats, then the PMC passes (each in a run of
# its own, counters only: MI355X_MICROARCH.md "rocprofv3 PMC slots"), then the summary JSON.
#   usage (repo root, on the box):  bash tools/profile_run.sh <tag> [config]
# writes gpurun_out/<tag>/{trace,pmc_*}/..., gpurun_out/<tag>/kernel_stats.csv, gpurun_out/<tag>/pmc_summary.json
set -u
TAG=${1:?tag}
CFG=${2:-C3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $ROOT/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
cp "$(ls "$OUT"/trace/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats.csv"
python3 "$ROOT/tools/timeline.py" "$(ls "$OUT"/trace/*/*kernel_trace.csv | head -1)" 10 > "$OUT/timeline.txt" 2>&1 || tail -3 "$OUT/timeline.txt"
BENCH="python3 $ROOT/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
n=0
for counters in "FETCH_SIZE" "WRITE_SIZE" \
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
    "SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE" \
    "GRBM_GUI_ACTIVE" \
    "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE"; do
  rocprofv3 --pmc $counters --output-format csv -d "$OUT/pmc_$n" -- $BENCH > "$OUT/pmc_$n.log" 2>&1 || { tail -5 "$OUT/pmc_$n.log"; exit 1; }
  echo "pmc pass $n done: $counters"
  n=$((n + 1))
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_*" "$OUT/pmc_summary.json" "$CFG" "$OUT/kernel_stats.csv"

#!/bin/bash
# Evidence for the rows next to the path: the fused loss, the leaf-parameter kernels, the one-launch Adam, simple-knn.
#   usage (GPU box, repo root):  bash tools/profile_extras.sh <tag>
# writes gpurun_out/<tag>/{kernel_stats_train.csv, train_iteration_timeline.txt, kernel_stats_knn.csv, pmc_summary_extras.json}
set -u
TAG=${1:?tag}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
TRAIN="python3 $ROOT/tools/train_iter.py --config C3 --iters 12"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_train" -- $TRAIN > "$OUT/trace_train.log" 2>&1 || { tail -5 "$OUT/trace_train.log"; exit 1; }
cp "$(ls "$OUT"/trace_train/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats_train.csv"
python3 "$ROOT/tools/timeline.py" --all "$(ls "$OUT"/trace_train/*/*kernel_trace.csv | head -1)" 8 > "$OUT/train_iteration_timeline.txt" 2>&1 || tail -3 "$OUT/train_iteration_timeline.txt"
rm -rf "$OUT/trace_train"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_knn" -- python3 $ROOT/tools/knn_bench.py > "$OUT/trace_knn.log" 2>&1 || { tail -5 "$OUT/trace_knn.log"; exit 1; }
cp "$(ls "$OUT"/trace_knn/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats_knn.csv"
rm -rf "$OUT/trace_knn"
n=0
for counters in "FETCH_SIZE" "WRITE_SIZE" \
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
    "SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE"; do
  rocprofv3 --pmc $counters --output-format csv -d "$OUT/pmcx_$n" -- $TRAIN > "$OUT/pmcx_$n.log" 2>&1 || { tail -5 "$OUT/pmcx_$n.log"; exit 1; }
  echo "pmc pass $n done: $counters"
  n=$((n + 1))
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmcx_*" "$OUT/pmc_summary_extras.json" "C3 all_fused training iteration" "$OUT/kernel_stats_train.csv"
rm -rf "$OUT"/pmcx_*/

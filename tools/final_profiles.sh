#!/bin/bash
# The round's evidence (GPU box, repo root): bash tools/final_profiles.sh <tag> [a|b|c|all]   -> gpurun_out/<tag>_*
#   a: bench lines (C3, C2, C5), kernel stats + PMC of the step and of the extras;  b: tile clocks, skewed scenes, views in flight;  c: parity report
#   (three calls: together they exceed one gpurun call's 20 minutes)
set -u
TAG=${1:?tag}
PART=${2:-all}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/gpurun_out
cd $ROOT
if [ "$PART" != b ]; then
for c in C3 C2 C5; do
  extra=""; [ "$c" != C3 ] && extra="--no-cpu-baseline --no-extras"
  timeout -k 10 400 python bench.py --config $c $extra 2>/dev/null | tail -1 > $O/${TAG}_bench_$(echo $c | tr A-Z a-z).json.log
  python -c "import json;d=json.load(open('$O/${TAG}_bench_$(echo $c | tr A-Z a-z).json.log'));print('$c',d['value'],d['ms_per_step'])"
done
timeout -k 10 500 bash tools/profile_run.sh $TAG C3 > $O/${TAG}_profile_run.log 2>&1; tail -2 $O/${TAG}_profile_run.log
rm -rf $O/$TAG/trace $O/$TAG/pmc_*/
timeout -k 10 500 bash tools/profile_extras.sh ${TAG}x > $O/${TAG}_profile_extras.log 2>&1; tail -2 $O/${TAG}_profile_extras.log
fi
[ "$PART" = a ] && exit 0
if [ "$PART" != c ]; then
for s in uniform blob lowop; do
  k=""; [ "$s" = uniform ] && k="--forward-key --backward-key-length"
  timeout -k 10 300 python tools/tile_clock.py --config C3 --scene $s $k --out $O/${TAG}_tile_clock_c3_$s.txt > /dev/null 2>&1; tail -1 $O/${TAG}_tile_clock_c3_$s.txt | cut -c1-150
done
rm -f $O/${TAG}_skew.jsonl
for v in uniform blob lowop; do timeout -k 10 200 python tools/skew_bench.py --variant $v --out $O/${TAG}_skew.jsonl > /dev/null 2>&1; done; wc -l $O/${TAG}_skew.jsonl
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_vif.trace -- python3 $ROOT/bench.py --config C3 --steps 6 --warmup 2 --settle-steps 20 --views-per-rank 2 --views-in-flight 2 --staggered --no-cpu-baseline --no-extras > $O/${TAG}_vif.log 2>&1 )
python3 tools/timeline.py --raw "$(ls $O/${TAG}_vif.trace/*/*kernel_trace.csv | head -1)" 64 > $O/${TAG}_views_in_flight_timeline.txt 2>&1; rm -rf $O/${TAG}_vif.trace; head -3 $O/${TAG}_views_in_flight_timeline.txt
fi
[ "$PART" = b ] && exit 0
timeout -k 10 900 python tools/parity_report.py $O/${TAG}_parity_report.txt > $O/${TAG}_parity_report.log 2>&1; tail -3 $O/${TAG}_parity_report.log

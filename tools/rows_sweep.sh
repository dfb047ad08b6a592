# usage: bash tools/rows_sweep.sh <out.jsonl> "<shapes>" "<waves list, 0 = library choice>"
set -e
OUT=$1; S=$2; WV=${3:-0}
mkdir -p $(dirname $OUT)
TVL_ROWS_KERNEL=0 timeout -k 10 200 python tools/time_plans.py $S | sed "s/^{/{\"rows\": 0, \"waves\": 0, /" >> $OUT
for W in $WV; do
  TVL_ROWS_WAVES=$W TVL_ROWS_KERNEL=1 timeout -k 10 200 python tools/time_plans.py $S | sed "s/^{/{\"rows\": 1, \"waves\": $W, /" >> $OUT
done

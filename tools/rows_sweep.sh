# usage: bash tools/rows_sweep.sh <out.jsonl> "<shapes>"   (each shape with TVL_ROWS_LDS = 0 and 1)
set -e
OUT=$1; S=$2
mkdir -p $(dirname $OUT)
for L in 0 1; do
  TVL_ROWS_LDS=$L timeout -k 10 300 python tools/time_plans.py $S | sed "s/^{/{\"lds\": $L, /" >> $OUT
done

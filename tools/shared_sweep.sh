# usage: bash tools/shared_sweep.sh <out.jsonl> "<shapes>" "<settings of TVL_ROWS_SHARED>"
set -e
OUT=$1; S=$2; VARS=$3
mkdir -p $(dirname $OUT)
for V in $VARS; do
  TVL_ROWS_SHARED=$V TVL_ROWS_SHARED_MIN=${MINW:-2.5e7} timeout -k 10 300 python tools/time_plans.py $S | sed "s/^{/{\"shared\": \"$V\", /" >> $OUT
done

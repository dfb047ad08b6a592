# usage: bash tools/plans_sweep.sh <out.jsonl> "<shapes>" "<plans>" "<variants>" (the 4th argument sets TVL_MID_VARIANT, which only experimental builds of the mid-size kernel read; default 1)
set -e
OUT=$1; S=$2; PLANS=$3; VARS=${4:-1}
mkdir -p $(dirname $OUT)
for V in $VARS; do for P in $PLANS; do
  if [ "$P" = auto ]; then unset TVL_FORCE_PLAN; else export TVL_FORCE_PLAN=$P; fi
  TVL_MID_VARIANT=$V timeout -k 10 120 python tools/time_plans.py $S | sed "s/^{/{\"variant\": $V, /" >> $OUT
done; done

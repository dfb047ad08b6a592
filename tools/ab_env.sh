# usage: bash tools/ab_env.sh <out.jsonl> "<shapes>" "<label>=<ENV1=v,ENV2=v>" ...   (time_plans.py under each environment)
set -e
OUT=$1; S=$2; shift 2
mkdir -p $(dirname $OUT)
for spec in "$@"; do
  label=${spec%%=*}; envs=${spec#*=}
  ( for kv in ${envs//,/ }; do export $kv; done; timeout -k 10 300 python tools/time_plans.py $S | sed "s/^{/{\"shared\": \"$label\", /" >> $OUT )
done

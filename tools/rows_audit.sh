# Few-row dispatch audit: every shape under the library's own choice and with the few-row kernels switched off / forced on.
# usage: bash tools/rows_audit.sh <out.jsonl> "<M list>" "<N,K list>"
OUT=$1; MS=$2; NKS=$3
SHAPES=""
for nk in $NKS; do for m in $MS; do SHAPES="$SHAPES $m,$nk"; done; done
run() {  # run <tag> env...
  local tag=$1; shift
  env "$@" timeout -k 10 300 python tools/time_plans.py $SHAPES 2>/dev/null | sed "s/\"plan\": \"auto\"/\"plan\": \"$tag\"/" >> $OUT
}
run default TVL_NOP=1
run shared_off TVL_ROWS_SHARED=-1
run shared_any TVL_ROWS_SHARED_ANY=1
run staged_off TVL_ROWS_STAGED=0
run staged_any TVL_ROWS_STAGED=2
run both_off TVL_ROWS_SHARED=-1 TVL_ROWS_STAGED=0

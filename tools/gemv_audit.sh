# One-to-four-row dispatch audit: the library's own choice, the table-lookup kernel off / forced, the few-row kernels off.
# usage: bash tools/gemv_audit.sh <out.jsonl> "<M list>" "<N,K list>"
OUT=$1; MS=$2; NKS=$3
SHAPES=""
for nk in $NKS; do for m in $MS; do SHAPES="$SHAPES $m,$nk"; done; done
run() { local tag=$1; shift; env "$@" timeout -k 10 300 python tools/time_plans.py $SHAPES 2>/dev/null | sed "s/\"plan\": \"auto\"/\"plan\": \"$tag\"/" >> $OUT; }
run default TVL_NOP=1
run gemv_off TVL_GEMV=0
run gemv_any TVL_GEMV=2
run rows_off TVL_ROWS_SHARED=-1 TVL_ROWS_STAGED=0
run gemv_any_rows_off TVL_GEMV=2 TVL_ROWS_SHARED=-1 TVL_ROWS_STAGED=0

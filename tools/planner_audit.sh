# Times every shape of a grid under the library's own plan and under a list of pinned plans (tools/plans_sweep.sh), so that planner
# misses show up: bash tools/planner_audit.sh <out.jsonl> "<M list>" "<N,K list>"
OUT=$1; MS=$2; NKS=$3
SHAPES=""
for nk in $NKS; do for m in $MS; do SHAPES="$SHAPES $m,$nk"; done; done
bash $(dirname $0)/plans_sweep.sh $OUT "$SHAPES" "auto 1:128:256:1 1:128:256:2 1:128:256:4 1:128:128:1 1:128:128:2 1:128:128:4 1:128:128:8 1:64:128:1 1:64:128:2 1:64:128:4 1:64:128:8 0:32:128:4 0:32:128:8 2:256:256:1"

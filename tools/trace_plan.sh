# usage: bash tools/trace_plan.sh <tag> <plan or auto> M,N,K [M,N,K ...]   -> gpurun_out/r2/trace_<tag>.txt (per-shape kernel times)
set -e
TAG=$1; PLAN=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r2
cd /tmp && export TMPDIR=/tmp
if [ "$PLAN" != auto ]; then export TVL_FORCE_PLAN=$PLAN; fi
rm -rf /tmp/tp_$TAG
REPS=20 timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/tp_$TAG -o t --output-format csv -- python3 $R/tools/run_shapes.py "$@" > /dev/null 2>&1
python3 $R/tools/trace_shapes.py $(find /tmp/tp_$TAG -name "*kernel_trace.csv") 20 "$@" > $R/gpurun_out/r2/trace_$TAG.txt
cat $R/gpurun_out/r2/trace_$TAG.txt

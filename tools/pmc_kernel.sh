# usage: bash tools/pmc_kernel.sh <tag> <kernel substring> M,N,K  -> gpurun_out/pmc_<tag>.txt : SQ / TCP / TCC counters of one kernel on one shape
set -e
TAG=$1; PAT=$2; SHAPE=$3
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_$TAG.txt
mkdir -p $R/gpurun_out; : > $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "FETCH_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pk_${TAG}_$i
  REPS=12 timeout -s KILL 90 rocprofv3 --pmc $grp -d /tmp/pk_${TAG}_$i -o pmc --output-format csv -- python3 $R/tools/run_shapes.py $SHAPE > /dev/null 2>&1 || echo "pass failed: $grp" >> $O
  f=$(find /tmp/pk_${TAG}_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_summary.py $PAT $f >> $O; fi
done
cat $O

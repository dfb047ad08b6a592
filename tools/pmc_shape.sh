# usage: bash tools/pmc_shape.sh <tag> <plan|auto> <variant> M N K  -> gpurun_out/r2/pmc_<tag>.txt : counters of the matmul kernel(s) for one shape  (<variant> sets TVL_MID_VARIANT, read by experimental builds only: pass 0)
set -e
TAG=$1; PLAN=$2; VAR=$3; M=$4; N=$5; K=$6
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2/pmc_$TAG.txt
mkdir -p $R/gpurun_out/r2; : > $O
cd /tmp && export TMPDIR=/tmp
if [ "$PLAN" != auto ]; then export TVL_FORCE_PLAN=$PLAN; fi
export TVL_MID_VARIANT=$VAR
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pm_$TAG_$i
  timeout -s KILL 90 rocprofv3 --pmc $grp -d /tmp/pm_${TAG}_$i -o pmc --output-format csv -- python3 $R/tools/run_gemm.py $M $N $K 12 f16 > /dev/null 2>&1 || echo "pass failed: $grp" >> $O
  f=$(find /tmp/pm_${TAG}_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_summary.py gemm_mfma $f >> $O; fi
done
cat $O

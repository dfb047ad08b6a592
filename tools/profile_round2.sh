# Profiles kept as profiles/r02_*: kernel trace + stats of the bench command, counters of the 4096^3 GEMM, of two mid-size shapes before / after
# the 128-row tiles (the older tiles pinned with TVL_FORCE_PLAN), of the arena pack and of the few-row kernel.  Output: gpurun_out/prof_r02/
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the bench command under the kernel trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/pb2 -o bench --output-format csv -- python3 $R/bench.py --steps 200 --warmup 50 --no-cpu > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(find /tmp/pb2 -name "*kernel_stats.csv") $O/bench_kernel_stats.csv
python3 $R/tools/trace_tail.py $(find /tmp/pb2 -name "*kernel_trace.csv") gemm_mfma16_kernelIDF16_DF16_ 200 > $O/bench_timed_region.txt
echo "bench trace done"
pmc() {  # pmc <outfile> <kernel substring> <counter group> -- <program args...>
  local out=$1 pat=$2 grp=$3; shift 4
  rm -rf /tmp/pmc_run
  timeout -s KILL 120 rocprofv3 --pmc $grp -d /tmp/pmc_run -o pmc --output-format csv -- python3 "$@" > /dev/null 2>&1 || echo "pass failed: $grp" >> $out
  local f=$(find /tmp/pmc_run -name "*counter_collection.csv" | head -1)
  echo "== $grp" >> $out
  if [ -n "$f" ]; then python3 $R/tools/pmc_summary.py $pat $f >> $out; fi
}
G1="GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS"
G2="SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM"
# 2. the headline GEMM
: > $O/gemm_4096_pmc_summary.txt
for grp in "FETCH_SIZE" "WRITE_SIZE" "$G1" "$G2"; do pmc $O/gemm_4096_pmc_summary.txt gemm_mfma16_kernel "$grp" -- $R/tools/run_gemm.py 4096 4096 4096 12 f16; done
echo "4096 counters done"
# 3. mid-size shapes, before (older tiles, pinned) and after (the library's choice)
for spec in "256 4096 4096 0:128:128:4" "745 6144 2048 0:128:128:1"; do
  set -- $spec; M=$1; N=$2; K=$3; OLD=$4
  for which in before after; do
    out=$O/mid_${M}x${N}x${K}_${which}.txt; : > $out
    if [ $which = before ]; then export TVL_FORCE_PLAN=$OLD; else unset TVL_FORCE_PLAN; fi
    rm -rf /tmp/kt_run
    REPS=20 timeout -k 10 120 rocprofv3 --kernel-trace --stats -d /tmp/kt_run -o t --output-format csv -- python3 $R/tools/run_shapes.py $M,$N,$K > /dev/null 2>&1
    echo "== kernel trace (20 calls; $which, plan ${TVL_FORCE_PLAN:-auto})" >> $out
    python3 $R/tools/trace_shapes.py $(find /tmp/kt_run -name "*kernel_trace.csv") 20 $M,$N,$K >> $out
    grep -E "gemm_mfma|splitk" $(find /tmp/kt_run -name "*kernel_stats.csv") | cut -c1-220 >> $out
    for grp in "$G1" "$G2" "FETCH_SIZE" "WRITE_SIZE"; do pmc $out gemm_mfma "$grp" -- $R/tools/run_gemm.py $M $N $K 12 f16; done
  done
done
unset TVL_FORCE_PLAN
echo "mid counters done"
# 4. few-row kernel and decode kernel traffic, arena pack traffic
: > $O/rows_decode_pmc_summary.txt
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  pmc $O/rows_decode_pmc_summary.txt gemm_rows_staged_kernel "$grp" -- $R/tools/run_shapes.py 16,4096,4096
  pmc $O/rows_decode_pmc_summary.txt gemm_rows_shared_kernel "$grp" -- $R/tools/run_shapes.py 16,28672,4096
  pmc $O/rows_decode_pmc_summary.txt gemm_rows_shared_kernel "$grp" -- $R/tools/run_shapes.py 16,16384,16384
  pmc $O/rows_decode_pmc_summary.txt gemv_kernel "$grp" -- $R/tools/run_shapes.py 1,16384,16384
  pmc $O/rows_decode_pmc_summary.txt quantize_pack "$grp" -- $R/tools/time_pack.py 4096,4096
done
echo "all done"
ls -la $O

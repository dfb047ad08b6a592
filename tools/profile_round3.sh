# Profiles kept as profiles/r03_*: kernel trace + stats of the bench command, counters of the 4096^3 GEMM and of the persistent kernel on the ViT-Base
# fc1 shape, kernel trace of the batch-256 decode shapes, traffic of the few-row / one-row / pack kernels.  Output: gpurun_out/prof_r03/
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the bench command under the kernel trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/pb3 -o bench --output-format csv -- python3 $R/bench.py --steps 200 --warmup 50 --no-cpu > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(find /tmp/pb3 -name "*kernel_stats.csv") $O/bench_kernel_stats.csv
# the headline kernel (f16 -> f16, bias, one 256 x 256 tile per workgroup): its first 350 dispatches are the bench's own 300 clock-raising launches and the 50 warm-up steps, the next 200 the timed region
python3 $R/tools/trace_tail.py $(find /tmp/pb3 -name "*kernel_trace.csv") gemm_mfma16_kernelIDF16_DF16_Lb1ELi0ELi1ELb0ELi4ELi256ELi8ELb0ELi2ELi32ELi2E 200 350 > $O/bench_timed_region.txt
python3 $R/tools/trace_tail.py $(find /tmp/pb3 -name "*kernel_trace.csv") gemm_mfma16_kernelIDF16_DF16_Lb1ELi0ELi1ELb0ELi4ELi256ELi8ELb0ELi2ELi32ELi2E 350 0 >> $O/bench_timed_region.txt
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
# 3. the persistent kernel on ViT-Base fc1 (25216 x 3072 x 768) and the batch-256 decode shapes under the kernel trace
: > $O/vit_fc1_pmc_summary.txt
for grp in "FETCH_SIZE" "WRITE_SIZE" "$G1" "$G2"; do pmc $O/vit_fc1_pmc_summary.txt gemm_mfma16_kernel "$grp" -- $R/tools/run_gemm.py 25216 3072 768 12 f16; done
rm -rf /tmp/kt_run
REPS=20 timeout -k 10 120 rocprofv3 --kernel-trace --stats -d /tmp/kt_run -o t --output-format csv -- python3 $R/tools/run_shapes.py 25216,2304,768 25216,768,768 25216,3072,768 25216,768,3072 256,14336,2048 256,2048,2048 256,2048,8192 > $O/run_shapes.log 2>&1
cp $(find /tmp/kt_run -name "*kernel_trace.csv") $O/configs_kernel_trace.csv
python3 $R/tools/trace_shapes.py $O/configs_kernel_trace.csv 20 25216,2304,768 25216,768,768 25216,3072,768 25216,768,3072 256,14336,2048 256,2048,2048 256,2048,8192 > $O/configs_kernel_trace.txt
echo "config shapes done"
# 4. few-row kernel and decode kernel traffic, arena pack traffic (the order tools/make_traffic_json.py expects)
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

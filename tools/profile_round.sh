set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof_r01
mkdir -p $O
# 1. kernel trace + stats of the bench command itself
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/pb -o bench --output-format csv -- python3 $R/bench.py --steps 200 --warmup 50 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(find /tmp/pb -name "*kernel_stats.csv") $O/bench_kernel_stats.csv
# the timed region alone: the last 200 dispatches of the GEMM kernel in the trace (the stats row also averages the cold pre-warm launches)
python3 $R/tools/trace_tail.py $(find /tmp/pb -name "*kernel_trace.csv") gemm_mfma16_kernelIDF16_DF16_ 200 > $O/bench_timed_region.txt
# 2. counters for the 4096^3 GEMM kernel, one pass per counter group
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 120 rocprofv3 --pmc $grp -d /tmp/pc_$tag -o pmc --output-format csv -- python3 $R/tools/run_gemm.py 4096 4096 4096 12 f16 > /dev/null 2>&1
  echo "== $grp" >> $O/gemm_4096_pmc_summary.txt
  python3 $R/tools/pmc_summary.py gemm_mfma16_kernel $(find /tmp/pc_$tag -name "*counter_collection.csv") >> $O/gemm_4096_pmc_summary.txt
done
# 3. traffic of the decode kernel (M=1, 16384^2) and the pack kernel
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -s KILL 120 rocprofv3 --pmc $grp -d /tmp/pd_$grp -o pmc --output-format csv -- python3 $R/tools/run_shapes.py 1,16384,16384 > /dev/null 2>&1
  echo "== $grp (decode 1x16384x16384)" >> $O/decode_pmc_summary.txt
  python3 $R/tools/pmc_summary.py gemv_kernel $(find /tmp/pd_$grp -name "*counter_collection.csv") >> $O/decode_pmc_summary.txt
done
cat $O/bench_timed_region.txt $O/gemm_4096_pmc_summary.txt $O/decode_pmc_summary.txt
head -12 $O/bench_kernel_stats.csv | cut -c1-180
tail -c 1500 $O/bench_under_rocprof.json

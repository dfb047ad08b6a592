# Average shader clock during a kernel = GRBM_GUI_ACTIVE (cycles while the GPU is busy, summed over the 8 XCDs) / 8 / the dispatch's own
# duration, both from ONE rocprofv3 run (--pmc GRBM_GUI_ACTIVE --kernel-trace).  usage: bash tools/clock_probe.sh M N K [reps]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
M=$1; N=$2; K=$3; REPS=${4:-40}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/clk_run
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/clk_run -o c --output-format csv -- python3 $R/tools/run_gemm.py $M $N $K $REPS f16 > /dev/null 2>&1
python3 - <<PY
import csv, glob
cc = glob.glob('/tmp/clk_run/**/*counter_collection.csv', recursive=True)[0]
kt = glob.glob('/tmp/clk_run/**/*kernel_trace.csv', recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    if 'gemm_mfma' in r['Kernel_Name']:
        dur[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
rows = []
for r in csv.DictReader(open(cc)):
    if 'gemm_mfma' in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE' and r['Dispatch_Id'] in dur:
        rows.append((float(r['Counter_Value']) / 8.0, dur[r['Dispatch_Id']]))
rows = rows[len(rows) // 2:]          # the second half: past the clock ramp
cyc = sum(c for c, _ in rows) / len(rows); ns = sum(d for _, d in rows) / len(rows)
print(f"$M x $N x $K: {len(rows)} dispatches, {cyc:.0f} busy cycles per XCD, {ns/1e3:.1f} us per dispatch under the counter pass -> {cyc/ns:.2f} GHz average shader clock")
PY

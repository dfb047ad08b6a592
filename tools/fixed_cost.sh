cd /tmp && export TMPDIR=/tmp
S="4096,4096,64 4096,4096,128 4096,4096,256 4096,4096,512 4096,4096,1024 4096,4096,4096"
timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/pf -o t --output-format csv -- python3 /root/repo/tools/run_shapes.py $S > /dev/null 2>&1
python3 /root/repo/tools/trace_shapes.py $(find /tmp/pf -name "*kernel_trace.csv") 20 $S

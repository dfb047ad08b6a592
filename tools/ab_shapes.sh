A=tools/ab_old/ternary-vllm_amd/libternary_hip.so
B=ternary-vllm_amd/libternary_hip.so
for shape in "4096 4096 4096" "25216 2304 768" "25216 3072 768" "25216 768 3072" "8192 4096 4096"; do
  timeout -k 10 100 python tools/ab_libs.py $shape $A $B 2>/dev/null || exit 1
done

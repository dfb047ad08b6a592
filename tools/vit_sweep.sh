set -e
S="25216,2304,768 25216,768,768 25216,3072,768 25216,768,3072 8192,8192,1024"
for PM in 0 1; do for P in 1:128:256:1 2:256:256:1; do
  TVL_MID_PERSIST=$PM TVL_FORCE_PLAN=$P timeout -k 10 200 python tools/time_plans.py $S | sed "s/^{/{\"persist\": $PM, /" >> gpurun_out/r2/vit1.jsonl
done; done
cat gpurun_out/r2/vit1.jsonl

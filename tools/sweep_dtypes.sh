for dt in f16 bf16; do
  export TVL_AB_DTYPE=$dt
  for s in "1 4096 4096" "1 16384 16384" "8 4096 4096" "8 8192 2048" "16 16384 4096" "32 4096 4096" "64 4096 4096" "745 6144 2048" "4096 4096 4096"; do
    timeout -k 10 100 python tools/ab_libs.py $s ternary-vllm_amd/libternary_hip.so | sed "s/^/$dt /" || exit 1
  done
done

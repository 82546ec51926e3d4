# training step A/B on one box: the previous library (popnet_amd/build/libpopnet_prev.so) against the current one
for p in bf16x3 fp32; do
  for i in 1 2; do
    POPNET_LIB_PATH=popnet_amd/build/libpopnet_prev.so python3 scripts/train_bench.py 32 20 $p 2>&1 | grep "ms/step" | sed 's/^/prev /'
    python3 scripts/train_bench.py 32 20 $p 2>&1 | grep "ms/step" | sed 's/^/new  /'
  done
done
